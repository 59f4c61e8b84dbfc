// 4mc_amd/csrc/lz4_parse.hip - K1p: finds the token chain of an LZ4 block with the whole workgroup instead of one
// serial walker (the serial loop it replaces: native/lz4/lz4.c:1936-2339, one token after the other).
//
// Where a token starts depends on every token before it.  The chain is recovered in pieces of kPch stream bytes:
//   A  every byte position p is treated as a token start and gets the distance to the token that would follow it
//      ("simple" tokens: at most one length continuation byte each; anything else is an ESCAPE value),
//   B  one lane per 64-byte row sweeps its row backwards and turns the distances into "where does a chain that
//      enters the row at p leave it" - 64 dependent steps for 16 KiB, every lane busy,
//   C  every row guesses where the real chain enters it by walking the exit table from a few rows back (chains that
//      start at different bytes merge after a few tokens), then the guesses are VERIFIED against the left neighbour's
//      exit, row 0 being exact; rows that disagree are repaired from the left until nothing changes.  The result is
//      exact, the guess only decides how many repair rounds are needed,
//   D  every row decodes its own tokens (lengths, offset), prefix sums give sequence numbers and output positions,
//      every rule of the reference's safe decoder that can reject a block is checked, and the records are written:
//      tok[i] and, for every output window, the sequence that covers its first byte (lz4par.h).
// Escapes (long length runs, the last bytes of the block) take a scalar decoder that follows the reference's rules
// one byte at a time.  A block that breaks ANY rule here is handed to the exact kernel (kRetry), which reproduces the
// reference's return code; so this kernel only ever accepts streams the reference accepts, with the same meaning.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"

using namespace lz4par;

namespace {

constexpr int kPch   = 16384;                 // stream bytes per step
constexpr int kMarg  = 320;                   // bytes staged behind the piece: a simple token looks at most 274 ahead
constexpr int kPT    = 256;                   // threads = rows of 64 bytes
constexpr int kTile  = kPch + kMarg;
constexpr int kTilePhys = kTile + 4 * (kTile / 64) + 8;
constexpr int kTabPhys  = kPch + 4 * (kPch / 64);
constexpr int kStageCap = kTabPhys / 8;       // the exit table's space carries the records of a piece on their way out
constexpr int POS_END = 0x3fffffff;           // the chain ended with a valid last sequence
constexpr int POS_BAD = 0x40000000;           // the chain ran into something the strict rules reject
constexpr int POS_UNK = 0x40000001;           // a speculative walk gave up
constexpr int POS_NONE = 0x40000002;
constexpr int kSpecCap = 0;                   // length bytes a walk from an arbitrary entry offset follows before it gives up (the real chain never does)
constexpr int kGroup = 1024;                 // rows are grouped by 16 for the two-level chain walk

// rows are padded by one bank so that lanes working on the same column of 64 different rows hit 32 different banks
__device__ __forceinline__ int phys(int i) { return i + ((i >> 6) << 2); }

struct Ctx {
    const uint8_t* src;     // block's stream
    int iend;               // its size
    int oend;               // output capacity
    int c0;                 // stream position of the piece
    const uint8_t* tile;    // LDS copy of [c0, c0 + kTile), zero beyond iend
};

__device__ __forceinline__ uint32_t cb(const Ctx& c, int p)
{
    const int i = p - c.c0;
    if (unsigned(i) < unsigned(kTile)) return c.tile[phys(i)];
    return c.src[p];
}

struct Tok { uint32_t ll, ml, off; int next; int kind; };     // next: stream position
enum { kSeq = 0, kFinal = 1, kInvalid = 2, kGaveUp = 3 };

// Scalar decoder of the token at p with the stream-side rules of the strict path (lz4.c:2114-2325 as restated in
// oracle/lz4_port.c: safe_literals / copy_match): what only depends on the compressed stream is decided here, the rules
// that involve the output position are checked where it is known (phase D).
__device__ Tok gdecode(const Ctx& c, int p, int cap)
{
    Tok t; t.ll = t.ml = t.off = 0; t.next = 0; t.kind = kInvalid;
    const int iend = c.iend;
    if (p >= iend) return t;
    const uint32_t tk = cb(c, p);
    int ip = p + 1;
    uint32_t lit = tk >> 4;
    if (lit == 15) {
        const int lim = iend - 15;                         // more_len(.., iend - 15, check_first)
        if (ip >= lim) return t;
        uint32_t b; int n = 0;
        do { if (ip + 1 > lim) return t; if (++n > cap) { t.kind = kGaveUp; return t; } b = cb(c, ip); ip++; lit += b; } while (b == 255);
    }
    t.ll = lit;
    if ((long long)ip + lit > (long long)iend - 8) {       // must be the block's last, literal-only sequence
        if ((long long)ip + lit == iend) { t.kind = kFinal; t.next = iend; }
        return t;
    }
    ip += int(lit);
    t.off = cb(c, ip) | (cb(c, ip + 1) << 8);
    ip += 2;
    uint32_t ml = tk & 15;
    if (ml == 15) {
        const int lim = iend - 4;                          // more_len(.., iend - 4)
        uint32_t b; int n = 0;
        do { if (ip + 1 > lim) return t; if (++n > cap) { t.kind = kGaveUp; return t; } b = cb(c, ip); ip++; ml += b; } while (b == 255);
    }
    t.ml = ml + 4; t.next = ip; t.kind = kSeq;
    return t;
}

// the common token shapes straight from the tile: i < kPch is the local position; false = take gdecode
__device__ __forceinline__ bool sdecode(const Ctx& c, int i, uint32_t& ll, uint32_t& ml, uint32_t& off, int& nexti, int& liti)
{
    const uint32_t tk = c.tile[phys(i)], b1 = c.tile[phys(i + 1)];
    const uint32_t ll0 = tk >> 4, ml0 = tk & 15;
    bool esc = (ll0 == 15) & (b1 == 255);
    ll = ll0 == 15 ? 15 + b1 : ll0;
    liti = i + 1 + (ll0 == 15 ? 1 : 0);
    const int q = liti + int(ll);
    const uint32_t o0 = c.tile[phys(q)], o1 = c.tile[phys(q + 1)], e1 = c.tile[phys(q + 2)];
    off = o0 | (o1 << 8);
    esc |= (ml0 == 15) & (e1 == 255);
    ml = ml0 == 15 ? 19 + e1 : ml0 + 4;
    nexti = q + 2 + (ml0 == 15 ? 1 : 0);
    esc |= c.c0 + nexti > c.iend - 16;                     // near the end every rule matters: scalar decoder
    return !esc;
}

// continuation bytes of a match length ml (>= 19 means the nibble was 15): 1 + (ml - 19) / 255
__device__ __forceinline__ uint32_t ml_ext_bytes(uint32_t ml) { return ml >= 19u ? 1u + (ml - 19u) / 255u : 0u; }

// first chain position at or behind the end of pos's row, for a chain that passes through pos (local, < kPch)
__device__ int step(const Ctx& c, const uint8_t* tab, int pos, int cap)
{
    const int lim = ((pos >> 6) + 1) << 6;
    const uint32_t x = tab[phys(pos)];
    if (x != 255) return lim + int(x);
    while (pos < lim) {
        const Tok t = gdecode(c, c.c0 + pos, cap);
        if (t.kind == kInvalid) return POS_BAD;
        if (t.kind == kGaveUp) return POS_UNK;
        if (t.kind == kFinal) return POS_END;
        pos = t.next - c.c0;
    }
    return pos;
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)       // inclusive
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}

// profiling build (make prof): thread 0 charges the cycles since the previous mark to a phase counter (units of 256 clk)
#ifdef K1X_PROF
#define PPROF_DECL unsigned long long pp_[10] = {0,0,0,0,0,0,0,0,0,0}; unsigned long long pl_ = __builtin_readcyclecounter();
#define PPT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pp_[i] += n_ - pl_; pl_ = n_; } while (0)
#define PPADD(i, v) do { pp_[i] += (v); } while (0)
#define PPROF_OUT(h) do { if (t == 0) for (int i_ = 0; i_ < 10; i_++) (h)->dbg[i_] = uint32_t(pp_[i_] >> (i_ < 8 ? 8 : 0)); } while (0)
#else
#define PPROF_DECL
#define PPT(i) do {} while (0)
#define PPADD(i, v) do {} while (0)
#define PPROF_OUT(h) do {} while (0)
#endif

struct PShared {
    uint32_t wsum[2][4];        // wave totals of the two scans
    int      gmin;
    int      flags;             // bit 0: a token broke a rule, bit 1: the last sequence was seen
    uint32_t fin_total;
    int      xs[kPT];           // first chain position of every row (POS_NONE: the chain does not touch the row)
    uint16_t gmap[kPch / 1024][64];   // positions are < 2^15; POS_* are stored as 0xFFF0 + (value - POS_END)
    int      gentry[kPch / 1024 + 1];
};

} // namespace

__global__ __launch_bounds__(kPT, 4)
void lz4_parse_kernel(const uint8_t* __restrict__ src_base, const uint8_t* dst_base, fourmc_block* blocks,
                      uint32_t nblocks, int container_mode, uint8_t* work)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTilePhys];
    __shared__ __attribute__((aligned(16))) uint8_t tab[kTabPhys];
    __shared__ PShared sh;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const fourmc_block blk = uniform_block(blocks[b]);
    uint8_t* slot = work + size_t(b) * kSlotBytes;
    ParHdr* hdr = reinterpret_cast<ParHdr*>(slot);
    uint4* wdesc = reinterpret_cast<uint4*>(slot + kWdescOff);
    uint2* rec = reinterpret_cast<uint2*>(slot + kTokOff);
    const uint8_t* src = src_base + blk.src_off;

    auto leave = [&](int32_t status, int result, bool set_result) {
        if (t == 0) { hdr->status = status; if (set_result) blocks[b].result = result; }
    };
    if (container_mode) {
        if (blk.result == FOURMC_BLK_BADSUM) { leave(kDone, 0, false); return; }
        if (blk.src_len == blk.dst_cap) {                  // stored block (native/4mc.c:635-642): plain copy, all four waves
            uint8_t* dst = const_cast<uint8_t*>(dst_base) + blk.dst_off;
            const int n = int(blk.src_len);
            const int per = ((n + 3) / 4 + 15) & ~15;
            const int lo = min(n, wave * per), hi = min(n, lo + per);
            wave_copy(dst + lo, src + lo, hi - lo, lane);
            leave(kDone, n, true);
            return;
        }
    }
    if (blk.dst_cap < 64 || blk.src_len < 16 || blk.src_len > kSrcMax || blk.dst_cap > kDstMax) { leave(kRetry, kRetryCode, true); return; }

    Ctx c; c.src = src; c.iend = int(blk.src_len); c.oend = int(blk.dst_cap); c.c0 = 0; c.tile = tile;
    const uint32_t a0 = uint32_t(reinterpret_cast<uintptr_t>(dst_base + blk.dst_off) & 127);
    uint32_t seq_base = 0, out_base = 0;
    
    uint2* stage = reinterpret_cast<uint2*>(tab);
    PPROF_DECL

    for (;;) {
        PPT(7);
        PPADD(9, 1);
        // ---------------------------------------------------------------- stage [c0, c0 + kTile)
        __syncthreads();
        if (t == 0) { sh.flags = 0; }
        for (int k = t * 16; k < kTile; k += kPT * 16) {
            const int g = c.c0 + k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g + 16 <= c.iend) v = ld16u(src + g);
            else if (g < c.iend) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int i = 0; i < c.iend - g; i++) w[i >> 2] |= uint32_t(src[g + i]) << (8 * (i & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            uint32_t* d = reinterpret_cast<uint32_t*>(tile + phys(k));
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        PPT(0);
        // ---------------------------------------------------------------- A: distance to the next token, per byte
        for (int i = 4 * t; i < kPch; i += 4 * kPT) {
            const uint32_t w0 = *reinterpret_cast<const uint32_t*>(tile + phys(i));
            const uint32_t w1 = *reinterpret_cast<const uint32_t*>(tile + phys(i + 4));
            const uint64_t w = uint64_t(w0) | (uint64_t(w1) << 32);
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t tk = uint32_t(w >> (8 * k)) & 255u, b1 = uint32_t(w >> (8 * k + 8)) & 255u;
                const uint32_t ll0 = tk >> 4, ml0 = tk & 15;
                bool esc = (ll0 == 15) & (b1 == 255);
                const uint32_t ll = ll0 == 15 ? 15 + b1 : ll0;
                int q = i + k + 1 + (ll0 == 15 ? 1 : 0) + int(ll) + 2;
                const uint32_t e1 = tile[phys(q)];
                esc |= (ml0 == 15) & (e1 == 255);
                q += (ml0 == 15 ? 1 : 0);
                const int delta = q - (i + k);
                esc |= (delta > 254) | (c.c0 + q > c.iend - 16);
                out |= (esc ? 255u : uint32_t(delta)) << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(tab + phys(i)) = out;
        }
        __syncthreads();
        PPT(1);
        // ---------------------------------------------------------------- B: exits of every row, backwards
        {
            uint8_t* row = tab + 68 * t;
            uint32_t r[16];
#pragma unroll
            for (int k = 0; k < 16; k++) r[k] = reinterpret_cast<const uint32_t*>(row)[k];
            // three columns per round trip: a distance is at least 3, so column e only reads exits of columns > e + 2
#pragma unroll
            for (int e0 = 63; e0 >= 0; e0 -= 3) {
                uint32_t x[3]; int n[3]; uint32_t d[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int e = e0 - u;
                    if (e < 0) continue;
                    d[u] = (r[e >> 2] >> (8 * (e & 3))) & 255u;
                    n[u] = e + int(d[u]);
                    x[u] = row[min(n[u], 63)];
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const int e = e0 - u;
                    if (e < 0) continue;
                    const uint32_t v = d[u] == 255u ? 255u : (n[u] >= 64 ? uint32_t(n[u] - 64) : x[u]);
                    row[e] = uint8_t(v);
                }
            }
        }
        __syncthreads();
        PPT(2);
        // ---------------------------------------------------------------- C: where the chain enters every row (exact, two levels)
        // C1: for each group of 16 rows and each of the 64 offsets a chain can enter its first row at: where it leaves the group
        {
            const int g = t >> 4;
            int pos[4];
#pragma unroll
            for (int i = 0; i < 4; i++) pos[i] = kGroup * g + (t & 15) + 16 * i;
            const int lim = kGroup * (g + 1);
            for (int it = 0; it < kGroup / 64; it++) {
#pragma unroll
                for (int i = 0; i < 4; i++) if (pos[i] < lim) pos[i] = step(c, tab, pos[i], kSpecCap);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) sh.gmap[g][(t & 15) + 16 * i] = uint16_t(pos[i] >= POS_END ? 0xFFF0 + (pos[i] - POS_END) : pos[i]);
        }
        for (int r = t; r < kPT; r += kPT) sh.xs[r] = POS_NONE;
        __syncthreads();
        // C2: the real chain from group to group (one lane; at most one table lookup per group unless a long token lands deep)
        if (t == 0) {
            int pos = 0;
            for (int g = 0; g < kPch / kGroup; g++) {
                sh.gentry[g] = pos;
                const int lim = kGroup * (g + 1);
                if (pos >= lim) continue;
                const int o = pos - kGroup * g;
                int v = o < 64 ? int(sh.gmap[g][o]) : POS_UNK;
                if (o < 64 && v >= 0xFFF0) v = POS_END + (v - 0xFFF0);
                if (v == POS_UNK) { v = pos; while (v < lim) v = step(c, tab, v, 0x7fffffff); }
                pos = v;
            }
            sh.gentry[kPch / kGroup] = pos;
        }
        __syncthreads();
        // C3: inside every group, from its real entry: the first chain position of every row
        if (t < kPch / kGroup) {
            int pos = sh.gentry[t];
            const int lim = kGroup * (t + 1);
            while (pos < lim) { sh.xs[pos >> 6] = pos; pos = step(c, tab, pos, 0x7fffffff); }
        }
        __syncthreads();
        const int E = sh.xs[t];
        const int next_local = sh.gentry[kPch / kGroup];
        PPT(3);
        // ---------------------------------------------------------------- D: this row's tokens
        const int lim = 64 * (t + 1);
        uint32_t n_tok = 0, n_out = 0;
        {
            int pos = E;
            while (pos < lim) {
                uint32_t ll, ml, off; int nx, li; int kind = kSeq;
                if (!sdecode(c, pos, ll, ml, off, nx, li)) {
                    const Tok tk = gdecode(c, c.c0 + pos, 0x7fffffff);
                    ll = tk.ll; ml = tk.ml; nx = tk.next - c.c0; kind = tk.kind;
                }
                if (kind == kInvalid) { atomicOr(&sh.flags, 1); break; }
                n_tok++; n_out += ll + (kind == kFinal ? 0u : ml);
                if (kind == kFinal) break;
                pos = nx;
            }
        }
        __syncthreads();        // everyone is done with the exit table: its space now stages records
        PPT(4);
        uint32_t pn, po, N, O;
        {
            const uint32_t in = wave_scan_add(n_tok), io = wave_scan_add(n_out);
            if (lane == 63) { sh.wsum[0][wave] = in; sh.wsum[1][wave] = io; }
            __syncthreads();
            uint32_t bn = 0, bo = 0; N = 0; O = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint32_t a = sh.wsum[0][w], o = sh.wsum[1][w]; if (w < wave) { bn += a; bo += o; } N += a; O += o; }
            pn = bn + in - n_tok; po = bo + io - n_out;
        }
        if (sh.flags & 1) { leave(kRetry, kRetryCode, true); return; }
        PPT(5);
        for (uint32_t r0 = 0; r0 < N; r0 += kStageCap) {
            uint32_t k = pn, opos = out_base + po;
            int pos = E;
            while (pos < lim && k < r0 + kStageCap) {
                uint32_t ll, ml, off; int nx, li; int kind = kSeq;
                if (!sdecode(c, pos, ll, ml, off, nx, li)) {
                    const Tok tk = gdecode(c, c.c0 + pos, 0x7fffffff);
                    ll = tk.ll; ml = tk.ml; off = tk.off; nx = tk.next - c.c0; kind = tk.kind;
                    li = nx - (kind == kFinal ? 0 : 2 + int(ml_ext_bytes(ml))) - int(ll);
                }
                if (kind == kFinal) { ml = 0; off = 0; }
                const uint32_t litpos = uint32_t(c.c0 + li);
                if (k >= r0) {
                    // output-side rules (lz4.c:2175-2225 literals, :2250 / :2315-2317 match)
                    bool ok;
                    const long long op = opos, oe = c.oend;
                    if (kind == kFinal) ok = op + ll <= oe;
                    else ok = op + ll <= oe - 12 && off != 0 && (long long)off <= op + ll && op + ll + ml <= oe - 5;
                    if (!ok) { atomicOr(&sh.flags, 1); break; }
                    { const uint32_t lc = min(ll, kRecLLSat); stage[k - r0] = make_uint2(litpos | ((lc & 511u) << 23), off | (min(ml, kRecMLSat) << 16) | ((lc >> 9) << 27)); }
                    const uint32_t sp0 = opos + a0, sp1 = sp0 + ll + ml;
                    const uint4 dA = make_uint4(seq_base + k, opos, uint32_t(c.c0 + pos), litpos);
                    const uint4 dB = make_uint4(ll, ml, off, 0);
                    uint32_t w = (sp0 + kWin - 1) >> kWinLog;
                    if (seq_base + k == 0) w = 0;                          // the block's first sequence also owns window 0 (shift a0)
                    for (; (w << kWinLog) < sp1 || (w == 0 && seq_base + k == 0); w++) { wdesc[2 * w] = dA; wdesc[2 * w + 1] = dB; }
                    if (kind == kFinal) { sh.fin_total = opos + ll; atomicOr(&sh.flags, 2); }
                }
                if (kind == kFinal) break;
                k++; opos += ll + ml; pos = nx;
            }
            __syncthreads();
            const uint32_t cnt = min(uint32_t(kStageCap), N - r0);
            for (uint32_t i = t; i < cnt; i += kPT) rec[seq_base + r0 + i] = stage[i];
            __syncthreads();
        }
        PPT(6);
        const int flags = sh.flags;
        if (flags & 1) { leave(kRetry, kRetryCode, true); return; }
        seq_base += N; out_base += O;
        if (flags & 2) break;
        if (next_local >= POS_END) { leave(kRetry, kRetryCode, true); return; }     // ended without a last sequence, or broke a rule
        c.c0 += next_local;
    }
    if (t == 0) {
        const uint32_t total = sh.fin_total;
        const uint32_t nwin = (total + a0 + kWin - 1) >> kWinLog;
        wdesc[2 * nwin] = make_uint4(seq_base - 1, total, uint32_t(c.iend), uint32_t(c.iend));
        wdesc[2 * nwin + 1] = make_uint4(0, 0, 0, 0);
        hdr->nseq = seq_base; hdr->total = total; hdr->nwin = nwin; hdr->a0 = a0;
        PPROF_OUT(hdr);
        hdr->status = total ? kParsed : kDone;
        if (!total) blocks[b].result = 0;
    }
}

extern "C" hipError_t fourmc_launch_lz4_parse(const void* d_src, const void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                               int container_mode, void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_parse_kernel, dim3(n), dim3(kPT), 0, stream, static_cast<const uint8_t*>(d_src),
                       static_cast<const uint8_t*>(d_dst), d_blocks, n, container_mode, static_cast<uint8_t*>(d_work));
    return hipGetLastError();
}
