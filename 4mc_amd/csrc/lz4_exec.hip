// 4mc_amd/csrc/lz4_exec.hip - K1x: executes the sequence records of lz4_parse.hip; the LZ77 copy loop of the reference
// (native/lz4/lz4.c:2060-2110, :2300-2325) as a dataflow over a 32 KiB window of the output held in LDS.
//
// One workgroup per block, output produced in WINDOWS of 1 KiB (lz4par.h).  What a match copies is the only thing that
// depends on earlier output, and mostly on output produced long before: the block's critical path runs through the few
// matches whose source was produced a moment ago.  The waves of the workgroup split the work by that dependency:
//   * kNP "pre" waves take windows round robin, up to kAhead windows ahead of the completed output.  A lane is one
//     sequence: it decodes its token from the wave's private copy of the stream (prefetched a window ahead), copies its
//     literals into the ring and copies its match if the source is FINAL (completely below the completed output
//     position E_pos) - from the ring when the source is younger than the ring guarantees, from the block's flushed
//     output in HBM otherwise.  Matches whose source is not final yet are queued for the chain wave (<= 64 per slot).
//   * ONE "chain" wave executes the queued matches strictly in order.  It is the only wave that ever waits for data,
//     it reads and writes the LDS only (in order, so a copy sees every earlier copy), and it publishes E_pos.
//   * ONE "flush" wave writes completed windows to HBM with aligned 16-byte stores and publishes F_win.
// All irregular, byte-granular accesses stay in the LDS; HBM sees the coalesced stream reads of the pre waves, the far
// match gathers and the 16-byte flush stores.  Ring reuse: window w overwrites window w - kRW, which has been flushed
// (F_win) and which no reader may touch any more because readers never reach further back than kRW - kAhead - 1 windows
// through the ring.  Every wait is bounded; a wave that waits too long aborts the block to the exact kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"

using namespace lz4par;

namespace {

constexpr int kNP    = 4;                     // pre waves
constexpr int kRW    = 32;                    // windows in the ring
constexpr int kRing  = kRW * kWin;
constexpr uint32_t kRM = kRing - 1;
constexpr int kAhead = 8;                     // windows a pre wave may be ahead of the completed output
constexpr int kCB    = 2048;                  // bytes of the stream a pre wave stages per window
constexpr int kNSP   = 3;                     // queue slots per pre wave
constexpr int kShort = 32;                    // pieces up to this length are copied by their own lane
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr int kXT    = 64 * (kNP + 2);

struct NearSlot {
    uint32_t n, epos, last, pad;
    uint32_t dst[64];
    uint32_t ol[64];                          // offset | length << 16
};
struct XSync {
    uint32_t E_pos;                           // (shifted) output position below which everything is final
    uint32_t E_win;                           // windows completed
    uint32_t F_win;                           // windows flushed to HBM
    uint32_t abort;
    uint32_t ready[kNP];                      // slots published by pre wave p
    uint32_t consumed[kNP];                   // slots the chain wave is done with
};

__device__ __forceinline__ uint32_t ldv(const uint32_t* p) { return *reinterpret_cast<const volatile uint32_t*>(p); }
__device__ __forceinline__ void stv(uint32_t* p, uint32_t v) { *reinterpret_cast<volatile uint32_t*>(p) = v; }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t xdpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t xscan_add(uint32_t v)
{
    v += xdpp0<0x111, 0xf>(v); v += xdpp0<0x112, 0xf>(v); v += xdpp0<0x114, 0xf>(v); v += xdpp0<0x118, 0xf>(v);
    v += xdpp0<0x142, 0xa>(v); v += xdpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }

typedef volatile uint8_t* ring_t;

// profiling build (make prof, tools/k1x_prof.py): per wave, cycles spent per section; PT(i) charges the time since the
// previous mark to counter i
#ifdef K1X_PROF
struct Prof { unsigned long long t[8]; unsigned long long last; };
#define PROF_DECL Prof prof_; for (int i_ = 0; i_ < 8; i_++) prof_.t[i_] = 0; prof_.last = __builtin_readcyclecounter();
#define PT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); prof_.t[i] += n_ - prof_.last; prof_.last = n_; } while (0)
#define PADD(i, v) do { prof_.t[i] += (v); } while (0)
#define PROF_OUT(B, wave, lane) do { if ((lane) == 0) { unsigned long long* d_ = (B).dbg + 8 * (wave); for (int i_ = 0; i_ < 8; i_++) d_[i_] = prof_.t[i_]; } } while (0)
#else
#define PROF_DECL
#define PT(i) do {} while (0)
#define PADD(i, v) do {} while (0)
#define PROF_OUT(B, wave, lane) do {} while (0)
#endif

// one lane, n <= kShort bytes inside the ring, LZ4 (byte-serial) semantics: dst[k] = dst[k - off]
__device__ __forceinline__ void lane_copy_ring(ring_t ring, uint32_t dst, uint32_t off, uint32_t n)
{
    uint32_t k = 0;
    if (off >= 4) {
        for (; k + 4 <= n; k += 4) {
            const uint8_t a = ring[(dst - off + k) & kRM], b = ring[(dst - off + k + 1) & kRM];
            const uint8_t c = ring[(dst - off + k + 2) & kRM], d = ring[(dst - off + k + 3) & kRM];
            ring[(dst + k) & kRM] = a; ring[(dst + k + 1) & kRM] = b; ring[(dst + k + 2) & kRM] = c; ring[(dst + k + 3) & kRM] = d;
        }
    }
    for (; k < n; k++) { const uint8_t a = ring[(dst - off + k) & kRM]; ring[(dst + k) & kRM] = a; }
}

// whole wave, one piece of any length inside the ring (wave-uniform arguments)
__device__ __forceinline__ void wave_copy_ring(ring_t ring, uint32_t dst, uint32_t off, uint32_t n, int lane)
{
    if (off >= 64) {
        for (uint32_t k = lane; k < n; k += 64) { const uint8_t a = ring[(dst - off + k) & kRM]; ring[(dst + k) & kRM] = a; }
        return;
    }
    // overlapping: the output is periodic; D = the smallest multiple of off that is >= 64 keeps every later step a plain copy
    const uint32_t D = off * ((63u + off) / off);
    for (uint32_t k = lane; k < min(n, D); k += 64) { const uint8_t a = ring[(dst - off + (k % off)) & kRM]; ring[(dst + k) & kRM] = a; }
    for (uint32_t k0 = D; k0 < n; k0 += 64) {
        const uint32_t k = k0 + lane;
        if (k < n) { const uint8_t a = ring[(dst + k - D) & kRM]; ring[(dst + k) & kRM] = a; }
    }
}

// the chain wave: matches of one slot, in order.  Entries are sorted by destination and disjoint; every byte below the
// first entry that is not one of the entries is final.
__device__ __forceinline__ uint32_t chain_slot(ring_t ring, const NearSlot* s, uint32_t n, int lane)
{
    uint32_t rounds = 0;
    const bool act = uint32_t(lane) < n;
    const uint32_t dst = act ? ldv(&s->dst[lane]) : 0xffffffffu;
    const uint32_t ol = act ? ldv(&s->ol[lane]) : 0;
    const uint32_t off = ol & 0xffff, len = ol >> 16;
    const uint32_t hi = min(dst - off + len, dst);              // end of the part of the source that others produce
    unsigned long long undone = __ballot(act);
    while (undone) {
        const int f = __builtin_ctzll(undone);
        const uint32_t Df = rl(dst, f);
        const bool mine = ((undone >> lane) & 1) != 0;
        const bool ready = mine && (hi <= Df || lane == f);     // nothing that is still missing lies below Df
        if (ready && len <= uint32_t(kShort)) lane_copy_ring(ring, dst, off, len);
        unsigned long long lg = __ballot(ready && len > uint32_t(kShort));
        while (lg) {
            const int l = __builtin_ctzll(lg); lg &= lg - 1;
            wave_copy_ring(ring, rl(dst, l), rl(off, l), rl(len, l), lane);
        }
        undone &= ~__ballot(ready);
        rounds++;
    }
    return rounds;
}

struct Blk {
    const uint8_t* src; uint8_t* dst; const uint4* wdesc; const uint32_t* tok; unsigned long long* dbg;
    uint32_t iend, nseq, total, nwin, a0;
};

// stream byte at position p: from the wave's staged copy [cs, cs + kCB) or from HBM
__device__ __forceinline__ uint32_t sbyte(const Blk& B, const uint8_t* cbuf, uint32_t cs, uint32_t p)
{
    const uint32_t i = p - cs;
    if (i < uint32_t(kCB)) return cbuf[i];
    return B.src[p];
}

__device__ __forceinline__ bool spin_fail(XSync* sy, uint32_t& spins)
{
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit) { stv(&sy->abort, 1); return true; }
    return ldv(&sy->abort) != 0;
}

// ------------------------------------------------------------------------------------------------ pre wave
__device__ void pre_wave(const Blk& B, ring_t ring, uint8_t* cbuf, NearSlot* slots, XSync* sy, int pw, int lane)
{
    uint32_t produced = 0;
    const uint32_t endp = B.total + B.a0;
    PROF_DECL

    // prefetch registers: descriptors two windows ahead, stream bytes and token positions one window ahead
    auto load_desc = [&](uint32_t w) -> uint4 {
        const uint32_t i = min(w + uint32_t(lane & 1), B.nwin);
        return B.wdesc[i];
    };
    struct Data { uint4 c[kCB / 1024]; uint32_t t[3]; };
    auto load_data = [&](const uint4& dl, Data& d) {
        const uint32_t first = rl(dl.x, 0), cs = rl(dl.z, 0);
#pragma unroll
        for (int q = 0; q < kCB / 1024; q++) {
            const uint32_t g = cs + 1024u * q + 16u * lane;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g + 16 <= B.iend) v = ld16u(B.src + g);
            else if (g < B.iend) {
                uint32_t wv[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < B.iend - g; i++) wv[i >> 2] |= uint32_t(B.src[g + i]) << (8 * (i & 3));
                v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
            d.c[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 3; q++) { const uint32_t i = first + 64u * q + lane; d.t[q] = i < B.nseq ? B.tok[i] : 0; }
    };

    uint32_t w = pw;
    if (w >= B.nwin) { PROF_OUT(B, pw, lane); return; }
    uint4 dcur = load_desc(w);
    uint4 dnext = load_desc(w + kNP);
    Data cur; load_data(dcur, cur);
    Data nxt;
    for (; w < B.nwin; w += kNP) {
        const bool have_next = w + kNP < B.nwin;
        uint4 dnn = dnext;
        if (have_next) { load_data(dnext, nxt); dnn = load_desc(w + 2 * kNP); }
        const uint32_t first = rl(dcur.x, 0), opos0 = rl(dcur.y, 0), cs = rl(dcur.z, 0);
        const uint32_t last = min(rl(dcur.x, 1), B.nseq - 1);
        const uint32_t W0 = w << kWinLog, W1 = min(W0 + uint32_t(kWin), endp);
        const int lbw = int(w) + kAhead - kRW + 1;
        const uint32_t lowb = lbw > 0 ? uint32_t(lbw) << kWinLog : 0u;     // the ring is guaranteed from here on
        PT(1);
        // may this window be produced yet?
        for (uint32_t spins = 0; ldv(&sy->E_win) + kAhead < w || ldv(&sy->F_win) + kRW <= w; ) if (spin_fail(sy, spins)) return;
        cbar();
        PT(0);
        // stage the stream
#pragma unroll
        for (int q = 0; q < kCB / 1024; q++) *reinterpret_cast<uint4*>(cbuf + 1024 * q + 16 * lane) = cur.c[q];
        uint32_t obase = opos0;
        for (uint32_t j0 = 0; first + j0 <= last; j0 += 64) {
            const uint32_t sidx = first + j0 + lane;
            const bool act = sidx <= last;
            uint32_t tp = j0 == 0 ? cur.t[0] : j0 == 64 ? cur.t[1] : j0 == 128 ? cur.t[2] : (act ? B.tok[sidx] : 0u);
            // ---- decode (every rule was checked by the parser)
            uint32_t ll = 0, ml = 0, off = 0, litpos = 0;
            if (act) {
                const uint32_t tk = sbyte(B, cbuf, cs, tp);
                uint32_t p = tp + 1;
                ll = tk >> 4;
                if (ll == 15) { uint32_t bb; do { bb = sbyte(B, cbuf, cs, p); p++; ll += bb; } while (bb == 255); }
                litpos = p;
                if (sidx != B.nseq - 1) {
                    p += ll;
                    off = sbyte(B, cbuf, cs, p) | (sbyte(B, cbuf, cs, p + 1) << 8);
                    p += 2;
                    ml = tk & 15;
                    if (ml == 15) { uint32_t bb; do { bb = sbyte(B, cbuf, cs, p); p++; ml += bb; } while (bb == 255); }
                    ml += 4;
                }
            }
            const uint32_t len = ll + ml;
            const uint32_t incl = xscan_add(len);
            const uint32_t sp = obase + (incl - len) + B.a0;                 // shifted output position of the sequence
            obase += rl(incl, 63);
            // ---- wait for a free queue slot, then take the completed position once for the whole batch
            PT(1);
            for (uint32_t spins = 0; produced >= ldv(&sy->consumed[pw]) + kNSP; ) if (spin_fail(sy, spins)) return;
            const uint32_t E_pos = ldv(&sy->E_pos);
            cbar();
            PT(2);
            // ---- literals
            {
                const uint32_t ls = max(sp, W0), le = min(sp + ll, W1);
                const uint32_t n = (act && le > ls) ? le - ls : 0;
                const uint32_t cp = litpos + (ls - sp);
                if (n && n <= uint32_t(kShort)) for (uint32_t k = 0; k < n; k++) ring[(ls + k) & kRM] = uint8_t(sbyte(B, cbuf, cs, cp + k));
                unsigned long long lg = __ballot(n > uint32_t(kShort));
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    const uint32_t d0 = rl(ls, l), nn = rl(n, l), c0 = rl(cp, l);
                    for (uint32_t k = lane; k < nn; k += 64) ring[(d0 + k) & kRM] = uint8_t(sbyte(B, cbuf, cs, c0 + k));
                }
            }
            PT(3);
            // ---- match
            const uint32_t mstart = sp + ll;
            const uint32_t ds = max(mstart, W0), de = min(mstart + ml, W1);
            const uint32_t mn = (act && ml && de > ds) ? de - ds : 0;
            const uint32_t s0 = ds - off;
            const uint32_t hi = min(s0 + mn, ds);
            const bool fin = mn && hi <= E_pos;
            const bool in_ring = fin && s0 >= lowb;
            const bool in_hbm = fin && !in_ring;
            const bool near = mn && !fin;
            if (in_ring && mn <= uint32_t(kShort)) lane_copy_ring(ring, ds, off, mn);
            {
                unsigned long long lg = __ballot(in_ring && mn > uint32_t(kShort));
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    wave_copy_ring(ring, rl(ds, l), rl(off, l), rl(mn, l), lane);
                }
            }
            PT(4);
            if (__ballot(in_hbm)) {
                // the source left the ring's guaranteed part: read it from the flushed output (never overlapping: off > mn)
                const uint32_t need = in_hbm ? ((s0 + mn - 1) >> kWinLog) + 1 : 0;
                uint32_t need_all = need;
                for (int o = 32; o; o >>= 1) need_all = max(need_all, uint32_t(__shfl_xor(int(need_all), o)));
                for (uint32_t spins = 0; ldv(&sy->F_win) < need_all; ) if (spin_fail(sy, spins)) return;
                cbar();
                const uint8_t* g = B.dst + s0 - B.a0;
                if (in_hbm && mn <= uint32_t(kShort)) for (uint32_t k = 0; k < mn; k++) ring[(ds + k) & kRM] = g[k];
                unsigned long long lg = __ballot(in_hbm && mn > uint32_t(kShort));
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    const uint32_t d0 = rl(ds, l), nn = rl(mn, l), ss = rl(s0, l);
                    const uint8_t* gg = B.dst + ss - B.a0;
                    for (uint32_t k = lane; k < nn; k += 64) ring[(d0 + k) & kRM] = gg[k];
                }
            }
            PT(5);
            // ---- queue what is left for the chain wave
            NearSlot* s = slots + (produced % kNSP);
            const unsigned long long nb = __ballot(near);
            if (near) {
                const uint32_t idx = __builtin_amdgcn_mbcnt_hi(uint32_t(nb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(nb), 0));
                stv(&s->dst[idx], ds); stv(&s->ol[idx], off | (mn << 16));
            }
            const bool lastb = first + j0 + 64 > last;
            if (lane == 0) {
                stv(&s->n, uint32_t(__builtin_popcountll(nb)));
                stv(&s->epos, lastb ? W1 : min(W1, obase + B.a0));
                stv(&s->last, lastb ? 1u : 0u);
            }
            lds_fence();                                  // every byte of the batch is in the LDS before the slot is published
            produced++;
            if (lane == 0) stv(&sy->ready[pw], produced);
            PT(6);
            PADD(7, uint64_t(__builtin_popcountll(nb)));
        }
        dcur = dnext; dnext = dnn; cur = nxt;
    }
    PROF_OUT(B, pw, lane);
}

// ------------------------------------------------------------------------------------------------ chain wave
__device__ void chain_wave(const Blk& B, ring_t ring, NearSlot* slots, XSync* sy, int lane)
{
    uint32_t cons[kNP];
#pragma unroll
    for (int i = 0; i < kNP; i++) cons[i] = 0;
    __builtin_amdgcn_s_setprio(3);
    PROF_DECL
    for (uint32_t w0 = 0; w0 < B.nwin; w0 += kNP) {
#pragma unroll
        for (int i = 0; i < kNP; i++) {
            const uint32_t w = w0 + i;
            if (w >= B.nwin) break;
            for (;;) {
                for (uint32_t spins = 0; ldv(&sy->ready[i]) <= cons[i]; ) if (spin_fail(sy, spins)) return;
                cbar();
                PT(0);
                NearSlot* s = slots + i * kNSP + (cons[i] % kNSP);
                const uint32_t n = ldv(&s->n), epos = ldv(&s->epos), last = ldv(&s->last);
                if (n) { const uint32_t rounds = chain_slot(ring, s, n, lane); PADD(3, rounds); PADD(4, n); }
                PADD(5, 1);
                lds_fence();
                PT(1);
                cons[i]++;
                if (lane == 0) { stv(&sy->E_pos, epos); stv(&sy->consumed[i], cons[i]); if (last) stv(&sy->E_win, w + 1); }
                PT(2);
                if (last) break;
            }
        }
    }
    PROF_OUT(B, kNP, lane);
}

// ------------------------------------------------------------------------------------------------ flush wave
__device__ void flush_wave(const Blk& B, ring_t ring, XSync* sy, int lane)
{
    const uint32_t endp = B.total + B.a0;
    constexpr int kLag = 6;                    // stores in flight before the oldest one is waited for
    PROF_DECL
    for (uint32_t f = 0; f < B.nwin; f++) {
        PT(1);
        for (uint32_t spins = 0; ldv(&sy->E_win) <= f; ) if (spin_fail(sy, spins)) return;
        cbar();
        PT(0);
        const uint32_t p0 = (f << kWinLog) + 16u * lane;
        const uint4 v = *reinterpret_cast<const uint4*>(const_cast<const uint8_t*>(ring) + (p0 & kRM));   // behind the barrier above
        uint8_t* g = B.dst + p0 - B.a0;        // 16-byte aligned by construction of a0
        if (p0 >= B.a0 && p0 + 16 <= endp) *reinterpret_cast<uint4*>(g) = v;
        else {
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t k = 0; k < 16; k++) if (p0 + k >= B.a0 && p0 + k < endp) g[k] = uint8_t(wv[k >> 2] >> (8 * (k & 3)));
        }
        if (f == 0 || f + 1 == B.nwin) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) stv(&sy->F_win, f + 1); }
        else { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); if (lane == 0 && f >= uint32_t(kLag)) stv(&sy->F_win, f + 1 - kLag); }
    }
    PT(1);
    PROF_OUT(B, kNP + 1, lane);
}

} // namespace

__global__ __launch_bounds__(kXT)
void lz4_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                     const uint8_t* work)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[kNP][kCB];
    __shared__ NearSlot slots[kNP][kNSP];
    __shared__ XSync sy;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint8_t* slot = work + size_t(b) * kSlotBytes;
    const ParHdr* hdr = reinterpret_cast<const ParHdr*>(slot);
    if (hdr->status != kParsed) return;
    const fourmc_block blk = blocks[b];
    Blk B;
    B.src = src_base + blk.src_off; B.dst = dst_base + blk.dst_off;
    B.wdesc = reinterpret_cast<const uint4*>(slot + kWdescOff);
    B.tok = reinterpret_cast<const uint32_t*>(slot + kTokOff);
    B.dbg = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(slot) + kDbgOff);
    B.iend = blk.src_len; B.nseq = hdr->nseq; B.total = hdr->total; B.nwin = hdr->nwin; B.a0 = hdr->a0;
    if (threadIdx.x == 0) {
        sy.E_pos = 0; sy.E_win = 0; sy.F_win = 0; sy.abort = 0;
        for (int i = 0; i < kNP; i++) { sy.ready[i] = 0; sy.consumed[i] = 0; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (wave < kNP) pre_wave(B, ring, cbuf[wave], slots[wave], &sy, wave, lane);
    else if (wave == kNP) chain_wave(B, ring, &slots[0][0], &sy, lane);
    else {
        flush_wave(B, ring, &sy, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) blocks[b].result = ldv(&sy.abort) ? kRetryCode : int(B.total);
    }
}

extern "C" hipError_t fourmc_launch_lz4_exec(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                              const void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_exec_kernel, dim3(n), dim3(kXT), 0, stream, static_cast<const uint8_t*>(d_src),
                       static_cast<uint8_t*>(d_dst), d_blocks, n, static_cast<const uint8_t*>(d_work));
    return hipGetLastError();
}
