// 4mc_amd/csrc/lz4_decode.hip — K1: batched LZ4 block decode on gfx950 (wave64).
//
// Replaces the per-block call LZ4_decompress_safe(in, out, csize, usize) of the reference
// (native/4mc.c:661, native/jniDecompressor.c:88 -> native/lz4/lz4.c:2345-2350 -> :1936-2339).
//
// "exact" kernel: one wavefront walks one block's sequences in order and reproduces the
// reference's accept/reject set and negative return codes (including the wider acceptance of its
// x86-64 fast loop, lz4.c:1995-2110), so results are bit-identical on valid AND corrupt input.
//
// Data movement per block (HBM-bound byte work, no MFMA):
//   * compressed stream: 16 B/lane coalesced loads, one 1 KiB granule ahead of use, staged in a
//     4 KiB LDS ring; the parser sees it through a 64-byte per-lane lookahead register (`la`),
//     so token / length / offset bytes are wave-uniform v_readlane reads, not memory round trips;
//   * literals: stored straight from the lookahead register (lane j owns stream byte la_pos+j);
//   * matches: 64 bytes per step, read back from the block's own output (L2-resident, <=64 KiB
//     behind the write cursor); overlapping matches (offset < length) are expanded from the
//     period so that every step is a full-width copy instead of a byte-serial chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"
#include "lz4seg.h"
#include "lz4tile.h"
#include <string.h>
#include <atomic>

namespace {

constexpr int kRing  = 4096;   // LDS bytes of compressed-stream ring per wave
constexpr int kChunk = 1024;   // refill granule: 64 lanes x 16 B
constexpr int kAhead = 2048;   // keep this much of the stream staged beyond the read cursor

struct Stream {
    const uint8_t* abase;   // 16-byte aligned address at or below the block's first byte
    int      delta;         // first byte - abase            (0..15)
    int      qend;          // delta + csize  (end of the stream in aligned coordinates)
    int      fill_hi;       // ring holds aligned positions [.., fill_hi); multiple of kChunk
    uint8_t* ring;          // LDS
    uint4    pend;          // granule [fill_hi, fill_hi + kChunk) already requested from HBM
    uint32_t la;            // lookahead: lane j holds stream byte la_pos + j
    int      la_pos;
    int      lane;

    __device__ __forceinline__ uint4 fetch(int q) const {
        // aligned 16 B granules are safe to read whenever they contain at least one stream byte
        const int g = q + 16 * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < qend) v = *reinterpret_cast<const uint4*>(abase + g);
        return v;
    }
    __device__ __forceinline__ void advance() {
        *reinterpret_cast<uint4*>(ring + ((fill_hi + 16 * lane) & (kRing - 1))) = pend;
        fill_hi += kChunk;
        pend = fetch(fill_hi);
    }
    __device__ __forceinline__ void init(const uint8_t* src, int csize, uint8_t* lds, int ln) {
        // pointer arithmetic (not an integer round trip) keeps the address space known: global_load,
        // not flat_load - a pending FLAT access would force every later s_waitcnt to vmcnt(0)
        delta = int(reinterpret_cast<uintptr_t>(src) & 15);
        abase = src - delta;
        qend = delta + csize;
        ring = lds; lane = ln;
        fill_hi = 0;
        pend = fetch(0);
        la_pos = -(1 << 30);
        la = 0;
    }
    // load the lookahead register so that lane 0 sits on stream position p
    __device__ __forceinline__ void reload(int p) {
        const int q = p + delta;
        if (q >= fill_hi + kChunk) {               // jumped over staged data (long literal run)
            fill_hi = q & ~(kChunk - 1);
            pend = fetch(fill_hi);
        }
        while (fill_hi < q + kAhead && fill_hi < qend) advance();
        la = ring[(q + lane) & (kRing - 1)];
        la_pos = p;
    }
    __device__ __forceinline__ uint32_t get(int p) {      // p is wave-uniform
        if (p < la_pos || p >= la_pos + 64) reload(p);
        return (uint32_t)__builtin_amdgcn_readlane((int)la, p - la_pos);
    }
};

// Length continuation bytes (lz4.c:1903-1928).  On failure returns false with ip = position the
// reference reports.  Consumes whole runs of 255 per step via a ballot over the lookahead.
__device__ __forceinline__ bool more_len(Stream& s, int& ip, int lim, bool check_first, int& len)
{
    if (check_first && ip >= lim) return false;
    for (;;) {
        if (ip < s.la_pos || ip >= s.la_pos + 64) s.reload(ip);
        const int l0 = ip - s.la_pos;
        const unsigned long long not255 = __ballot(s.la != 255u) >> l0;
        const int avail = 64 - l0;
        int n, add; bool done;
        if (not255 == 0) { n = avail; add = 255 * avail; done = false; }
        else {
            const int t = __builtin_ctzll(not255);
            n = t + 1;
            add = 255 * t + __builtin_amdgcn_readlane((int)s.la, l0 + t);
            done = true;
        }
        if (ip + n > lim) { ip = (ip + 1 > lim + 1) ? ip + 1 : lim + 1; return false; }
        ip += n;
        len = len > 0x40000000 - add ? 0x40000000 : len + add;     // saturates far above any buffer: every later bound check fails as the reference's pointer-wrap tests do (lz4.c:1903-1928)
        if (done) return true;
    }
}

__device__ __forceinline__ void copy_literals(Stream& s, const uint8_t* src, uint8_t* dst,
                                              int ip, int op, int n)
{
    int done = 0;
    if (ip >= s.la_pos && ip < s.la_pos + 64) {
        const int n1 = min(n, s.la_pos + 64 - ip);
        const int p = s.la_pos + s.lane;
        if (p >= ip && p < ip + n1) dst[op + (p - ip)] = (uint8_t)s.la;
        done = n1;
    }
    // long runs: straight HBM -> HBM
    if (done < n) wave_copy(dst + op + done, src + ip + done, n - done, s.lane);
}

// One wave decodes one block.  Mirrors the control flow restated in oracle/lz4_port.c.
// (ip0, op0) != (0, 0): resume behind sequences another kernel has executed (lz4_seg.hip) - only ever at a token the reference's
// fast loop would still be in (at least kMargin stream bytes and kOMargin output bytes from the ends, lz4seg.h)
__device__ __forceinline__ int lz4_decode_block(const uint8_t* src, int csize, uint8_t* dst, int cap,
                                uint8_t* lds, int lane, int ip0 = 0, int op0 = 0)
{
    if (cap < 0) return -1;
    if (cap == 0) return (csize == 1 && src[0] == 0) ? 0 : -1;          // lz4.c:1977-1981
    if (csize == 0) return -1;

    Stream s; s.init(src, csize, lds, lane);
    const int iend = csize, oend = cap;
    int ip = ip0, op = op0;
    bool fast = (oend - op) >= 64;                                      // lz4.c:1990

    for (;;) {
        if (ip < s.la_pos || ip + 24 > s.la_pos + 64) s.reload(ip);
        const uint32_t token = s.get(ip); ip++;
        int lit = int(token >> 4);
        int mlen = int(token & 15);
        bool shortcut = false;

        if (lit == 15) { if (!more_len(s, ip, iend - 15, true, lit)) return -ip - 1; }

        bool check_end;   // does the literal run have to pass the end-of-block test?
        if (fast) {
            check_end = (token >> 4) == 15 ? (op + lit > oend - 32 || ip + lit > iend - 32)
                                           : (ip > iend - 17);
            if (check_end) fast = false;
        } else {
            shortcut = (token >> 4) != 15 && ip < iend - 16 && op <= oend - 32;   // :2128
            check_end = !shortcut;
        }
        if (check_end && (op + lit > oend - 12 || ip + lit > iend - 8)) {
            // terminating literal run (lz4.c:2175-2225)
            if (ip + lit != iend || op + lit > oend) return -ip - 1;
            copy_literals(s, src, dst, ip, op, lit);
            return op + lit;
        }
        copy_literals(s, src, dst, ip, op, lit);
        ip += lit; op += lit;

        const int off = int(s.get(ip)) | (int(s.get(ip + 1)) << 8);
        ip += 2;
        const int from = op - off;

        if (fast) {
            if (mlen == 15) {
                if (!more_len(s, ip, iend - 4, false, mlen)) return -ip - 1;
                mlen += 4;
                if (from < 0) return -ip - 1;
                if (op + mlen >= oend - 64) fast = false;
            } else {
                mlen += 4;
                if (op + mlen >= oend - 64) fast = false;
                else if (from < 0) return -ip - 1;
            }
            if (!fast) {
                if (from < 0) return -ip - 1;
                if (op + mlen > oend - 5) return -ip - 1;
            }
        } else if (shortcut && mlen != 15 && off >= 8 && from >= 0) {
            mlen += 4;                                                  // lz4.c:2143-2152
        } else {
            if (mlen == 15) { if (!more_len(s, ip, iend - 4, false, mlen)) return -ip - 1; }
            mlen += 4;
            if (from < 0) return -ip - 1;                               // lz4.c:2250
            if (op + mlen > oend - 5) return -ip - 1;                   // lz4.c:2317
        }
        if (off == 0) return INT32_MIN;     // undefined in the reference (reads unwritten output)
        copy_match(dst, op, off, mlen, lane);
        op += mlen;
    }
}

// container_mode = 0: raw codec call, result = LZ4_decompress_safe's return value.
// container_mode = 1: one iteration of decodeFourMC's loop (native/4mc.c:603-668) after the
//   checksum pass: blocks flagged FOURMC_BLK_BADSUM are skipped, src_len == dst_cap means a
//   stored block (plain copy, :635-642), a negative codec result becomes FOURMC_BLK_CORRUPT (:662).
__global__ __launch_bounds__(64)
void lz4_decode_exact_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                             fourmc_block* blocks, uint32_t nblocks, int container_mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    int r;
    if (container_mode) {
        if (blk.result == FOURMC_BLK_BADSUM) return;
        if (blk.src_len == blk.dst_cap) {
            wave_copy(dst, src, int(blk.src_len), threadIdx.x);
            r = int(blk.src_len);
        } else {
            r = lz4_decode_block(src, int(blk.src_len), dst, int(blk.dst_cap), ring, threadIdx.x);
            if (r < 0) r = FOURMC_BLK_CORRUPT;
        }
    } else {
        r = lz4_decode_block(src, int(blk.src_len), dst, int(blk.dst_cap), ring, threadIdx.x);
    }
    if (threadIdx.x == 0) blocks[b].result = r;
}


// ================================================================================================
// Fast path ("batch" decoder).  Same block format, same results on every stream it accepts; any
// irregularity (rule violation, offset beyond the produced output, odd end-of-block shape) makes it
// return kRetry and the exact kernel above redoes the block, so error codes stay the reference's.
//
// Instead of one sequence per step it takes a 64-byte window of the compressed stream and
//   1. treats EVERY byte as a candidate token in parallel (lane j: literal count, next-token slot),
//   2. walks the chain of real tokens with v_readlane only (no memory in the serial part),
//   3. prefix-sums the output sizes of all sequences found (typically 6-12),
//   4. produces the batch's output 64 bytes per step in OUTPUT order: each lane finds the sequence
//      that owns its byte (LDS owner map + max-scan), literals are pulled from the window register
//      with a lane permute, match bytes whose source precedes the step are loaded from the
//      block's own output, sources inside the step are resolved by pointer jumping over lanes.
// Long literal runs / long matches (length nibble 15) and the block tail take the one-sequence
// path, which is already 64 bytes wide per step.
constexpr int kRetry = -1000000003;      // internal: "let the exact kernel decide"
constexpr int kOwnBytes = 4096;          // cap on the bytes one batch may produce (owner map size)

// Wave64 inclusive scans on the VALU cross-lane network (DPP row shifts + row broadcasts, no LDS):
// 4 row_shr steps scan each row of 16 lanes, row_bcast15 / row_bcast31 carry the row totals.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }

__device__ __forceinline__ uint32_t scan_add(uint32_t v, int)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v);            // row_bcast15 -> rows 1,3
    v += dpp0<0x143, 0xc>(v);            // row_bcast31 -> rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t scan_max(uint32_t v, int)      // values are >= 0, identity 0
{
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v));
    v = max(v, dpp0<0x143, 0xc>(v));
    return v;
}

// The fast path runs on 1 + kCopiers wavefronts per block: what the tokens say (lengths, offsets, where each sequence's
// output starts) depends on the compressed stream alone, so a PARSER wave walks the stream and hands finished records
// (a batch of sequences, or one general sequence) to COPIER waves through a ring in LDS; record k belongs to copier
// k % kCopiers.  The output ranges of records are disjoint, so copiers only meet where a match reads what a record still
// in flight on another copier produces: the parser works out, per record, the youngest earlier record its sources touch
// (`need`), and a copier starts record k once every record <= need is complete (done[] counters, release / acquire at
// workgroup scope; the waves of a workgroup share the CU's L1, so completed stores are visible to plain loads).
constexpr int kCopiers = 2;
constexpr int kRec = 8;
enum : uint32_t { kRecBatch = 1, kRecGeneral = 2, kRecEnd = 3, kRecRetry = 4 };
struct Rec {
    uint32_t type;
    uint32_t T;              // batch: output bytes; general: match length (0: the block's last, literal-only sequence)
    uint32_t op;             // output position where the record starts (end: the decoded size)
    uint32_t lit, lit_ip, off;      // general sequence
    int      need;           // every record with an index <= need has to be complete before this one reads the output
    unsigned long long tokmask;     // batch: lanes (window slots) that are tokens
    uint32_t pack[64];       // batch, per token lane: output start in the batch | literals << 13 | (token header - 1) << 19
    uint32_t offb[64];       // batch: match offset | this slot's stream byte << 16
};
struct DSync {
    uint32_t produced;               // records published
    uint32_t total;                  // number of records of the block, once the parser has stopped (else 0xFFFFFFFF)
    uint32_t consumed[kCopiers];     // copier w has taken every record of its own below this index into registers
    int      done[kCopiers];         // copier w has completed every record of its own below this index
    uint32_t failed;                 // someone gave up: the retry kernel decides
    int      end_value;              // the End record's decoded size
};
__device__ __forceinline__ uint32_t ld_acq(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ld_acq(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_rel(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_rel(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
constexpr uint32_t kSpinLimit = 1u << 24;    // a wait that long means the other wave is gone: give up (-> retry kernel) instead of hanging

// PARSER wave: same acceptance rules as before, no output access at all
__device__ void lz4_fast_parse(const uint8_t* src, int csize, int cap, uint8_t* lds, Rec* recs, DSync* sy, int lane)
{
    uint32_t k = 0;                                                     // records published
    uint32_t op_km1 = 0, op_km2 = 0;                                    // where the two previous records start
    auto stop = [&](bool failed) { if (lane == 0) { if (failed) sy->failed = 1; st_rel(&sy->total, k); } };
    auto slot = [&]() -> Rec* {                                         // next record, once its copier has freed it
        if (k >= uint32_t(kRec)) {
            uint32_t* c = &sy->consumed[(k - kRec) % kCopiers];
            for (uint32_t spins = 0; ld_acq(c) <= k - kRec; ) { __builtin_amdgcn_s_sleep(1); if (++spins > kSpinLimit) { stop(true); return nullptr; } }
        }
        return recs + (k % kRec);
    };
    auto publish = [&](uint32_t at) { if (lane == 0) st_rel(&sy->produced, k + 1); k++; op_km2 = op_km1; op_km1 = at; };
    // the youngest earlier record whose output [its start, this record's start) a source range ending at `reach` touches
    auto need_of = [&](uint32_t reach) -> int {
        if (reach > op_km1) return int(k) - 1;
        if (kCopiers >= 3 && reach > op_km2) return int(k) - 2;
        return int(k) - kCopiers;
    };
    auto finish = [&](uint32_t type, int value) {
        Rec* r = slot(); if (!r) return;
        if (lane == 0) { r->type = type; r->op = uint32_t(value); r->need = int(k) - kCopiers; }
        publish(uint32_t(value)); stop(false);
    };
    if (cap < 64 || csize < 1) { finish(kRecRetry, 0); return; }
    Stream s; s.init(src, csize, lds, lane);
    const int iend = csize, oend = cap;
    int ip = 0, op = 0;
    for (;;) {
        ip = __builtin_amdgcn_readfirstlane(ip); op = __builtin_amdgcn_readfirstlane(op); k = uint32_t(__builtin_amdgcn_readfirstlane(int(k)));
        op_km1 = uint32_t(__builtin_amdgcn_readfirstlane(int(op_km1))); op_km2 = uint32_t(__builtin_amdgcn_readfirstlane(int(op_km2)));
        s.fill_hi = __builtin_amdgcn_readfirstlane(s.fill_hi); s.la_pos = __builtin_amdgcn_readfirstlane(s.la_pos);
        // ---------------------------------------------------------------- batch of sequences inside one window
        if (ip + 64 + 16 <= iend && op + kOwnBytes + 64 + 16 <= oend) {
            // w: 4 consecutive stream bytes per lane (lane j = bytes ip+j .. ip+j+3)
            s.reload(ip);
            const int q0 = ip + s.delta + lane;
            const uint32_t w = s.la | (uint32_t(s.ring[(q0 + 1) & (kRing - 1)]) << 8) |
                               (uint32_t(s.ring[(q0 + 2) & (kRing - 1)]) << 16) | (uint32_t(s.ring[(q0 + 3) & (kRing - 1)]) << 24);
            const uint32_t b = w & 0xff, b1 = (w >> 8) & 0xff;
            const uint32_t L0 = b >> 4, M0 = b & 15;
            // literal count: nibble, or 15 + ONE continuation byte (longer runs take the general path)
            const uint32_t L = (L0 == 15) ? 15 + b1 : L0;
            const uint32_t lhdr = (L0 == 15) ? 2 : 1;                   // token (+ continuation byte)
            const uint32_t offpos = uint32_t(lane) + lhdr + L;         // window slot of the offset's low byte
            const uint32_t wo = __shfl(w, offpos & 63);                 // offset lo, hi, first match continuation byte
            const uint32_t e1 = (wo >> 16) & 0xff;
            const uint32_t ml = (M0 == 15) ? 19 + e1 : M0 + 4;
            const uint32_t nxt = offpos + 2 + (M0 == 15 ? 1 : 0);
            const bool ok = (L0 != 15 || b1 != 255) && (M0 != 15 || e1 != 255) && nxt <= 64;
            // Token chain over a per-lane jump table, without a branch per token: a token never starts in slots 62 / 63 (it needs
            // three bytes), so slot 63 is an absorbing end state and the walk is a straight run of v_readlane + s_bitset1
            // pairs, checked for the end every 7 tokens (a window holds at most 21).
            const uint32_t jump = ok ? nxt : 128u + uint32_t(lane);     // 64: the window ends behind this token; >= 128: no batch token
            const uint32_t hop = lane == 63 ? 63u : min(jump, 63u);
            unsigned long long tokmask = 0;
            uint32_t pos = 0;
            for (int round = 0; round < 3 && pos != 63; round++) {
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    asm volatile("s_bitset1_b64 %0, %1" : "+s"(tokmask) : "s"(pos));
                    pos = uint32_t(__builtin_amdgcn_readlane(int(hop), int(pos)));
                }
            }
            tokmask &= ~(1ull << 63);
            {   // where the chain left the window: behind its last token (64), or at a slot that is no batch token (the batch ends before it)
                const uint32_t last = 63u - uint32_t(__builtin_clzll(tokmask));
                const uint32_t j = uint32_t(__builtin_amdgcn_readlane(int(jump), int(last)));
                if (j >= 128) { tokmask &= ~(1ull << last); pos = last; } else pos = j;
            }
            if (tokmask) {
                bool is_tok = (tokmask >> lane) & 1;
                uint32_t sz = is_tok ? L + ml : 0;
                uint32_t incl = scan_add(sz, lane);
                uint32_t T = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
                if (T > uint32_t(kOwnBytes)) {                           // cap the batch (first sequence always fits)
                    tokmask = __ballot(is_tok && incl <= uint32_t(kOwnBytes));
                    const int last = 63 - __builtin_clzll(tokmask);
                    pos = uint32_t(__builtin_amdgcn_readlane(int(nxt), last));
                    T = uint32_t(__builtin_amdgcn_readlane(int(incl), last));
                    is_tok = (tokmask >> lane) & 1;
                    sz = is_tok ? sz : 0;
                }
                const uint32_t ostart = incl - sz;
                const uint32_t off = wo & 0xffff;
                const uint32_t mstart = uint32_t(op) + ostart + L;       // where this sequence's match starts
                const bool bad = is_tok && (off == 0 || off > mstart);
                if (__ballot(bad)) { finish(kRecRetry, 0); return; }
                // how far the sources that lie before this record reach
                const uint32_t reach1 = (is_tok && mstart - off < uint32_t(op)) ? min(mstart - off + ml, uint32_t(op)) : 0u;
                const uint32_t reach = uint32_t(__builtin_amdgcn_readlane(int(scan_max(reach1, lane)), 63));
                Rec* r = slot();
                if (!r) return;
                r->pack[lane] = ostart | (L << 13) | ((lhdr - 1) << 19);
                r->offb[lane] = off | (b << 16);
                if (lane == 0) { r->type = kRecBatch; r->T = T; r->op = uint32_t(op); r->tokmask = tokmask; r->need = need_of(reach); }
                publish(uint32_t(op));
                op += int(T);
                ip += int(pos);
                continue;
            }
        }
        // ---------------------------------------------------------------- one general sequence (strict rules)
        if (ip >= iend) { finish(kRecRetry, 0); return; }
        if (ip < s.la_pos || ip + 24 > s.la_pos + 64) s.reload(ip);
        const uint32_t token = s.get(ip); ip++;
        int lit = int(token >> 4), mlen = int(token & 15);
        if (lit == 15) { if (!more_len(s, ip, iend - 15, true, lit)) { finish(kRecRetry, 0); return; } }
        if (op + lit > oend - 12 || ip + lit > iend - 8) {
            if (ip + lit != iend || op + lit > oend) { finish(kRecRetry, 0); return; }
            Rec* r = slot();
            if (!r) return;
            if (lane == 0) { r->type = kRecGeneral; r->T = 0; r->op = uint32_t(op); r->lit = uint32_t(lit); r->lit_ip = uint32_t(ip); r->off = 0; r->need = int(k) - kCopiers; }
            publish(uint32_t(op));
            finish(kRecEnd, op + lit);
            return;
        }
        const int lit_ip = ip;
        ip += lit;
        const int op2 = op + lit;
        const int off = int(s.get(ip)) | (int(s.get(ip + 1)) << 8);
        ip += 2;
        if (mlen == 15) { if (!more_len(s, ip, iend - 4, false, mlen)) { finish(kRecRetry, 0); return; } }
        mlen += 4;
        if (off == 0 || off > op2 || op2 + mlen > oend - 5) { finish(kRecRetry, 0); return; }
        Rec* r = slot();
        if (!r) return;
        const uint32_t reach = (op2 - off < op) ? uint32_t(min(op2 - off + mlen, op)) : 0u;
        if (lane == 0) { r->type = kRecGeneral; r->T = uint32_t(mlen); r->op = uint32_t(op); r->lit = uint32_t(lit); r->lit_ip = uint32_t(lit_ip); r->off = uint32_t(off); r->need = need_of(reach); }
        publish(uint32_t(op));
        op = op2 + mlen;
    }
}

// COPIER wave w: executes records w, w + kCopiers, ... in order
__device__ void lz4_fast_copy(const uint8_t* src, uint8_t* dst, Rec* recs, DSync* sy, uint8_t* own, int w, int lane)
{
    auto leave = [&](bool failed) {                                      // nobody may wait for this wave any more
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) { if (failed) sy->failed = 1; st_rel(&sy->done[w], 0x7FFFFFFF); st_rel(&sy->consumed[w], 0xFFFFFFFFu); }
    };
    for (uint32_t k = uint32_t(w);; k += kCopiers) {
        for (uint32_t spins = 0; ld_acq(&sy->produced) <= k; ) {
            if (ld_acq(&sy->total) <= k) { leave(false); return; }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) { leave(true); return; }
        }
        Rec* r = recs + (k % kRec);
        const uint32_t type = r->type, T = r->T;
        const int op = int(r->op), need = r->need;
        const unsigned long long tokmask = r->tokmask;
        const uint32_t pack = r->pack[lane], ob = r->offb[lane];
        const uint32_t lit = r->lit, lit_ip = r->lit_ip, goff = r->off;
        if (lane == 0) st_rel(&sy->consumed[w], k + 1);                 // everything of the record is in registers now
        if (type == kRecEnd) { if (lane == 0) sy->end_value = op; leave(false); return; }
        if (type != kRecBatch && type != kRecGeneral) { leave(true); return; }
        // the records this one reads from have to be complete
        for (int o = 0; o < kCopiers; o++) {
            if (o == w) continue;
            for (uint32_t spins = 0; ld_acq(&sy->done[o]) <= need; ) { __builtin_amdgcn_s_sleep(1); if (++spins > kSpinLimit) { leave(true); return; } }
        }
        if (type == kRecBatch) {
            const uint32_t off = ob & 0xffff, b = ob >> 16;
            const bool is_tok = (tokmask >> lane) & 1;
            // owner map: own[o] = token lane + 1 at the first output byte of each sequence
            for (uint32_t i = 4u * lane; i < T; i += 256) *reinterpret_cast<uint32_t*>(own + i) = 0;
            if (is_tok) own[pack & 0x1FFF] = uint8_t(lane + 1);
            // One 64-byte step of output is kept PENDING in registers: it is stored only after the next step's loads
            // have been issued, so a step waits for its own loads (vmcnt leaves the younger store outstanding) and never
            // for a store acknowledgement.  Sources that fall into the pending step are forwarded from its registers.
            uint32_t carry = 0, pv = 0;
            auto step = [&](uint32_t c0, auto first) {
                const uint32_t o = c0 + lane;
                const bool live = o < T;
                uint32_t m = live ? uint32_t(own[o]) : 0u;
                m = max(scan_max(m, lane), carry);
                carry = uint32_t(__builtin_amdgcn_readlane(int(m), 63));
                const int tl = int(m) - 1;                                  // owning token lane
                const uint32_t P = __shfl(pack, tl & 63);
                const uint32_t offt = __shfl(off, tl & 63);
                const uint32_t rel = o - (P & 0x1FFF);
                const uint32_t Lt = (P >> 13) & 63, hdr = 1 + ((P >> 19) & 1);
                const bool is_lit = rel < Lt;
                uint32_t v = __shfl(b, (tl + int(hdr) + int(rel)) & 63);    // literal byte from the window
                const int sp = op + int(o) - int(offt);                     // absolute source of a match byte
                const int cs = op + int(c0);
                const int pn = decltype(first)::value ? 0 : 64;             // bytes pending (the previous step was a full one)
                const bool is_match = live && !is_lit;
                const bool from_mem = is_match && sp < cs - pn;
                const bool in_pend = is_match && sp >= cs - pn && sp < cs;
                const uint32_t ld = dst[from_mem ? sp : 0];                 // issue this step's loads (branch-free) ...
                if constexpr (!decltype(first)::value) {
                    dst[cs - 64 + lane] = uint8_t(pv);                      // ... then store the previous step
                    const uint32_t fw = __shfl(pv, (sp - (cs - 64)) & 63);
                    if (in_pend) v = fw;
                }
                if (from_mem) v = ld;
                bool done = !is_match || from_mem || in_pend;
                int dep = sp - cs;                                          // lane that produces my byte
                while (__ballot(!done)) {                                   // pointer jumping, <= 6 rounds
                    const int d = dep & 63;
                    const uint32_t v2 = __shfl(v, d);
                    const int dn = __shfl(int(done), d);
                    const int dd = __shfl(dep, d);
                    if (!done) { if (dn) { v = v2; done = true; } else dep = dd; }
                }
                pv = v;
            };
            step(0u, std::true_type{});
            uint32_t c0 = 64;
            for (; c0 < T; c0 += 64) step(c0, std::false_type{});
            if (uint32_t(lane) < T - (c0 - 64)) dst[op + int(c0 - 64) + lane] = uint8_t(pv);
        } else {
            wave_copy(dst + op, src + lit_ip, int(lit), lane);
            if (T) copy_match(dst, op + int(lit), int(goff), int(T), lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) st_rel(&sy->done[w], int(k) + kCopiers);
    }
}

// retry_only = 0: fast path for every block (container rules as in the exact kernel);
__global__ __launch_bounds__(64 * (kCopiers + 1))
void lz4_decode_fast_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                            fourmc_block* blocks, uint32_t nblocks, int container_mode, const uint32_t* pick, uint32_t want)
{
    if (pick && *pick != want) return;                                  // (see lz4_pick_kernel)
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    __shared__ __attribute__((aligned(16))) uint8_t own[kCopiers][kOwnBytes];
    __shared__ Rec recs[kRec];
    __shared__ DSync sy;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    if (threadIdx.x == 0) {
        sy.produced = 0; sy.total = 0xFFFFFFFFu; sy.failed = 0; sy.end_value = kRetry;
        for (int w = 0; w < kCopiers; w++) { sy.consumed[w] = 0; sy.done[w] = w; }
    }
    __syncthreads();
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const bool stored = container_mode && blk.src_len == blk.dst_cap;
    if (stored) {
        if (wave == 0) { wave_copy(dst, src, int(blk.src_len), lane); if (lane == 0) blocks[b].result = int(blk.src_len); }
        return;
    }
    if (wave == kCopiers) lz4_fast_parse(src, int(blk.src_len), int(blk.dst_cap), ring, recs, &sy, lane);
    else lz4_fast_copy(src, dst, recs, &sy, own[wave], wave, lane);
    __syncthreads();
    if (threadIdx.x == 0) blocks[b].result = sy.failed ? kRetry : sy.end_value;
}

// second pass: blocks the fast path handed back (result == kRetry) are decoded by the exact walker
__global__ __launch_bounds__(64)
void lz4_decode_retry_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                             fourmc_block* blocks, uint32_t nblocks, int container_mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (blk.result != kRetry) return;
    int r = lz4_decode_block(src_base + blk.src_off, int(blk.src_len), dst_base + blk.dst_off, int(blk.dst_cap),
                             ring, threadIdx.x);
    if (container_mode && r < 0) r = FOURMC_BLK_CORRUPT;
    if (threadIdx.x == 0) blocks[b].result = r;
}

// the same for the segment-parallel path (lz4_seg.hip): blocks it handed back entirely (kRetry), and blocks it executed up to the
// token where the reference's end-of-block rules begin (kResume: token and output position in the block's workspace slot)
__global__ __launch_bounds__(64)
void lz4_decode_resume_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                              fourmc_block* blocks, uint32_t nblocks, int container_mode, const uint32_t* ws, int redo,
                              uint32_t ws_stride, uint32_t res_at)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    int ip0 = 0, op0 = 0;
    if (blk.result == lz4seg::kResumeCode) {
        const uint32_t* meta = ws + size_t(b) * ws_stride;            // the block's slot: {token position, output position} at res_at
        ip0 = __builtin_amdgcn_readfirstlane(int(meta[res_at]));
        op0 = __builtin_amdgcn_readfirstlane(int(meta[res_at + 1]));
    } else if (blk.result != kRetry || !redo) return;
    int r = lz4_decode_block(src_base + blk.src_off, int(blk.src_len), dst_base + blk.dst_off, int(blk.dst_cap),
                             ring, threadIdx.x, ip0, op0);
    if (container_mode && r < 0) r = FOURMC_BLK_CORRUPT;
    if (threadIdx.x == 0) blocks[b].result = r;
}

} // namespace

// Block-parallel path (lz4_parse.hip + lz4_exec.hip) for every block, then the exact walker for whatever they handed back
// (rule violations, blocks beyond the parallel path's size limits), so results and error codes stay the reference's.
extern "C" size_t fourmc_lz4_decode_tok_offset(void) { return lz4par::kTokOff; }
// Device workspace of the block-parallel pair (parse records): 10.8 MB per block, 21.7 GiB for a full batch.  Only launches
// that run that pair need it; every other decode path leases nothing (tens of GiB of dead HBM per stream).
extern "C" size_t fourmc_lz4_parse_work_bytes(uint32_t n)
{
    const uint32_t m = n < lz4par::kMaxBatch ? n : lz4par::kMaxBatch;
    return size_t(m ? m : 1) * lz4par::kSlotBytes;
}
extern "C" int fourmc_gpu_get_lz4_decode_path(void);
// Which kernels a launch of n blocks runs, in pieces of how many blocks, with how much workspace: resolved ONCE per call (the
// lease and the launch see the same answer even if another thread changes the selection in between).  shrink: how often the
// workspace could not be had - the pieces halve; below 64 blocks an automatic choice falls back to the walk + window copier,
// which needs no workspace.
// "auto" by launch size (measured on the container corpus, tools/decode_sizes.sh: 128 .. 512 blocks 13.5 - 15 ms through the tile
// path against 30 ms through either other path, 1024: 24 / 31 ms, 1536: 34 / 34 ms, 2048: 43 / 35 ms, 16 384: 299 / 172 ms): the tile
// path - one workgroup per block, two per CU, the shortest chain per block - up to kAutoTileMax blocks, the segment-parallel path
// - one wave per block, 24 per CU - for launches that fill the chip several times over.  FOURMC_TILE_MAX overrides.
static uint32_t auto_tile_max()
{
    static const uint32_t v = [] { const char* e = getenv("FOURMC_TILE_MAX"); const long x = e ? atol(e) : -1; return x >= 0 ? uint32_t(x) : 1536u; }();
    return v;
}
extern "C" fourmc_lz4_plan fourmc_lz4_decode_plan(uint32_t n, uint32_t shrink)
{
    fourmc_lz4_plan pl; pl.path = fourmc_gpu_get_lz4_decode_path(); pl.batch = n ? n : 1; pl.work_bytes = 0; pl.ok = 1;
    const bool automatic = pl.path == 6;
    if (pl.path == 6) pl.path = n <= auto_tile_max() ? 13 : 11;
    if (pl.path >= 11 && pl.path <= 16) {
        const bool tile = pl.path == 13 || pl.path == 14;
        uint32_t b = tile ? fourmc_lz4_tile_batch() : fourmc_lz4_seg_batch();
        if (b < 64) b = 64;                                       // (FOURMC_TILE_BATCH / FOURMC_SEG_BATCH below 64: an explicit choice would fail before any allocation)
        for (uint32_t k = 0; k < shrink && b >= 64; k++) b /= 2;
        if (b < 64) {
            if (automatic) { pl.path = 9; return pl; }
            pl.ok = 0; return pl;
        }
        pl.batch = n < b ? (n ? n : 1) : b;
        pl.work_bytes = tile ? fourmc_lz4_tile_work_bytes(pl.batch) : fourmc_lz4_seg_work_bytes(pl.batch);
        return pl;
    }
#ifdef FOURMC_RESEARCH
    if (pl.path == 1 || pl.path == 3) pl.work_bytes = fourmc_lz4_parse_work_bytes(n);
#endif
    return pl;
}

// Which fast path serves LZ4 decode launches.  Both produce identical results (anything irregular goes to the exact
// walker either way); they differ in how a block is parallelised:
//   6  "auto"            9: the default since the end of round 3
//   9  "wx"              walk wave + sequence / literal wave + window copier (plan and execute waves), lz4_rows.hip (K1wx)
//   4  "rows"            row-parallel pipeline of four waves per block (lz4_rows.hip)
//   7  "lanes"           one lane per sequence, wide pieces (lz4_rows.hip, K1w): 84 ms on the S-mix, opt-in
//   0  "wave trio"       one parser wave walks the token chain, two copier waves execute (lz4_decode_fast_kernel)
//   1  "block parallel"  parse kernel (token chain found by the whole workgroup, records in HBM) + executor kernel
//                        (16 KiB LDS ring, literal / chain / flush waves)            lz4_parse.hip, lz4_exec.hip
// Measured on 2048 x 4 MiB of S-mix (profiles/r02_*): 0 = 57 ms, 1 = 38 + 52 ms, so 0 stays the default; the
// environment variable FOURMC_DECODE (auto | wx | rows | lanes | exact | trio | par | paronly | rowsonly | lanesonly | wxonly) or fourmc_gpu_set_lz4_decode_path() select.
static int g_decode_path = -1;
// The product library carries three decoders: the exact walker (2), the walk + window copier (9, 10) and the segment-parallel path
// (11, 12); 6 = auto.  The wave trio, the row pipeline, the lane-per-sequence path and the block-parallel pair are measured
// alternatives kept in the research side build (make research: libhadoop-4mc-research.so, -DFOURMC_RESEARCH).
static bool path_known(int path)
{
#ifdef FOURMC_RESEARCH
    return path >= 0 && path <= 16;
#else
    return path == 2 || path == 6 || (path >= 9 && path <= 16);
#endif
}
extern "C" void fourmc_gpu_set_lz4_decode_path(int path) { g_decode_path = path_known(path) ? path : 6; }
extern "C" int fourmc_gpu_get_lz4_decode_path(void)
{
    if (g_decode_path < 0) {
        const char* mode = getenv("FOURMC_DECODE");
        int p = 6;
        if (mode && !strcmp(mode, "rows")) p = 4;
        if (mode && !strcmp(mode, "trio")) p = 0;
        if (mode && !strcmp(mode, "rowsonly")) p = 5;
        if (mode && !strcmp(mode, "lanes")) p = 7;
        if (mode && !strcmp(mode, "lanesonly")) p = 8;
        if (mode && !strcmp(mode, "wx")) p = 9;
        if (mode && !strcmp(mode, "wxonly")) p = 10;
        if (mode && !strcmp(mode, "exact")) p = 2;
        if (mode && !strcmp(mode, "seg")) p = 11;
        if (mode && !strcmp(mode, "segonly")) p = 12;
        if (mode && !strcmp(mode, "tile")) p = 13;
        if (mode && !strcmp(mode, "tileonly")) p = 14;
        if (mode && !strcmp(mode, "ring")) p = 15;
        if (mode && !strcmp(mode, "ringonly")) p = 16;
        if (mode && !strcmp(mode, "par")) p = 1;
        if (mode && !strcmp(mode, "paronly")) p = 3;
        g_decode_path = path_known(p) ? p : 6;
    }
    return g_decode_path;
}

extern "C" hipError_t fourmc_launch_lz4_decode(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                               uint32_t n, int container_mode, const fourmc_lz4_plan* plan, void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint8_t* s8 = static_cast<const uint8_t*>(d_src);
    uint8_t* d8 = static_cast<uint8_t*>(d_dst);
    const int path = plan->path;
    // (6 "auto" was resolved by fourmc_lz4_decode_plan: the tile path - lz4_tile.hip, one workgroup per block with the LZ4 window in
    // LDS - up to 1536 blocks, the segment-parallel path above; the walk + window copier, K1wx, when neither can have its workspace)
    if (path >= 11 && path <= 16) {
        // walk + executor (tile: lz4_tile.hip, segment-parallel: lz4_seg.hip), then the exact walker for the last bytes of every
        // block and for whatever was handed back; 12 / 14: test aid, blocks handed back stay kRetry
        const bool tile = path == 13 || path == 14, ring = path >= 15;
        const uint32_t step = plan->batch ? plan->batch : n;
        if (plan->work_bytes == 0 || d_work == nullptr) return hipErrorInvalidValue;
        for (uint32_t b0 = 0; b0 < n; b0 += step) {
            const uint32_t m = n - b0 < step ? n - b0 : step;
            hipError_t e = tile ? fourmc_launch_lz4_tile(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream)
                         : ring ? fourmc_launch_lz4_ring(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream)
                                : fourmc_launch_lz4_seg(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(lz4_decode_resume_kernel, dim3(m), dim3(64), 0, stream, s8, d8, d_blocks + b0, m, container_mode,
                               static_cast<const uint32_t*>(d_work), (path == 11 || path == 13 || path == 15) ? 1 : 0,
                               tile ? uint32_t(lz4tile::kWsWords) : uint32_t(lz4seg::kWsWords), tile ? lz4tile::kMetaResIp : lz4seg::kMetaResIp);
        }
        return hipGetLastError();
    }
    if (path == 2) {
        hipLaunchKernelGGL(lz4_decode_exact_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    if (path == 9 || path == 10) {
        hipError_t e = fourmc_launch_lz4_wx(d_src, d_dst, d_blocks, n, container_mode, stream, nullptr, 0);
        if (e != hipSuccess || path == 10) return e;      // 10: test aid, shows what the path alone did
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
#ifdef FOURMC_RESEARCH
    if (path == 4 || path == 5) {
        hipError_t e = fourmc_launch_lz4_rows(d_src, d_dst, d_blocks, n, container_mode, stream);
        if (e != hipSuccess || path == 5) return e;       // 5: test aid, shows what the row pipeline alone did
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    if (path == 7 || path == 8) {
        hipError_t e = fourmc_launch_lz4_lanes(d_src, d_dst, d_blocks, n, container_mode, stream);
        if (e != hipSuccess || path == 8) return e;       // 8: test aid, shows what the lane-per-sequence path alone did
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    if (path == 0) {
        hipLaunchKernelGGL(lz4_decode_fast_kernel, dim3(n), dim3(64 * (kCopiers + 1)), 0, stream, s8, d8, d_blocks, n, container_mode, (const uint32_t*)nullptr, 0u);
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    for (uint32_t b0 = 0; b0 < n; b0 += lz4par::kMaxBatch) {
        const uint32_t m = n - b0 < lz4par::kMaxBatch ? n - b0 : lz4par::kMaxBatch;
        hipError_t e = fourmc_launch_lz4_parse(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream);
        if (e != hipSuccess) return e;
        e = fourmc_launch_lz4_exec(d_src, d_dst, d_blocks + b0, m, d_work, stream);
        if (e != hipSuccess) return e;
    }
    if (path == 3) return hipGetLastError();              // test aid: show what the parallel path alone did
    hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;                          // (path_known() keeps every other value out)
#endif
}
