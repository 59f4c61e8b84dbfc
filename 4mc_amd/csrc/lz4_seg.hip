// 4mc_amd/csrc/lz4_seg.hip - K1s: segment-parallel LZ4 block decode on gfx950 (wave64).
//
// Replaces LZ4_decompress_safe(in, out, csize, usize) per block (native/4mc.c:661, native/jniDecompressor.c:88 ->
// native/lz4/lz4.c:2345-2350 -> :1936-2339) for every block the exact walker (lz4_decode.hip) does not have to see.
//
// What is serial in an LZ4 block is only WHERE THE TOKENS ARE: token k+1 starts where token k's bytes end.  But the map
// "position -> next position" is a function of the stream alone, and a chain started at an arbitrary byte falls onto the true chain
// after a few hops (it only has to land on one true token start; measured on the S-mix: 5-40 true hops, tools/model/seg_model.c).  So:
//
//   WALK kernel, one wave per block, one LANE per stream segment (64 segments of csize/64 bytes): every lane walks the chain that
//     starts at its segment's first byte - a per-lane scalar loop, 64 of them in lockstep, two dependent loads per hop - and records
//     {token position, literal length} of what it meets.  Then the true chain is threaded through the segments: lane j takes the
//     position where the chain of the segment before it left (speculatively: where the SPECULATIVE chain left), walks from there until
//     it meets its own recorded chain (a two-pointer merge against its list; the hops in between go to a short fix list), and a serial
//     pass over the 64 segments checks every hand-over and redoes what was assumed wrong.  The result is exact: a list of live
//     segments, each {fix records, where its recorded list becomes true, how many records}.
//   EXEC kernel, one wave per block: 64 CONSECUTIVE sequences per step, one per lane.  Token and offset from the stream, a prefix sum
//     places every sequence; the batch's output is assembled in an LDS staging buffer that was zeroed before - literals and matches
//     whose source lies well before the batch are loaded 32 bytes per lane (unaligned loads, phase-aligned to the destination) and
//     OR-ed in as dwords (ds_or_b32: no read-modify-write hazard between lanes that share a dword); matches whose source lies in the
//     batch itself, overlapping and long matches run one sequence at a time, a byte per lane, inside the buffer (an LDS round trip per
//     sequence instead of a trip to memory per dependency level); the buffer leaves for memory in aligned 16-byte stores.
//   The last 64 stream bytes / 128 output bytes of a block - where the reference's end-of-block rules apply (lz4.c:2120-2330) - and
//     anything irregular go to the exact walker: it RESUMES at the token the fast path stopped at (kResume) or redoes the block
//     (kRetry), so accept / reject set and return codes stay the reference's.
//
// All byte work; no MFMA.  tools/model/seg_decode_model.c is the executable model these kernels were written from.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"
#include "lz4seg.h"
#ifndef FOURMC_SEG_RING
#define FOURMC_SEG_RING 64
#endif

namespace {

using namespace lz4seg;

typedef __attribute__((address_space(1))) uint8_t gbyte;
typedef __attribute__((address_space(1))) const uint8_t cgbyte;
typedef __attribute__((address_space(1))) uint32_t gword;
typedef __attribute__((address_space(1))) const uint32_t cgword;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld4u(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32_u*>(p); }
__device__ __forceinline__ u32x2 ld8u(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32x2_u*>(p); }
__device__ __forceinline__ u32x4 ld16u_g(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32x4_u*>(p); }
__device__ __forceinline__ void st16g(gbyte* p, u32x4 v) { *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(p) = v; }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t scan_add(uint32_t v)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(l))); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
// LDS operations of one wave execute in order; what has to be stopped is the COMPILER moving one lane's accesses across another
// lane's (it reasons per thread).  A full wait is cheap here (the queue is a few operations deep) and is the barrier it respects.
#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define VM_DRAIN()  asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// profiling build (make sprof, -DK1S_PROF): cycle counters per phase, left in the spare words of the block's meta area
#ifdef K1S_PROF
struct Prof {
    unsigned long long t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void add(int i, unsigned long long& since) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t[i] += n - since; since = n; }
    __device__ __forceinline__ void count(int i, unsigned long long n = 1) { t[i] += n; }
    __device__ __forceinline__ void dump(gword* meta, uint32_t at, int lane) const { if (lane == 0) for (int i = 0; i < 12; i++) { meta[at + 2 * i] = uint32_t(t[i]); meta[at + 2 * i + 1] = uint32_t(t[i] >> 32); } }
};
#else
struct Prof {
    __device__ __forceinline__ unsigned long long now() const { return 0; }
    __device__ __forceinline__ void add(int, unsigned long long&) {}
    __device__ __forceinline__ void count(int, unsigned long long = 1) {}
    __device__ __forceinline__ void dump(gword*, uint32_t, int) const {}
};
#endif

// ================================================================================================ token at a position
// Length extension bytes from q on - 255, 255, ..., b (b < 255) - eight per load: adds them to `len`, leaves q behind the last.
// false: they reach `limit`, or len passes cap on a 255 (an incompressible block is one long run per token: a byte per trip to
// memory made its walk 2 x slower than a text block's).
__device__ __forceinline__ bool ext_run(cgbyte* s, uint32_t limit, uint32_t& q, uint32_t& len, uint32_t cap)
{
    for (;;) {
        if (q + 8u <= limit) {
            const u32x2 v = ld8u(s + q);
            const uint32_t nlo = ~v.x, nhi = ~v.y;
            if ((nlo | nhi) == 0u) { len += 8u * 255u; q += 8u; if (len > cap) return false; continue; }
            const uint32_t k = nlo ? uint32_t(__builtin_ctz(nlo)) >> 3 : 4u + (uint32_t(__builtin_ctz(nhi)) >> 3);
            const uint32_t b = k < 4u ? (v.x >> (8u * k)) & 255u : (v.y >> (8u * (k - 4u))) & 255u;
            len += 255u * k;
            if (k && len > cap) return false;
            len += b; q += k + 1u;
            return true;
        }
        if (q >= limit) return false;
        const uint32_t b = s[q++]; len += b;
        if (b != 255u) return true;
        if (len > cap) return false;
    }
}

struct Hop { uint32_t ll, next, offml; bool stop, esc; };      // offml: match offset | match length << 16 (exact unless esc)
// One token at p (per lane).  stop: the token or its bytes reach beyond limit = csize - kMargin; the chain halts AT p and the exact
// walker takes over there.  Every byte read lies below csize.
__device__ __forceinline__ Hop decode_tok(cgbyte* s, uint32_t limit, uint32_t p)
{
    Hop h; h.ll = 0; h.next = p; h.offml = 0; h.stop = true; h.esc = false;
    if (p >= limit) return h;
    const u32x2 L0 = ld8u(s + p);                               // p + 8 <= csize - 56
    const uint32_t tok = L0.x & 255u, mn = tok & 15u;
    uint32_t ll = tok >> 4, q = p + 1;
    if (ll == 15) {
        uint32_t b = (L0.x >> 8) & 255u; ll += b; q++;
        if (b == 255u) {
            if (ll > (1u << 23) || !ext_run(s, limit, q, ll, 1u << 23)) return h;
        }
    }
    const uint32_t mo = q + ll;
    if (mo + 2 > limit) return h;
    const uint32_t L1 = ld4u(s + mo);                           // mo + 4 <= limit + 2
    uint32_t q2 = mo + 2, ml = mn + 4u; bool esc = ll >= kEscLL;
    if (mn == 15) {
        const uint32_t e0 = (L1 >> 16) & 255u; q2++; ml += e0;
        if (e0 == 255u) {
            const uint32_t e1 = L1 >> 24; q2++; ml += e1;
            if (e1 == 255u) { esc = true; uint32_t mx = 0; if (!ext_run(s, limit, q2, mx, 0xFFFFFFFFu)) return h; }
        }
    }
    if (q2 > limit) return h;
    h.ll = ll; h.next = q2; h.offml = (L1 & 0xFFFFu) | (ml << 16); h.stop = false; h.esc = esc;
    return h;
}
__device__ __forceinline__ uint32_t pack_rec(uint32_t p, const Hop& h) { return p | ((h.esc ? kEscLL : h.ll) << kPosBits); }     // a record's first word; the second is h.offml

// ================================================================================================ WALK kernel
struct LaneSeg {            // one lane's segment
    gword* F; gword* L;     // fix list (kFixCap), recorded list
    uint32_t seg_end;
    uint32_t f, k, n, exitp, entry; bool tail, pure;
};
// (re)walk the segment from `start`, recording from index 0.
// The lane reads its stretch of the stream through a window of its own in LDS - a ring of 64 dwords (256 stream bytes), dword k of
// lane l at word k * 64 + l, so that 64 lanes reading "their" dword never meet in a bank.  The windows of ALL lanes still walking are
// topped up together whenever one of them has less than 64 bytes ahead (up to 15 loads of 16 bytes per lane, issued back to back: one
// trip to memory per ~25 hops instead of two per hop; 64 lanes x 8 waves per CU walking lines of their own overflow the L1: 2.1 us per
// hop without the window), and four records leave in one 16-byte store.
constexpr uint32_t kRingDw = FOURMC_SEG_RING;   // dwords of stream window per lane (LDS: 256 bytes x kRingDw per wave)
__device__ __forceinline__ void walk_from(LaneSeg& g, cgbyte* s, uint32_t csize, uint32_t limit, uint32_t start, uint32_t* ring)
{
    constexpr uint32_t M = kRingDw - 1u, W = 4u * kRingDw, LOW = W / 4u;       // refill when less than a quarter of the window is ahead
    uint32_t p = start, n = 0; bool tail;
    uint32_t wlo = start & ~15u, whi = wlo;                            // the ring holds stream bytes [wlo, whi), both multiples of 16
    const uint32_t fill_end = csize & ~15u;                            // whole 16-byte pieces only
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    // 4 stream bytes at a: from the lane's window; from memory for the lanes whose window does not hold them (a literal run longer than
    // the window, the last bytes of the stream) - on a path of its own, taken when ANY lane needs it: a load under a per-lane condition
    // would put a wait for all memory operations, the record stores' acknowledgements included, behind every token of every lane
    auto in_win = [&](uint32_t a) -> bool { return a >= wlo && a + 4u <= whi; };
    auto win4 = [&](uint32_t a) -> uint32_t {
        const uint32_t d = a >> 2, lo = ring[(d & M) * 64u], hi = ring[((d + 1u) & M) * 64u];
        return __builtin_amdgcn_alignbyte(hi, lo, a & 3u);
    };
    auto get4 = [&](uint32_t a) -> uint32_t {
        if (__builtin_expect(__ballot(!in_win(a)) != 0, 0)) {
            uint32_t v = 0;
            if (!in_win(a)) v = ld4u(s + a);
            asm volatile("" : "+v"(v));                    // the load is waited for HERE, on this path (the compiler's own wait)
            return in_win(a) ? win4(a) : v;
        }
        return win4(a);
    };
    for (;;) {
        if (p >= g.seg_end) { tail = false; break; }
        if (p >= limit) { tail = true; break; }
        if (__ballot(p + LOW > whi && whi < fill_end)) {
            if (p >= whi || p < wlo) { wlo = p & ~15u; whi = wlo; }
            uint32_t target = (p & ~15u) + (W - 16u); target = target < fill_end ? target : fill_end;
            for (int round = 0; round < (W > 128u ? 2 : 1); round++) {
                if (!__ballot(whi < target)) break;
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) if (whi + 16u * j < target) v[j] = ld16u_g(s + whi + 16u * j);
#pragma unroll
                for (int j = 0; j < 8; j++) if (whi + 16u * j < target) {
                    const uint32_t d = (whi >> 2) + 4u * j;
                    ring[(d & M) * 64u] = v[j].x; ring[((d + 1u) & M) * 64u] = v[j].y; ring[((d + 2u) & M) * 64u] = v[j].z; ring[((d + 3u) & M) * 64u] = v[j].w;
                }
                const uint32_t got = target - whi; whi += got < 128u ? got : 128u;
            }
            if (whi - wlo > W) wlo = whi - W;
        }
        // the token (as decode_tok)
        const uint32_t L0 = get4(p);
        const uint32_t tok = L0 & 255u, mn = tok & 15u;
        uint32_t ll = tok >> 4, q = p + 1; bool stop = false;
        if (ll == 15) {
            uint32_t bq = (L0 >> 8) & 255u; ll += bq; q++;
            if (bq == 255u) stop = ll > (1u << 23) || !ext_run(s, limit, q, ll, 1u << 23);
        }
        const uint32_t mo = q + ll;
        if (stop || mo + 2 > limit) { tail = true; break; }
        const uint32_t L1 = get4(mo);
        uint32_t q2 = mo + 2, ml = mn + 4u; bool esc = ll >= kEscLL;
        if (mn == 15) {
            const uint32_t e0 = (L1 >> 16) & 255u; q2++; ml += e0;
            if (e0 == 255u) {
                const uint32_t e1 = L1 >> 24; q2++; ml += e1;
                if (e1 == 255u) { esc = true; uint32_t mx = 0; stop = !ext_run(s, limit, q2, mx, 0xFFFFFFFFu); }
            }
        }
        if (stop || q2 > limit) { tail = true; break; }
        r0 = r2; r1 = r3; r2 = p | ((esc ? kEscLL : ll) << kPosBits); r3 = (L1 & 0xFFFFu) | (ml << 16);      // two records (of two words) per 16-byte store
        n++;
        if ((n & 1u) == 0) *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(g.L + 2 * (n - 2)) = u32x4{r0, r1, r2, r3};
        p = q2;
    }
    if (n & 1u) { g.L[2 * (n - 1)] = r2; g.L[2 * (n - 1) + 1] = r3; }       // the record still in the registers
    g.exitp = p; g.n = n; g.f = 0; g.k = 0; g.entry = start; g.tail = tail; g.pure = true;
}
// the true chain enters the segment at e: walk it until it falls onto the recorded chain (which must be a pure one)
__device__ __forceinline__ void fix_from(LaneSeg& g, cgbyte* s, uint32_t csize, uint32_t limit, uint32_t e, uint32_t* ring)
{
    uint32_t q = e, idx = 0, f = 0; const uint32_t n = g.n;
    uint32_t cur = n ? (g.L[0] & kPosMask) : 0xFFFFFFFFu;        // position of record idx
    for (;;) {
        while (idx < n && cur < q) { idx++; cur = idx < n ? (g.L[2 * idx] & kPosMask) : 0xFFFFFFFFu; }
        if (idx < n && cur == q) { g.k = idx; break; }
        if (idx == n && q == g.exitp) { g.k = n; break; }
        if (q >= g.seg_end) { g.k = n; g.exitp = q; g.tail = false; break; }
        const Hop h = decode_tok(s, limit, q);
        if (h.stop) { g.k = n; g.exitp = q; g.tail = true; break; }
        if (f == kFixCap) { walk_from(g, s, csize, limit, e, ring); return; }
        g.F[2 * f] = pack_rec(q, h); g.F[2 * f + 1] = h.offml; f++; q = h.next;
    }
    g.f = f; g.entry = e; g.pure = (f == 0 && g.k == 0);
}

__device__ __forceinline__ bool eligible(const fourmc_block& blk)
{ return blk.src_len >= kMinSrc && blk.src_len <= kMaxSrc && blk.dst_cap >= kMinCap && blk.dst_cap <= lz4par::kDstMax; }

__global__ __launch_bounds__(64)
void lz4_seg_walk_kernel(const uint8_t* __restrict__ src_base, const fourmc_block* blocks, uint32_t nblocks,
                         int container_mode, uint32_t* ws)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    gword* meta = (gword*)(ws + size_t(b) * kWsWords);
    const uint32_t lane = threadIdx.x;
    const bool skip = (container_mode && (blk.result == FOURMC_BLK_BADSUM || blk.src_len == blk.dst_cap)) || !eligible(blk);
    if (skip) { if (lane == 0) meta[kMetaStatus] = 0; return; }
    cgbyte* s = (cgbyte*)(src_base + blk.src_off);
    const uint32_t csize = blk.src_len, limit = csize - kMargin;
    uint32_t nseg = limit / kMinSeg; nseg = nseg < 1 ? 1 : (nseg > uint32_t(kSegs) ? uint32_t(kSegs) : nseg);
    const uint32_t seglen = ((limit + nseg - 1) / nseg + 3) & ~3u;
    const uint32_t stride = kRecWords * ((kFixCap + seglen / 3 + 7) & ~3u);          // lists start on 16-byte boundaries
    const uint32_t area = kMetaWords + lane * stride;
    __shared__ uint32_t win[kRingDw * 64];
    uint32_t* ring = win + lane;

    LaneSeg g;
    g.F = meta + area; g.L = g.F + kRecWords * kFixCap;
    g.seg_end = (lane + 1 == nseg) ? 0xFFFFFFFFu : (lane + 1) * seglen;
    g.f = g.k = g.n = 0; g.exitp = 0; g.entry = 0xFFFFFFFFu; g.tail = true; g.pure = true;
    const bool mine = lane < nseg;
    Prof pf; unsigned long long tp = pf.now();
    // phase 1: every lane its own chain
    if (mine) walk_from(g, s, csize, limit, lane * seglen, ring);
    pf.add(0, tp); pf.count(3, rdl(g.n, 0));
    // phase 2: the chain of the segment in front left at `pe`: assume it is the true one, thread it into this segment
    {
        const int from = int(lane ? lane - 1 : 0) * 4;           // lane j reads lane j-1
        const uint32_t pe = uint32_t(__builtin_amdgcn_ds_bpermute(from, int(g.exitp)));
        const uint32_t pt = uint32_t(__builtin_amdgcn_ds_bpermute(from, int(g.tail ? 1u : 0u)));
        uint32_t sj = pe / seglen; sj = sj > nseg - 1 ? nseg - 1 : sj;
        if (mine && lane >= 1 && !pt && sj == lane && pe != lane * seglen) fix_from(g, s, csize, limit, pe, ring);
    }
    pf.add(1, tp);
    // phase 3: follow the true chain through the segments; redo what was assumed wrong (one lane at a time: rare)
    uint32_t cur = 0, nlive = 0, tail_ip = 0;
    for (;;) {
        if (lane == cur) {
            gword* e = meta + kMetaLive + 4 * nlive;
            e[0] = area; e[1] = g.f; e[2] = g.k; e[3] = g.f + g.n - g.k;
        }
        nlive++;
        const uint32_t ex = rdl(g.exitp, cur);
        if (rdl(g.tail ? 1u : 0u, cur)) { tail_ip = ex; break; }
        uint32_t j = ex / seglen; j = j > nseg - 1 ? nseg - 1 : j;
        if (rdl(g.entry, j) != ex) {
            if (lane == j) { if (g.pure) fix_from(g, s, csize, limit, ex, ring); else walk_from(g, s, csize, limit, ex, ring); }
            pf.count(4);
        }
        cur = j;
    }
    pf.add(2, tp); pf.dump(meta, kMetaProf, lane);
    if (lane == 0) { meta[kMetaNLive] = nlive; meta[kMetaTailIp] = tail_ip; meta[kMetaStatus] = 1; }
}

// ================================================================================================ EXEC kernel
// OR `len` (1..32) string bytes into the staging buffer at byte address pd.  R[0..8] holds the string phase-aligned to the
// destination: string byte b sits at byte (da + b) of R, da = pd & 3; bytes of R outside the string are arbitrary.
__device__ __forceinline__ void or_store(uint8_t* st, uint32_t pd, const uint32_t (&R)[9], uint32_t len, bool on)
{
    // A lane is switched off for the dwords it does not reach (measured: letting it OR a zero instead - no exec-mask round trip -
    // made the kernel 1.5 x SLOWER: an LDS atomic costs per lane and per lane that meets another on the same dword); the dwords
    // beyond the fifth are skipped as a whole when no lane reaches them (strings of up to 17 bytes: most batches)
    const uint32_t da = pd & 3u, end = da + len, last = (end + 3u) / 4u - 1u, tb = end & 3u;
    const uint32_t hmask = 0xFFFFFFFFu << (8u * da), tmask = tb ? ((1u << (8u * tb)) - 1u) : 0xFFFFFFFFu;
    uint32_t* w = reinterpret_cast<uint32_t*>(st + (pd & ~3u));
#pragma unroll
    for (uint32_t j = 0; j < 5; j++) {
        uint32_t v = R[j];
        if (j == 0) v &= hmask;
        v = j == last ? (v & tmask) : v;
        if (on && j <= last) __hip_atomic_fetch_or(w + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    if (__ballot(on && last >= 5u)) {
#pragma unroll
        for (uint32_t j = 5; j < 9; j++) {
            uint32_t v = R[j];
            v = j == last ? (v & tmask) : v;
            if (on && j <= last) __hip_atomic_fetch_or(w + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}
// nine dwords from memory so that the byte at `base + a` lands on byte `da` of R[0]
__device__ __forceinline__ void load_phase(cgbyte* base, uint32_t a, uint32_t da, uint32_t (&R)[9], bool on, uint32_t len)
{
    // (the second 16 bytes and the last dword are loaded only when some lane's string reaches them: a load of 64 lanes at 64
    // places costs the L1 64 look-ups whatever it returns)
    const bool far1 = __ballot(on && da + len > 16u) != 0, far2 = __ballot(on && da + len > 32u) != 0;
#pragma unroll
    for (int j = 4; j < 9; j++) R[j] = 0;
    if (on) {
        if (a >= da) {
            cgbyte* p = base + (a - da);
            const u32x4 v0 = ld16u_g(p);
            R[0] = v0.x; R[1] = v0.y; R[2] = v0.z; R[3] = v0.w;
            if (far1) { const u32x4 v1 = ld16u_g(p + 16); R[4] = v1.x; R[5] = v1.y; R[6] = v1.z; R[7] = v1.w; }
            if (far2) R[8] = ld4u(p + 32);
        } else {
            // the first bytes of the buffer with a destination phase that would read in front of it (a < da <= 3): the same
            // loads from the buffer's first byte, shifted up by da - a bytes in registers
            uint32_t L[9];
            const u32x4 v0 = ld16u_g(base), v1 = ld16u_g(base + 16); const uint32_t v2 = ld4u(base + 32);
            L[0] = v0.x; L[1] = v0.y; L[2] = v0.z; L[3] = v0.w; L[4] = v1.x; L[5] = v1.y; L[6] = v1.z; L[7] = v1.w; L[8] = v2;
            const uint32_t k = 4u - (da - a);
#pragma unroll
            for (int j = 0; j < 9; j++) R[j] = __builtin_amdgcn_alignbyte(L[j], j ? L[j - 1] : 0u, k);
        }
    }
}

__global__ __launch_bounds__(64)
void lz4_seg_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                         int container_mode, uint32_t* ws)
{
    __shared__ __attribute__((aligned(16))) uint8_t st[kStage];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    const int lane = threadIdx.x;
    if (container_mode && blk.src_len == blk.dst_cap) {                 // stored block (native/4mc.c:635-642)
        wave_copy(dst_base + blk.dst_off, src_base + blk.src_off, int(blk.src_len), lane);
        if (lane == 0) blocks[b].result = int(blk.src_len);
        return;
    }
    gword* meta = (gword*)(ws + size_t(b) * kWsWords);
    if (!eligible(blk) || rfl(meta[kMetaStatus]) != 1u) { if (lane == 0) blocks[b].result = lz4par::kRetryCode; return; }
    cgbyte* s = (cgbyte*)(src_base + blk.src_off);
    gbyte* dst = (gbyte*)(dst_base + blk.dst_off);
    const uint32_t cap = blk.dst_cap, limit = blk.src_len - kMargin, olimit = cap - kOMargin;
    const uint32_t nlive = rfl(meta[kMetaNLive]);
    uint32_t res_ip = rfl(meta[kMetaTailIp]);
    uint32_t opos = 0;
    bool failed = false, cut = false;

    for (uint32_t a = 16u * lane; a < uint32_t(kStage); a += 1024u) *reinterpret_cast<u32x4*>(st + a) = u32x4{0, 0, 0, 0};
    LDS_FENCE();
    Prof pf; unsigned long long tp = pf.now();

    for (uint32_t li = 0; li < nlive && !failed && !cut; li++) {
        cgword* e = (cgword*)(meta + kMetaLive + 4 * li);
        const uint32_t area = rfl(e[0]), f = rfl(e[1]), k = rfl(e[2]), c = rfl(e[3]);
        cgword* FL = (cgword*)(meta + area);
        uint32_t t0 = 0;
        // the records of the NEXT step are loaded as soon as this step knows how many sequences it takes; their wait is placed in
        // front of this step's flush (where nothing else is outstanding), not behind the flush's stores
        u32x2 nrec = u32x2{0, 0}; bool have = false;
        auto rec_at = [&](uint32_t t) -> u32x2 { return *reinterpret_cast<__attribute__((address_space(1))) const u32x2*>(FL + 2 * (t < f ? t : kFixCap + k + (t - f))); };
        while (t0 < c) {
            // ---- records and fields, one sequence per lane
            const uint32_t t = t0 + lane;
            const bool valid = t < c;
            u32x2 rec2 = u32x2{0, 0};
#ifdef K1S_NOPREFETCH
            if (valid) rec2 = rec_at(t);
#else
            // (a step without prefetched records loads them here and waits INSIDE the branch: a register that is either ready or still
            // on its way when the paths meet would be waited for on both)
            if (!have) { nrec = rec_at(t < c ? t : c - 1u); asm volatile("" : "+v"(nrec.x), "+v"(nrec.y)); }
            if (valid) rec2 = nrec;
            have = false;
#endif
            const uint32_t rec = rec2.x, pos = rec & kPosMask, ll = rec >> kPosBits;
            const bool esc = valid && ll == kEscLL;
            uint32_t off = 0, ml = 0, lsrc = 0;
            if (valid && !esc) {
                const uint32_t llx = ll < 15u ? 0u : 1u + (ll >= 270u);
                lsrc = pos + 1u + llx;
                off = rec2.y & 0xFFFFu; ml = rec2.y >> 16;               // (the walk read them: no trip to the stream for the token and the offset)
            }
            const uint32_t sz = ll + ml;                               // 0 for lanes without a sequence (ll = ml = 0) - escapes are cut off below
            const uint32_t incl = scan_add(valid && !esc ? sz : 0u);
            // ---- how many sequences the batch takes: a prefix
            const bool fits = valid && !esc && incl <= uint32_t(kCapB) && opos + incl <= olimit;
            const unsigned long long okm = __ballot(fits);
            const uint32_t cnt = ~okm ? uint32_t(__builtin_ctzll(~okm)) : 64u;
            if (cnt == 0) {
                if (!rdl(esc ? 1u : 0u, 0)) { res_ip = rdl(pos, 0); cut = true; break; }       // the output-side tail starts here
                // ---- one long sequence, wave-wide, straight in memory (lz4.c:2175-2330 without the end-of-block cases)
                const uint32_t p0 = rdl(pos, 0);
                // (every lane loads the same bytes: say so, or the compiler takes the lengths - and with them the output position
                // and every loop around this - for per-lane values)
                uint32_t tok = rfl(s[p0]), q = p0 + 1, L = tok >> 4, M = (tok & 15u) + 4u;
                if (L == 15u) for (;;) { if (q >= limit) { failed = true; break; } const uint32_t bb = rfl(s[q++]); L += bb; if (bb != 255u) break; }
                if (failed || L > (1u << 23) || q + L + 2 > limit) { failed = true; break; }
                const uint32_t mo = q + L, o16 = rfl(uint32_t(s[mo]) | (uint32_t(s[mo + 1]) << 8));
                uint32_t q2 = mo + 2;
                if ((tok & 15u) == 15u) for (;;) { if (q2 >= limit) { failed = true; break; } const uint32_t bb = rfl(s[q2++]); M += bb; if (bb != 255u) break; if (M > (1u << 23)) { failed = true; break; } }
                if (failed) break;
                if (uint64_t(opos) + L + M > uint64_t(olimit)) { res_ip = p0; cut = true; break; }
                const uint32_t m = opos + L;
                if (o16 == 0 || o16 > m) { failed = true; break; }
                VM_DRAIN();                                              // the flushes before this sequence are its sources
                wave_copy((uint8_t*)dst + opos, (const uint8_t*)s + q, int(L), lane);
                VM_DRAIN();
                copy_match((uint8_t*)dst, int(m), int(o16), int(M), lane);
                VM_DRAIN();
                opos = m + M; t0 += 1; pf.count(9);
                {   // the prologue of the next batch comes from memory (the one in the buffer is the previous batch's: out)
                    if (lane < 4) *reinterpret_cast<u32x4*>(st + 32 + 16 * lane) = u32x4{0, 0, 0, 0};
                    LDS_FENCE();
                    const uint32_t P1 = 64u + uint32_t(uintptr_t(dst + opos) & 15u);
                    if (lane < kPro && opos + lane >= uint32_t(kPro)) st[P1 - kPro + lane] = dst[opos - kPro + lane];
                    LDS_FENCE();
                }
                pf.add(5, tp);
                continue;
            }
#ifndef K1S_NOPREFETCH
            if (t0 + cnt < c) { const uint32_t tn = t0 + cnt + lane; nrec = rec_at(tn < c ? tn : c - 1u); have = true; }
#endif
            pf.add(0, tp); pf.count(6);
            const bool act = uint32_t(lane) < cnt;
            const uint32_t T = rdl(incl, cnt - 1);
            const uint32_t P0 = 64u + uint32_t(uintptr_t(dst + opos) & 15u);
            const uint32_t outl = incl - sz, mrel = outl + ll;
            if (__ballot(act && (off == 0u || off > opos + mrel))) { failed = true; break; }
            // ---- strings into the staging buffer: the first 32 bytes of every string by its own lane, then the rest of the long ones
            // four at a time, sixteen lanes and 32 bytes per lane each (a string a lane can hold is at most 528 bytes long: one step
            // per string, and four strings' loads in flight together - a lane looping over its own string makes a trip to memory per
            // 32 bytes: 14 K clk per step on audio-like data)
            auto copy_strings = [&](cgbyte* base, uint32_t srcpos, uint32_t dstrel, uint32_t len, bool on) {
                {
                    const uint32_t l1 = len < 32u ? len : 32u, pd = P0 + dstrel;
                    uint32_t R[9];
                    load_phase(base, srcpos, pd & 3u, R, on, l1);
                    or_store(st, pd, R, l1, on);
                }
                unsigned long long lm = __ballot(on && len > 32u);
                while (lm) {
                    uint32_t q[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) { q[g] = lm ? uint32_t(__builtin_ctzll(lm)) : 64u; lm &= lm - 1; }
                    const uint32_t g = uint32_t(lane) >> 4, i = uint32_t(lane) & 15u;
                    const uint32_t lq = g == 0 ? q[0] : g == 1 ? q[1] : g == 2 ? q[2] : q[3];
                    const uint32_t sp_ = uint32_t(__builtin_amdgcn_ds_bpermute(int(lq << 2), int(srcpos)));
                    const uint32_t dr_ = uint32_t(__builtin_amdgcn_ds_bpermute(int(lq << 2), int(dstrel)));
                    const uint32_t ln_ = uint32_t(__builtin_amdgcn_ds_bpermute(int(lq << 2), int(len)));
                    const uint32_t o = 32u + 32u * i;
                    const bool on2 = lq < 64u && o < ln_;
                    const uint32_t l2 = on2 ? (ln_ - o < 32u ? ln_ - o : 32u) : 0u, pd = P0 + dr_ + o;
                    uint32_t R[9];
                    load_phase(base, sp_ + o, pd & 3u, R, on2, l2);
                    or_store(st, pd, R, l2, on2);
                }
            };
            // ---- literals, from the stream
            copy_strings(s, lsrc, outl, ll, act && ll > 0u);
            pf.add(1, tp);
            // ---- matches.  M1: sources that end in front of the batch, from memory
            const int x = int(mrel) - int(off);                         // source, relative to the batch's first byte
#ifdef K1S_NODEPS
            // CEILING build (tools/ubench/seq_copy.py; never the product): the same sequences and the same copies - every match's
            // bytes fetched and placed - but NO dependency is honoured: a match's source is read from memory whatever state it is in
            // (no wait for the flush before, no order between the matches of a step).  The output is wrong wherever a source was not
            // there yet; the time is what this engine would take if dependencies were free.
            const bool m1 = act;
            if (__ballot(m1)) {
                copy_strings((cgbyte*)dst, x < 0 && uint32_t(-x) > opos ? 0u : opos + uint32_t(x), mrel, ml < 528u ? ml : 528u, m1 && int(opos) + x >= 0);
            }
            LDS_FENCE();
            const bool pend = false;
#else
            const bool m1 = act && x + int(ml) <= 0;
            if (__ballot(m1)) {
                VM_DRAIN();                                              // the flush of the batch before has to have arrived
                copy_strings((cgbyte*)dst, opos + uint32_t(x), mrel, ml, m1);
            }
            LDS_FENCE();
            const bool pend = act && !m1;
#endif
            pf.add(2, tp);
            // ---- M2: sources in the batch or its prologue, and overlapping matches: one sequence at a time in lane order, a byte
            // per lane, inside the staging buffer.  Up to 64 bytes with the source in the buffer: the short loop (an overlapping
            // match reads its period: byte i from i mod offset).  Everything else: the general one (bytes in front of the
            // prologue come from memory, long matches go in rounds).
            const bool tight = pend && ml <= 64u && x >= -kPro;
            unsigned long long m2 = __ballot(pend);
            const unsigned long long tmask = __ballot(tight), omask = __ballot(tight && off < ml);
            pf.count(7, __builtin_popcountll(m2)); pf.count(8, __builtin_popcountll(m2 & ~tmask));
            const uint32_t srcA = P0 + uint32_t(x), dstA = P0 + mrel;
            uint32_t recip = 0;
            if (omask) recip = uint32_t(65536.0f * __builtin_amdgcn_rcpf(float(off))) + 2u;   // floor(i * recip / 65536) = i / off for i < 64
            if (m2 & ~tmask) VM_DRAIN();
            // which sequences may share an LDS round trip with the pending one in front of them: both short-loop, not overlapping, and
            // this one's source clear of that one's destination (chains of near matches are a few sequences apart: database-like
            // records copy the record before them field by field, so neighbours are independent) - decided for all lanes at once
            unsigned long long pairm = 0;
            if (__builtin_popcountll(m2) >= 4) {
                const uint32_t lo = uint32_t(m2) & ((1u << (lane & 31)) - 1u) & (lane < 32 ? 0xFFFFFFFFu : 0xFFFFFFFFu);
                const uint32_t below_lo = lane < 32 ? lo : uint32_t(m2), below_hi = lane < 32 ? 0u : (uint32_t(m2 >> 32) & ((1u << (lane & 31)) - 1u));
                const uint32_t p = below_hi ? 63u - uint32_t(__builtin_clz(below_hi)) : (below_lo ? 31u - uint32_t(__builtin_clz(below_lo)) : 64u);
                const uint32_t pdst = uint32_t(__builtin_amdgcn_ds_bpermute(int(p << 2), int(dstA)));
                const uint32_t plen = uint32_t(__builtin_amdgcn_ds_bpermute(int(p << 2), int(ml)));
                const bool simple = tight && off >= ml;
                const unsigned long long simm = __ballot(simple);
                const bool psimple = p < 64u && ((simm >> p) & 1ull);
                pairm = __ballot(simple && psimple && (srcA + ml <= pdst || srcA >= pdst + plen));
            }
            uint8_t* const dummy = st + kStage - 8;                       // where lanes beyond a sequence's length read and write
            while (m2) {
                const uint32_t l = uint32_t(__builtin_ctzll(m2));
                m2 &= m2 - 1;
                if ((tmask >> l) & 1) {
                    const uint32_t sA = rdl(srcA, l), dA = rdl(dstA, l), n = rdl(ml, l);
                    const uint32_t l2 = m2 ? uint32_t(__builtin_ctzll(m2)) : 0u;
                    if (m2 && ((pairm >> l2) & 1)) {                       // l2's pending predecessor is l
                        m2 &= m2 - 1;
                        const uint32_t sB = rdl(srcA, l2), dB = rdl(dstA, l2), nB = rdl(ml, l2);
                        const bool oa = uint32_t(lane) < n, ob = uint32_t(lane) < nB;
                        const uint8_t va = *(oa ? st + sA + lane : dummy), vb = *(ob ? st + sB + lane : dummy);
                        *(oa ? st + dA + lane : dummy) = va;
                        *(ob ? st + dB + lane : dummy + 1) = vb;
                        pf.count(10);
                        continue;
                    }
                    uint32_t idx = uint32_t(lane);
                    if ((omask >> l) & 1) { const uint32_t o = rdl(off, l), M = rdl(recip, l); idx = uint32_t(lane) - ((uint32_t(lane) * M) >> 16) * o; }
                    if (uint32_t(lane) < n) { const uint8_t v = st[sA + idx]; st[dA + lane] = v; }
                    continue;
                }
                const int m = int(rdl(mrel, l)), o = int(rdl(off, l)), n = int(rdl(ml, l));
                if (o >= 64 || o >= n) {
                    for (int k0 = 0; k0 < n; k0 += 64) {
                        const int i = k0 + lane;
                        if (i < n) {
                            const int sx = m - o + i;
                            const uint8_t v = sx < -kPro ? dst[int(opos) + sx] : st[int(P0) + sx];
                            st[int(P0) + m + i] = v;
                        }
                        LDS_FENCE();
                    }
                } else {
                    // periodic: every byte derives from [m - o, m), which is final
                    int P = o; while (P < 64) P <<= 1;
                    int r = lane; for (int tt = P; tt >= o; tt >>= 1) if (r >= tt) r -= tt;      // lane mod o
                    int c64 = 64; for (int tt = P; tt >= o; tt >>= 1) if (c64 >= tt) c64 -= tt;  // 64 mod o
                    for (int k0 = 0; k0 < n; k0 += 64) {
                        const int i = k0 + lane;
                        if (i < n) {
                            const int sx = m - o + r;
                            const uint8_t v = sx < -kPro ? dst[int(opos) + sx] : st[int(P0) + sx];
                            st[int(P0) + m + i] = v;
                        }
                        r += c64; if (r >= o) r -= o;
                    }
                    LDS_FENCE();
                }
            }
            LDS_FENCE();
            pf.add(3, tp);
            // ---- flush: aligned 16-byte stores; then the next batch's prologue, and the buffer zeroed again
#ifndef K1S_NOPREFETCH
            asm volatile("" : "+v"(nrec.x), "+v"(nrec.y));               // (the next step's records: waited for here)
#endif
            {
                gbyte* g = dst + opos;
                const uint32_t head = min(T, (16u - uint32_t(uintptr_t(g) & 15u)) & 15u);
                if (uint32_t(lane) < head) g[lane] = st[P0 + lane];
                uint32_t k0 = head;
                for (; k0 + 16u <= T; k0 += 1024u) {
                    const uint32_t a = k0 + 16u * lane;
                    if (a + 16u <= T) st16g(g + a, *reinterpret_cast<const u32x4*>(st + P0 + a));
                }
                const uint32_t body = head + ((T - head) & ~15u);
                if (body + lane < T) g[body + lane] = st[P0 + body + lane];
                uint32_t pb = 0;
                if (lane < kPro) pb = st[P0 + T - kPro + lane];
                LDS_FENCE();
                for (uint32_t a = 32u + 16u * lane; a < 64u + 16u + T + 48u && a + 16u <= uint32_t(kStage); a += 1024u)
                    *reinterpret_cast<u32x4*>(st + a) = u32x4{0, 0, 0, 0};
                LDS_FENCE();
                const uint32_t P1 = 64u + uint32_t(uintptr_t(g + T) & 15u);
                if (lane < kPro) st[P1 - kPro + lane] = uint8_t(pb);
                LDS_FENCE();
            }
            opos += T; t0 += cnt;
            pf.add(4, tp);
        }
    }
    pf.dump(meta, kMetaProf + 24, lane);
    if (lane == 0) {
        if (failed) blocks[b].result = lz4par::kRetryCode;
        else { meta[kMetaResIp] = res_ip; meta[kMetaResOp] = opos; blocks[b].result = kResumeCode; }
    }
}


#ifdef FOURMC_RESEARCH
#include "../../tools/research/lz4_seg_exec2.inc"
#endif

} // namespace

extern "C" size_t fourmc_lz4_seg_work_bytes(uint32_t n) { return size_t(n) * lz4seg::kWsWords * 4u; }

// Blocks per launch pair: the workspace is sized for the largest stream a block can hold (11.3 MB of records per block), so a
// launch is cut into pieces whose workspace stays below 40 % of the device's memory (8192 blocks = 92 GB on a 288 GB MI355X).
// FOURMC_SEG_BATCH overrides.  (A caller whose lease fails halves the pieces for its own call: fourmc_lz4_decode_plan.)
static std::atomic<uint32_t> g_seg_batch{0};
extern "C" uint32_t fourmc_lz4_seg_batch(void)
{
    uint32_t v = g_seg_batch.load(std::memory_order_relaxed);
    if (v == 0) {
        uint32_t b = 8192;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) {
            const size_t fit = tot / 10 * 4 / (size_t(lz4seg::kWsWords) * 4u);
            if (fit < b) b = fit < 1 ? 1u : uint32_t(fit);
        }
        if (const char* e = getenv("FOURMC_SEG_BATCH")) { const long x = atol(e); if (x > 0) b = uint32_t(x); }
        g_seg_batch.store(b, std::memory_order_relaxed);
        v = b;
    }
    return v;
}

// the walk alone: the group executor (lz4_ring.hip) runs on its records too
extern "C" hipError_t fourmc_launch_lz4_seg_walk(const void* d_src, fourmc_block* d_blocks, uint32_t n, int container_mode,
                                                 void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_seg_walk_kernel, dim3(n), dim3(64), 0, stream, static_cast<const uint8_t*>(d_src), d_blocks, n,
                       container_mode, static_cast<uint32_t*>(d_work));
    return hipGetLastError();
}

extern "C" hipError_t fourmc_launch_lz4_seg(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                            int container_mode, void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_seg_walk_kernel, dim3(n), dim3(64), 0, stream, static_cast<const uint8_t*>(d_src), d_blocks, n,
                       container_mode, static_cast<uint32_t*>(d_work));
#ifdef FOURMC_RESEARCH
    // FOURMC_SEG_EXEC=2 (research build): the pipelined executor of tools/research/lz4_seg_exec2.inc
    static const int which = [] { const char* e = getenv("FOURMC_SEG_EXEC"); return e ? atoi(e) : 1; }();
    if (which == 2) {
        hipLaunchKernelGGL(lz4_seg_exec2_kernel, dim3(n), dim3(64), 0, stream, static_cast<const uint8_t*>(d_src),
                           static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, static_cast<uint32_t*>(d_work));
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(lz4_seg_exec_kernel, dim3(n), dim3(64), 0, stream, static_cast<const uint8_t*>(d_src),
                       static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, static_cast<uint32_t*>(d_work));
    return hipGetLastError();
}
