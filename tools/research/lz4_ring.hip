// tools/research/lz4_ring.hip - K1g: group executor of the LZ4 block decode on gfx950 (wave64), round 6.
//
// Replaces LZ4_decompress_safe(in, out, csize, usize) per block (native/4mc.c:661, native/jniDecompressor.c:88 ->
// native/lz4/lz4.c:2345-2350 -> :1936-2339) for every block the exact walker (lz4_decode.hip) does not have to see.  Where the
// tokens are is found by lz4_seg.hip's walk kernel (sequence records in the block's workspace slot, lz4seg.h); this file executes
// them.
//
// ONE WORKGROUP OF 512 THREADS PER BLOCK, two per CU, and the whole 64 KiB LZ4 window of the block in an LDS RING: a match never
// goes to memory, the stream is read once and the output written once.  A group step takes the next <= 512 sequences, ONE PER
// THREAD:
//   records -> sizes -> prefix sum over the workgroup (DPP inside a wave, eight totals through LDS) -> every sequence knows where its
//     bytes go; the step takes the longest prefix of its sequences that fits kStepB bytes of the ring;
//   LITERALS: a lane loads its run from the stream with unaligned 16-byte loads at the address that puts the bytes in the phase of
//     the destination dwords, masks head and tail and ORs <= 9 dwords into the ring (ds_or_b32: lanes - of any wave - that share a
//     dword cannot lose each other's bytes; the step's region of the ring is zeroed beforehand);
//   MATCHES, in ROUNDS: beside the ring there is one VALID BIT per byte of the step's region, set by whoever writes the byte (after
//     the bytes: LDS operations of a wave execute in order).  A match goes as soon as the bits of its source are all set (sources in
//     front of the step are final by construction): aligned dword reads around the source, v_alignbyte to the destination's phase,
//     OR into the ring, set its own bits.  A round = every pending lane tests and, if ready, copies; one barrier; until no lane of
//     the workgroup is pending.  The number of rounds is the depth of the step's dependency chains IN BYTES WRITTEN, not the number
//     of near matches (lz4_seg.hip runs those one sequence at a time, 256 clk each: the serial part of its steps).
//     Overlapping matches (offset < length) wait for [match - offset, match) and then double what they have (period preserved);
//     strings beyond 32 bytes are finished sixteen lanes per string;
//   FLUSH: whole 16-byte pieces by ADDRESS (ring index = output position + phase of the output pointer, mod the ring), the piece a
//     step ends in waits for the next step.
//   Sequences a record cannot hold (literal run >= 511, match length with more than two extension bytes) are taken by the whole
//     workgroup, 8 KiB of ring per round.
//   The last 64 stream bytes / 128 output bytes of a block and anything irregular go to the exact walker (kResume / kRetry), as for
//     lz4_seg.hip: accept / reject set and return codes stay the reference's.
// All byte work; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"
#include "lz4seg.h"
#include "lz4ring.h"

namespace {

using namespace lz4seg;
using namespace lz4ring;

typedef __attribute__((address_space(1))) uint8_t gbyte;
typedef __attribute__((address_space(1))) const uint8_t cgbyte;
typedef __attribute__((address_space(1))) uint32_t gword;
typedef __attribute__((address_space(1))) const uint32_t cgword;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld4u(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32_u*>(p); }
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
// LDS operations of one wave execute in order; what has to be stopped is the COMPILER moving one access across another
#define LDS_ORDER() asm volatile("" ::: "memory")
// workgroup barrier for LDS traffic only (__syncthreads() also waits for the wave's outstanding global loads and stores)
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define VM_DRAIN()   asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#ifdef K1G_PROF
struct Prof {
    unsigned long long t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void add(int i, unsigned long long& since) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t[i] += n - since; since = n; }
    __device__ __forceinline__ void count(int i, unsigned long long n = 1) { t[i] += n; }
    __device__ __forceinline__ void dump(gword* meta, uint32_t at, uint32_t tid) const { if (tid == 0) for (int i = 0; i < 12; i++) { meta[at + 2 * i] = uint32_t(t[i]); meta[at + 2 * i + 1] = uint32_t(t[i] >> 32); } }
};
#else
struct Prof {
    __device__ __forceinline__ unsigned long long now() const { return 0; }
    __device__ __forceinline__ void add(int, unsigned long long&) {}
    __device__ __forceinline__ void count(int, unsigned long long = 1) {}
    __device__ __forceinline__ void dump(gword*, uint32_t, uint32_t) const {}
};
#endif

__device__ __forceinline__ bool eligible(const fourmc_block& blk)
{ return blk.src_len >= kMinSrc && blk.src_len <= kMaxSrc && blk.dst_cap >= kMinCap && blk.dst_cap <= lz4par::kDstMax; }

__device__ __forceinline__ uint32_t wrap_up(uint32_t i) { return i >= kR ? i - kR : i; }                  // i < 2 kR
__device__ __forceinline__ uint32_t wrap_rel(uint32_t rb, int x)                                           // rb + x, x in (-kR, kR)
{ int i = int(rb) + x; if (i < 0) i += int(kR); if (i >= int(kR)) i -= int(kR); return uint32_t(i); }

// ================================================================================================ strings into the ring
// OR `len` (1..32) string bytes into the ring at byte index pd (< kR).  Rg[0..8] holds the string phase-aligned to the destination:
// string byte b sits at byte (da + b) of Rg, da = pd & 3; bytes of Rg outside the string are arbitrary.  The ring's first kMirror
// bytes are kept twice (again behind its end), so that reads never have to wrap inside a string.
__device__ __forceinline__ void ring_or(uint32_t* ringw, uint32_t pd, const uint32_t (&Rg)[9], uint32_t len, bool on)
{
    const uint32_t da = pd & 3u, end = da + len, last = (end + 3u) / 4u - 1u, tb = end & 3u;
    const uint32_t hmask = 0xFFFFFFFFu << (8u * da), tmask = tb ? ((1u << (8u * tb)) - 1u) : 0xFFFFFFFFu;
    const uint32_t w0 = pd >> 2;
    if (__builtin_expect(__ballot(on && (w0 + 9u > kR / 4u || w0 < kMirror / 4u)) != 0, 0)) {
        // a string at the ring's end or in its first bytes: canonical index, and the mirror
#pragma unroll
        for (uint32_t j = 0; j < 9; j++) {
            uint32_t v = Rg[j];
            if (j == 0) v &= hmask;
            v = j == last ? (v & tmask) : v;
            if (on && j <= last) {
                uint32_t idx = w0 + j; if (idx >= kR / 4u) idx -= kR / 4u;
                __hip_atomic_fetch_or(ringw + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (idx < kMirror / 4u) __hip_atomic_fetch_or(ringw + idx + kR / 4u, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }
    uint32_t* w = ringw + w0;
#pragma unroll
    for (uint32_t j = 0; j < 5; j++) {
        uint32_t v = Rg[j];
        if (j == 0) v &= hmask;
        v = j == last ? (v & tmask) : v;
        if (on && j <= last) __hip_atomic_fetch_or(w + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (__ballot(on && last >= 5u)) {
#pragma unroll
        for (uint32_t j = 5; j < 9; j++) {
            uint32_t v = Rg[j];
            v = j == last ? (v & tmask) : v;
            if (on && j <= last) __hip_atomic_fetch_or(w + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
// valid bits [v, v + len), len 1..32
__device__ __forceinline__ void vb_or(uint32_t* vb, uint32_t v, uint32_t len, bool on)
{
    const uint32_t w = v >> 5, sh = v & 31u, m = len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u);
    const uint32_t lo = m << sh, hi = sh ? (m >> (32u - sh)) : 0u;
    if (on) {
        __hip_atomic_fetch_or(vb + w, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (hi) __hip_atomic_fetch_or(vb + w + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// are the bits [a, e) all set?  e - a in 1..32
__device__ __forceinline__ bool vb_all(const uint32_t* vb, uint32_t a, uint32_t e)
{
    const uint32_t w = a >> 5, sh = a & 31u, n = e - a;
    const uint32_t lo = vb[w], hi = vb[w + 1];
    const uint32_t bits = __builtin_amdgcn_alignbit(hi, lo, sh);
    const uint32_t m = n >= 32u ? 0xFFFFFFFFu : ((1u << n) - 1u);
    return (bits & m) == m;
}
// nine dwords from the stream so that the byte at `base + a` lands on byte `da` of Rg[0] (as lz4_seg.hip's)
__device__ __forceinline__ void load_phase_g(cgbyte* base, uint32_t a, uint32_t da, uint32_t (&Rg)[9], bool on, uint32_t len)
{
    const bool far1 = __ballot(on && da + len > 16u) != 0, far2 = __ballot(on && da + len > 32u) != 0;
#pragma unroll
    for (int j = 0; j < 9; j++) Rg[j] = 0;
    if (on) {
        if (a >= da) {
            cgbyte* p = base + (a - da);
            const u32x4 v0 = ld16u_g(p);
            Rg[0] = v0.x; Rg[1] = v0.y; Rg[2] = v0.z; Rg[3] = v0.w;
            if (far1) { const u32x4 v1 = ld16u_g(p + 16); Rg[4] = v1.x; Rg[5] = v1.y; Rg[6] = v1.z; Rg[7] = v1.w; }
            if (far2) Rg[8] = ld4u(p + 32);
        } else {
            uint32_t L[9];
            const u32x4 v0 = ld16u_g(base), v1 = ld16u_g(base + 16); const uint32_t v2 = ld4u(base + 32);
            L[0] = v0.x; L[1] = v0.y; L[2] = v0.z; L[3] = v0.w; L[4] = v1.x; L[5] = v1.y; L[6] = v1.z; L[7] = v1.w; L[8] = v2;
            const uint32_t k = 4u - (da - a);
#pragma unroll
            for (int j = 0; j < 9; j++) Rg[j] = __builtin_amdgcn_alignbyte(L[j], j ? L[j - 1] : 0u, k);
        }
    }
}
// the same from the ring: the byte at ring index ps lands on byte `da` of Rg[0].  Aligned dword reads around the source (an
// LDS read that is not naturally aligned is served a lane at a time: 64 clk), v_alignbyte to the phase.
__device__ __forceinline__ void load_phase_r(const uint32_t* ringw, uint32_t ps, uint32_t da, uint32_t (&Rg)[9], bool on, uint32_t len)
{
    int q = int(ps) - int(da); if (q < 0) q += int(kR);
    const uint32_t w = uint32_t(q) >> 2, sh = uint32_t(q) & 3u;
    const bool far1 = __ballot(on && sh + da + len > 24u) != 0;
    uint32_t D[10];
#pragma unroll
    for (int j = 0; j < 10; j++) D[j] = 0;
    if (on) {
#pragma unroll
        for (int j = 0; j < 6; j++) D[j] = ringw[w + j];
        if (far1) {
#pragma unroll
            for (int j = 6; j < 10; j++) D[j] = ringw[w + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 9; j++) Rg[j] = __builtin_amdgcn_alignbyte(D[j + 1], D[j], sh);
}
// 16 bytes at byte index i of an LDS array of dwords, any alignment
__device__ __forceinline__ u32x4 lds_read16(const uint32_t* base, uint32_t i)
{
    const uint32_t w = i >> 2, sh = i & 3u;
    const uint32_t d0 = base[w], d1 = base[w + 1], d2 = base[w + 2], d3 = base[w + 3], d4 = base[w + 4];
    return u32x4{__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                 __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
}

// ================================================================================================ EXEC kernel
__global__ __launch_bounds__(kThreads)
void lz4_ring_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                          int container_mode, uint32_t* ws)
{
    __shared__ __attribute__((aligned(16))) uint32_t ringw[(kR + kMirror) / 4];
    __shared__ uint32_t vb[kVbWords];
    __shared__ uint32_t wsum[16];                 // [w] bytes of wave w's sequences (in front of its first stop), [8 + w] wave w has a stop
    __shared__ uint32_t info[8];
    __shared__ __attribute__((aligned(16))) uint32_t patw[kPatBytes / 4];
    constexpr uint32_t I_CNT = 0, I_T = 1, I_ESC0 = 2, I_PF = 3 /* .. 5 */, I_FAIL = 6;
    uint8_t* const ringb = reinterpret_cast<uint8_t*>(ringw);
    uint8_t* const patb = reinterpret_cast<uint8_t*>(patw);

    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = rfl(tid >> 6);
    if (container_mode && blk.src_len == blk.dst_cap) {                 // stored block (native/4mc.c:635-642)
        const uint8_t* sp = src_base + blk.src_off; uint8_t* dp = dst_base + blk.dst_off;
        const uint32_t n = blk.src_len;
        const uint32_t head = min(n, uint32_t((16u - uint32_t(uintptr_t(dp) & 15u)) & 15u));
        if (tid < head) dp[tid] = sp[tid];
        uint32_t k = head + 16u * tid;
        for (; k + 16u <= n; k += 16u * kThreads) *reinterpret_cast<uint4*>(dp + k) = ld16u(sp + k);
        const uint32_t body = head + ((n - head) & ~15u);
        if (body + tid < n) dp[body + tid] = sp[body + tid];
        if (tid == 0) blocks[b].result = int(blk.src_len);
        return;
    }
    gword* meta = (gword*)(ws + size_t(b) * kWsWords);
    if (!eligible(blk) || rfl(meta[kMetaStatus]) != 1u) { if (tid == 0) blocks[b].result = lz4par::kRetryCode; return; }
    cgbyte* s = (cgbyte*)(src_base + blk.src_off);
    gbyte* dst = (gbyte*)(dst_base + blk.dst_off);
    const uint32_t cap = blk.dst_cap, limit = blk.src_len - kMargin, olimit = cap - kOMargin;
    const uint32_t nlive = rfl(meta[kMetaNLive]);
    uint32_t res_ip = rfl(meta[kMetaTailIp]);
    const uint32_t ph = uint32_t(uintptr_t(dst) & 15u);
    uint32_t opos = 0;            // output bytes produced
    uint32_t rb = ph;             // ring index of output position opos: (opos + ph) mod kR
    uint32_t Fa = ph, rf = ph;    // flushed up to (output position + ph), and its ring index
    bool failed = false, cut = false;
    uint32_t rnd = 0;             // rounds so far (the pending flags rotate through three words: the one a round uses was zeroed during the round before)
    Prof pf; unsigned long long tp = pf.now();

    if (tid < 8) info[tid] = 0;
    // ---- flush: what lies in [Fa, opos + ph) leaves the ring, whole 16-byte pieces by address (all of it when `final`)
    auto flush = [&](bool final) {
        const uint32_t E = opos + ph, hi = final ? E : (E & ~15u);
        if (hi <= Fa) return;
        const uint32_t lo = Fa, hl = min(hi, (lo + 15u) & ~15u);
        if (tid < hl - lo) dst[lo - ph + tid] = ringb[wrap_up(rf + tid)];          // (only a block's first bytes, or the last flush)
        const uint32_t be = hi & ~15u;
        if (be > hl) {
            const uint32_t a = hl + 16u * tid;
            if (a < be) st16g(dst + (a - ph), *reinterpret_cast<const u32x4*>(ringb + wrap_up(rf + (a - lo))));
        }
        if (final && be >= hl && hi > be && tid < hi - be) dst[be - ph + tid] = ringb[wrap_up(rf + (be - lo) + tid)];
        rf = wrap_up(rf + (hi - lo)); Fa = hi;
    };
    // ---- the ring in front of a step: [rb, rb + kZero + 15) zero (and its mirror)
    auto zero_ahead = [&]() {
        const uint32_t c = (rb + 15u) & ~15u;
        if (tid < c - rb) { const uint32_t i = rb + tid; ringb[i] = 0; if (i < kMirror) ringb[i + kR] = 0; }     // rb + tid < c <= kR
        const uint32_t z = wrap_up(c + 16u * tid);                                                               // c <= kR: c + 8176 < 2 kR
        *reinterpret_cast<u32x4*>(ringb + z) = u32x4{0, 0, 0, 0};
        if (z < kMirror) *reinterpret_cast<u32x4*>(ringb + z + kR) = u32x4{0, 0, 0, 0};
    };

    for (uint32_t li = 0; li < nlive && !failed && !cut; li++) {
        cgword* e = (cgword*)(meta + kMetaLive + 4 * li);
        const uint32_t area = rfl(e[0]), f = rfl(e[1]), k = rfl(e[2]), c = rfl(e[3]);
        cgword* FL = (cgword*)(meta + area);
        uint32_t t0 = 0;
        auto rec_at = [&](uint32_t t) -> u32x2 { return *reinterpret_cast<__attribute__((address_space(1))) const u32x2*>(FL + 2 * (t < f ? t : kFixCap + k + (t - f))); };
        while (t0 < c) {
            // ---- records and fields, one sequence per thread
            const uint32_t t = t0 + tid;
            const bool valid = t < c;
            u32x2 rec2 = u32x2{0, 0};
            if (valid) rec2 = rec_at(t);
            zero_ahead();
            if (tid < kVbWords) vb[tid] = 0;
            const uint32_t pos = rec2.x & kPosMask, ll = rec2.x >> kPosBits;
            const bool esc = valid && ll == kEscLL;
            const bool bad = !valid || esc;
            const uint32_t off = rec2.y & 0xFFFFu, ml = rec2.y >> 16;
            const uint32_t llx = ll < 15u ? 0u : 1u + (ll >= 270u);
            const uint32_t lsrc = pos + 1u + llx;
            const unsigned long long badm = __ballot(bad);
            const uint32_t ew = badm ? uint32_t(__builtin_ctzll(badm)) : 64u;            // first lane of the wave that stops the step
            const uint32_t sz = lane < ew ? ll + ml : 0u;
            const uint32_t incl = scan_add(sz);
            if (lane == 63) { wsum[wave] = incl; wsum[8 + wave] = ew < 64u ? 1u : 0u; }
            WG_BARRIER();                                                                // B1: totals, zeroed ring region and bits
            if (tid == 0) info[I_ESC0] = esc ? 1u : 0u;                                  // (read behind the rounds' barriers, written again behind the next B1)
            pf.add(0, tp);
            // ---- how many sequences the step takes: the longest prefix that fits
            const uint32_t lim = min(uint32_t(kStepB), olimit - opos);
            uint32_t base = 0, run = 0, wstar = 8; bool dead = false;
            {
                const uint32_t wv = wsum[lane & 15u];
#pragma unroll
                for (uint32_t w2 = 0; w2 < 8; w2++) {
                    const uint32_t tot = rdl(wv, w2), stp = rdl(wv, 8 + w2);
                    if (w2 == wave) base = run;
                    if (wstar == 8 && (stp || run + tot > lim)) wstar = w2;
                    run += tot;
                }
                dead = wstar < wave;
            }
            const bool act = !dead && lane < ew && base + incl <= lim;
            uint32_t cnt = 512, T = run;
            if (wstar < 8) {
                if (wave == wstar) {
                    const uint32_t cw = uint32_t(__builtin_popcountll(__ballot(act)));
                    const uint32_t Tw = cw ? base + rdl(incl, cw - 1u) : base;
                    if (lane == 0) { info[I_CNT] = 64u * wstar + cw; info[I_T] = Tw; }
                }
            }
            const uint32_t outl = base + incl - sz, mrel = outl + ll;
            if (__ballot(act && (off == 0u || off > opos + mrel))) { if (lane == 0) info[I_FAIL] = 1; }
            // ---- literals, from the stream
            auto rest_strings = [&](bool from_ring, unsigned long long lm, uint32_t srcpos, uint32_t dstrel, uint32_t len) {
                // the bytes beyond the first 32 of the strings in lm: four strings at a time, sixteen lanes and 32 bytes per lane each
                while (lm) {
                    uint32_t q[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) { q[g] = lm ? uint32_t(__builtin_ctzll(lm)) : 64u; lm &= lm - 1; }
                    const uint32_t g = lane >> 4, i = lane & 15u;
                    const uint32_t lq = g == 0 ? q[0] : g == 1 ? q[1] : g == 2 ? q[2] : q[3];
                    const uint32_t sp_ = uint32_t(__builtin_amdgcn_ds_bpermute(int(lq << 2), int(srcpos)));
                    const uint32_t dr_ = uint32_t(__builtin_amdgcn_ds_bpermute(int(lq << 2), int(dstrel)));
                    const uint32_t ln_ = uint32_t(__builtin_amdgcn_ds_bpermute(int(lq << 2), int(len)));
                    const uint32_t o = 32u + 32u * i;
                    const bool on2 = lq < 64u && o < ln_;
                    const uint32_t l2 = on2 ? (ln_ - o < 32u ? ln_ - o : 32u) : 0u;
                    const uint32_t pd = wrap_up(rb + dr_ + o);
                    uint32_t Rg[9];
                    if (from_ring) load_phase_r(ringw, wrap_rel(rb, int(sp_) + int(o)), pd & 3u, Rg, on2, l2);
                    else load_phase_g(s, sp_ + o, pd & 3u, Rg, on2, l2);
                    ring_or(ringw, pd, Rg, l2, on2);
                    LDS_ORDER();
                    vb_or(vb, dr_ + o, l2, on2);
                }
            };
            {
                const bool on = act && ll > 0u;
                if (__ballot(on)) {
                    const uint32_t l1 = ll < 32u ? ll : 32u, pd = wrap_up(rb + outl);
                    uint32_t Rg[9];
                    load_phase_g(s, lsrc, pd & 3u, Rg, on, l1);
                    ring_or(ringw, pd, Rg, l1, on);
                    LDS_ORDER();
                    vb_or(vb, outl, l1, on);
                    const unsigned long long lm = __ballot(on && ll > 32u);
                    if (lm) rest_strings(false, lm, lsrc, outl, ll);
                }
            }
            pf.add(1, tp);
            // ---- matches, in rounds
            const int x = int(mrel) - int(off);                          // source, relative to the step's first byte
            const bool ovl = off < ml;
            const uint32_t need = ovl ? off : ml;                        // source bytes that have to be there
            bool pend = act;
            uint32_t round = 0, wgfail = 0;
            for (;;) {
                LDS_ORDER();
                if (__ballot(pend)) {
                    const int a0 = x > 0 ? x : 0, e0 = x + int(need);
                    bool ready = pend;
                    if (pend && e0 > 0) {
                        if (e0 - a0 <= 32) ready = vb_all(vb, uint32_t(a0), uint32_t(e0));
                        else {
                            for (int p = a0; p < e0 && ready; p += 32) ready = vb_all(vb, uint32_t(p), uint32_t(min(p + 32, e0)));
                        }
                    }
                    if (__ballot(ready)) {
                        const uint32_t l1 = need < 32u ? need : 32u, pd = wrap_up(rb + mrel);
                        uint32_t Rg[9];
                        load_phase_r(ringw, wrap_rel(rb, x), pd & 3u, Rg, ready, l1);
                        ring_or(ringw, pd, Rg, l1, ready);
                        LDS_ORDER();
                        vb_or(vb, mrel, l1, ready);
                        const unsigned long long lm = __ballot(ready && !ovl && ml > 32u);
                        if (lm) rest_strings(true, lm, uint32_t(x), mrel, ml);
                        // overlapping matches: what is there is doubled (the distance stays a multiple of the offset)
                        uint32_t done = l1, dist = off;
                        while (__ballot(ready && ovl && done < ml)) {
                            LDS_ORDER();
                            const bool on = ready && ovl && done < ml;
                            if (dist < 32u && 2u * dist <= done + off) dist *= 2u;
                            uint32_t n = ml - done; n = n < 32u ? n : 32u; n = n < dist ? n : dist;
                            const uint32_t pd2 = wrap_up(rb + mrel + done);
                            uint32_t R2[9];
                            load_phase_r(ringw, wrap_rel(rb, int(mrel + done) - int(dist)), pd2 & 3u, R2, on, n);
                            ring_or(ringw, pd2, R2, n, on);
                            LDS_ORDER();
                            vb_or(vb, mrel + done, n, on);
                            if (on) done += n;
                        }
                        pend = pend && !ready;
                    }
                }
                const bool anyp = __ballot(pend) != 0;
                if (anyp && lane == 0) info[I_PF + rnd % 3u] = 1u;
                if (tid == 0) info[I_PF + (rnd + 1u) % 3u] = 0u;
                WG_BARRIER();
                const uint32_t pfv = rfl(info[I_PF + rnd % 3u]);
                rnd++;
                wgfail = rfl(info[I_FAIL]);
                if (pfv == 0u || wgfail) break;
                if (++round > kMaxRounds) { wgfail = 1; break; }
            }
            pf.add(2, tp); pf.count(6); pf.count(7, round + 1);
            if (wgfail) { failed = true; break; }
            if (wstar < 8) { cnt = rfl(info[I_CNT]); T = rfl(info[I_T]); }
            if (cnt == 0) {
                const uint32_t esc0 = rfl(info[I_ESC0]);
                const uint32_t p0 = rfl(rec_at(t0).x) & kPosMask;
                if (!esc0) { res_ip = p0; cut = true; break; }           // the output-side tail starts here
                // ---- one long sequence, by the whole workgroup (lz4.c:2175-2330 without the end-of-block cases)
                uint32_t tok = rfl(s[p0]), q = p0 + 1, L = tok >> 4, M = (tok & 15u) + 4u;
                if (L == 15u) for (;;) { if (q >= limit) { failed = true; break; } const uint32_t bb = rfl(s[q++]); L += bb; if (bb != 255u) break; }
                if (failed || L > (1u << 23) || q + L + 2 > limit) { failed = true; break; }
                const uint32_t mo = q + L, o16 = rfl(uint32_t(s[mo]) | (uint32_t(s[mo + 1]) << 8));
                uint32_t q2 = mo + 2;
                if ((tok & 15u) == 15u) for (;;) { if (q2 >= limit) { failed = true; break; } const uint32_t bb = rfl(s[q2++]); M += bb; if (bb != 255u) break; if (M > (1u << 23)) { failed = true; break; } }
                if (failed) break;
                if (uint64_t(opos) + L + M > uint64_t(olimit)) { res_ip = p0; cut = true; break; }
                if (o16 == 0 || o16 > opos + L) { failed = true; break; }
                // literals: pieces of 16 bytes by address, 8 KiB of ring per round
                for (uint32_t d0 = 0; d0 < L; ) {
                    const uint32_t n = min(L - d0, uint32_t(kStepB));
                    const uint32_t Ab = opos + ph, pa = (Ab & ~15u) + 16u * tid;           // this thread's piece [pa, pa + 16)
                    if (pa >= Ab && pa + 16u <= Ab + n) {
                        const u32x4 v = ld16u_g(s + q + d0 + (pa - Ab));
                        const uint32_t z = wrap_up(rb + (pa - Ab));
                        *reinterpret_cast<u32x4*>(ringb + z) = v;
                        if (z < kMirror) *reinterpret_cast<u32x4*>(ringb + z + kR) = v;
                    } else if (pa + 16u > Ab && pa < Ab + n) {
                        for (uint32_t j = 0; j < 16u; j++) {
                            const uint32_t a = pa + j;
                            if (a >= Ab && a < Ab + n) { const uint32_t z = wrap_up(rb + (a - Ab)); const uint8_t v = s[q + d0 + (a - Ab)]; ringb[z] = v; if (z < kMirror) ringb[z + kR] = v; }
                        }
                    }
                    opos += n; rb = wrap_up(rb + n); d0 += n;
                    WG_BARRIER();
                    flush(false);
                }
                // match
                const uint32_t m0 = opos;
                if (o16 < kPatMax) {
                    // short period: every byte of the match is a byte of the period in front of it
                    if (tid < kPatBytes) patb[tid] = ringb[wrap_rel(rb, -int(o16) + int(tid % o16))];
                    WG_BARRIER();
                }
                for (uint32_t d0 = 0; d0 < M; ) {
                    uint32_t n = min(M - d0, uint32_t(kStepB));
                    if (o16 >= kPatMax) n = min(n, o16);                                   // the source of a round lies in front of it
                    const uint32_t Ab = opos + ph, pa = (Ab & ~15u) + 16u * tid;
                    if (pa >= Ab && pa + 16u <= Ab + n) {
                        u32x4 v;
                        if (o16 < kPatMax) v = lds_read16(patw, (opos - m0 + (pa - Ab)) % o16);
                        else v = lds_read16(ringw, wrap_rel(rb, int(pa - Ab) - int(o16)));
                        const uint32_t z = wrap_up(rb + (pa - Ab));
                        *reinterpret_cast<u32x4*>(ringb + z) = v;
                        if (z < kMirror) *reinterpret_cast<u32x4*>(ringb + z + kR) = v;
                    } else if (pa + 16u > Ab && pa < Ab + n) {
                        for (uint32_t j = 0; j < 16u; j++) {
                            const uint32_t a = pa + j;
                            if (a >= Ab && a < Ab + n) {
                                const uint32_t z = wrap_up(rb + (a - Ab));
                                const uint8_t v = o16 < kPatMax ? patb[(opos - m0 + (a - Ab)) % o16] : ringb[wrap_rel(rb, int(a - Ab) - int(o16))];
                                ringb[z] = v; if (z < kMirror) ringb[z + kR] = v;
                            }
                        }
                    }
                    opos += n; rb = wrap_up(rb + n); d0 += n;
                    WG_BARRIER();
                    flush(false);
                }
                t0 += 1; pf.count(9);
                WG_BARRIER();
                continue;
            }
            opos += T; rb = wrap_up(rb + T); t0 += cnt;
            flush(false);
            pf.add(3, tp);
        }
    }
    WG_BARRIER();
    flush(true);
    pf.dump(meta, kMetaProf + 24, tid);
    if (tid == 0) {
        if (failed) blocks[b].result = lz4par::kRetryCode;
        else { meta[kMetaResIp] = res_ip; meta[kMetaResOp] = opos; blocks[b].result = kResumeCode; }
    }
}

} // namespace

extern "C" hipError_t fourmc_launch_lz4_seg_walk(const void* d_src, fourmc_block* d_blocks, uint32_t n, int container_mode,
                                                 void* d_work, hipStream_t stream);

extern "C" hipError_t fourmc_launch_lz4_ring(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                             int container_mode, void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipError_t e = fourmc_launch_lz4_seg_walk(d_src, d_blocks, n, container_mode, d_work, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lz4_ring_exec_kernel, dim3(n), dim3(lz4ring::kThreads), 0, stream, static_cast<const uint8_t*>(d_src),
                       static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, static_cast<uint32_t*>(d_work));
    return hipGetLastError();
}
