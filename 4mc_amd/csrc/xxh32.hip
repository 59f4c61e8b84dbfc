// 4mc_amd/csrc/xxh32.hip — K5: batched XXH32 (seeded, 32-bit) over block payloads on gfx950.
//
// Replaces the per-block XXH32(payload, n, 0) calls of the reference container loops
// (native/4mc.c:311,323 on encode, :637,645 on decode; JNI xxhash32 native/jniCompressor.c:183)
// -> native/lz4/xxhash.c:392-415 (stripe loop :352-389, tail :291-345).
//
// XXH32 is four independent multiply-rotate chains over 16-byte stripes; a chain step depends on
// the previous one and the round function is not associative, so there is no intra-block
// parallel prefix, and its two 32-bit multiplies run at quarter rate whatever the number of active
// lanes.  So lanes are not spent on streaming: four lanes are the four accumulators of one block and
// a wavefront carries G blocks side by side (G = 1..16, chosen at launch so that every SIMD still
// has a wave).  Each lane reads its own 4-byte words straight from memory (unaligned dword loads, the
// next 64 rounds in flight while 64 are consumed); the four lanes of a block cover 16 contiguous bytes per round, 128 bytes - one
// cache line - per unrolled iteration.  Bound: the multiply chain, then HBM read of `len` bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"

namespace {

constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;

struct __attribute__((packed, aligned(1))) W4 { uint32_t v; };
__device__ __forceinline__ uint32_t ldw(const uint8_t* p) { return reinterpret_cast<const W4*>(p)->v; }
__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t round1(uint32_t acc, uint32_t x) { return rotl(acc + x * P2, 13) * P1; }

// the stripes from s on (per-lane reads), then the merge of the four accumulators and the tail (xxhash.c:291-345);
// returns the digest in every lane of the group
__device__ uint32_t xxh32_finish(const uint8_t* p, uint32_t len, uint32_t seed, uint32_t acc, uint32_t s, int lane)
{
    const uint32_t nstripes = len >> 4;
    const uint8_t* q = p + 4 * (lane & 3) + 16 * size_t(s);
    for (; s < nstripes; s++, q += 16) acc = round1(acc, ldw(q));
    uint32_t h;
    if (len >= 16) {
        const int g0 = lane & ~3;
        const uint32_t a0 = __shfl(acc, g0), a1 = __shfl(acc, g0 + 1), a2 = __shfl(acc, g0 + 2), a3 = __shfl(acc, g0 + 3);
        h = rotl(a0, 1) + rotl(a1, 7) + rotl(a2, 12) + rotl(a3, 18);
    } else {
        h = seed + P5;
    }
    h += len;
    const uint8_t* t = p + (len & ~15u);               // tail, < 16 bytes (xxhash.c:291-345)
    const uint32_t tail = len & 15;
    uint32_t i = 0;
    for (; i + 4 <= tail; i += 4) h = rotl(h + ldw(t + i) * P3, 17) * P4;
    for (; i < tail; i++) h = rotl(h + uint32_t(t[i]) * P5, 11) * P1;
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

// lanes 4g..4g+3 hash block g of this wave; returns the digest in every lane of the group
__device__ uint32_t xxh32_group(const uint8_t* p, uint32_t len, uint32_t seed, int lane)
{
    const int chain = lane & 3;
    const uint32_t nstripes = len >> 4;
    uint32_t acc = (chain == 0) ? seed + P1 + P2 : (chain == 1) ? seed + P2 : (chain == 2) ? seed : seed - P1;
    const uint8_t* q = p + 4 * chain;
    uint32_t s = 0;
    // xxhash.c:352-389.  A round is ~40 clocks of dependent arithmetic, a read from HBM ~2000: the words of the next
    // 32 rounds are read while the current 32 are consumed (two register banks, ping-pong).
    constexpr int kB = 64;
    if (nstripes >= uint32_t(kB)) {
        uint32_t xa[kB], xb[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) xa[u] = ldw(q + 16 * u);
        q += 16 * kB; s = kB;                                           // s = stripes whose words are loaded
        for (;;) {
            const bool more_b = s + kB <= nstripes;
            if (more_b) {
#pragma unroll
                for (int u = 0; u < kB; u++) xb[u] = ldw(q + 16 * u);
                q += 16 * kB; s += kB;
            }
#pragma unroll
            for (int u = 0; u < kB; u++) acc = round1(acc, xa[u]);
            if (!more_b) break;
            const bool more_a = s + kB <= nstripes;
            if (more_a) {
#pragma unroll
                for (int u = 0; u < kB; u++) xa[u] = ldw(q + 16 * u);
                q += 16 * kB; s += kB;
            }
#pragma unroll
            for (int u = 0; u < kB; u++) acc = round1(acc, xb[u]);
            if (!more_a) break;
        }
    }
    return xxh32_finish(p, len, seed, acc, s, lane);
}

__global__ __launch_bounds__(64)
void xxh32_kernel(const uint8_t* __restrict__ base, fourmc_block* blocks, uint32_t nblocks,
                  uint32_t seed, int mode, uint32_t groups)
{
    const int lane = threadIdx.x;
    const uint32_t g = uint32_t(lane) >> 2;
    const uint32_t b = blockIdx.x * groups + g;
    if (g >= groups || b >= nblocks) return;
    const fourmc_block blk = blocks[b];
    uint64_t off; uint32_t len;
    if (mode == FOURMC_HASH_DST_RESULT) { off = blk.dst_off; len = blk.result > 0 ? uint32_t(blk.result) : 0u; }
    else                                { off = blk.src_off; len = blk.src_len; }
    const uint32_t h = xxh32_group(base + off, len, seed, lane);
    if ((lane & 3) == 0) {
        if (mode == FOURMC_VERIFY_SRC) blocks[b].result = (h == blk.xxh32) ? 0 : FOURMC_BLK_BADSUM;
        else blocks[b].xxh32 = h;
    }
}

// Staged variant for G <= 4 blocks per wavefront (the 8 GiB bench batch: G = 2).  The chain of an accumulator is
// add -> rotate -> multiply per stripe and nothing shortens it, but the product x * P2 of each input word is not on it, and
// neither is the load: here ALL 64 lanes fetch the words (64 / G consecutive words of a block per instruction, coalesced) and
// multiply them by P2 - one quarter-rate multiply per 64 words instead of one per 4 - and hand the products to the four
// accumulator lanes through LDS, transposed so that a lane reads its next four as one 16-byte word.  Per stripe the
// accumulator lanes then issue add, rotate, multiply and a quarter of an LDS read: about 30 clocks instead of 52.
template <int G>
__global__ __launch_bounds__(64)
void xxh32_staged_kernel(const uint8_t* __restrict__ base, fourmc_block* blocks, uint32_t nblocks, uint32_t seed, int mode)
{
    constexpr int W = 64 / G;                          // loader lanes = words per load instruction, per block
    constexpr int NI = 256 / W;                        // load instructions per bank of 64 stripes (256 words)
    __shared__ __attribute__((aligned(16))) uint32_t stage[G * 256];
    const int lane = threadIdx.x;
    const int ga = lane >> 2, chain = lane & 3;        // accumulator role: block ga of this wave (lanes 4 ga .. 4 ga + 3)
    const bool is_acc = ga < G;
    const uint32_t b = blockIdx.x * G + uint32_t(is_acc ? ga : 0);
    const bool have = is_acc && b < nblocks;
    fourmc_block blk = {};
    if (have) blk = blocks[b];
    uint64_t off = 0; uint32_t len = 0;
    if (have) {
        if (mode == FOURMC_HASH_DST_RESULT) { off = blk.dst_off; len = blk.result > 0 ? uint32_t(blk.result) : 0u; }
        else                                { off = blk.src_off; len = blk.src_len; }
    }
    const uint8_t* const p = base + off;
    const uint32_t nstripes = len >> 4;
    // loader role: lanes [gl W, gl W + W) fetch block gl's words
    const int gl = lane / W, j = lane % W;
    const uint64_t off_l = (uint64_t(uint32_t(__shfl(int(uint32_t(off >> 32)), 4 * gl))) << 32) | uint32_t(__shfl(int(uint32_t(off)), 4 * gl));
    const uint32_t nstripes_l = uint32_t(__shfl(int(nstripes), 4 * gl));
    const uint8_t* const pl = base + off_l + 4 * j;
    uint32_t acc = (chain == 0) ? seed + P1 + P2 : (chain == 1) ? seed + P2 : (chain == 2) ? seed : seed - P1;
    uint32_t s = 0;
    // Three register banks in rotation (the loop body is written out three times so that no loaded value is ever copied:
    // a copy would make the compiler wait for the youngest load): the words of bank k + 2 are requested while bank k is
    // consumed, 128 stripes (~5000 clocks) ahead - an HBM read takes about half of that.
    uint32_t x0[NI], x1[NI], x2[NI];
    auto fetch = [&](uint32_t (&x)[NI], uint32_t at) {
        // (branch-free: a lane whose block has no such bank reads the descriptor array instead and its words are never used;
        // a conditional load would sit in a basic block of its own and the compiler would wait for each before the next)
        const bool on = at + 64 <= nstripes_l;
        const uint8_t* const from = on ? pl + 16 * size_t(at) : reinterpret_cast<const uint8_t*>(blocks);
        const int stride = on ? 4 * W : 0;
#pragma unroll
        for (int i = 0; i < NI; i++) x[i] = ldw(from + i * stride);
    };
    auto consume = [&](const uint32_t (&x)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; i++) {                  // word w = i W + j of the bank: stripe w >> 2, accumulator w & 3
            const int w = i * W + j;
            stage[gl * 256 + (w & 3) * 64 + (w >> 2)] = x[i] * P2;
        }
        if (is_acc && s + 64 <= nstripes) {
            const uint4* const mine = reinterpret_cast<const uint4*>(stage + ga * 256 + chain * 64);
            uint4 v = mine[0], v1 = mine[1];
#pragma unroll
            for (int r4 = 0; r4 < 16; r4++) {
                const uint4 v2 = mine[r4 < 14 ? r4 + 2 : 15];      // two reads ahead of the four stripes being folded in
                acc = rotl(acc + v.x, 13) * P1; acc = rotl(acc + v.y, 13) * P1;
                acc = rotl(acc + v.z, 13) * P1; acc = rotl(acc + v.w, 13) * P1;
                v = v1; v1 = v2;
            }
        }
        s += 64;
    };
    fetch(x0, 0); fetch(x1, 64);
    for (;;) {
        if (!__ballot(s + 64 <= nstripes_l)) break;
        fetch(x2, s + 128); consume(x0);
        if (!__ballot(s + 64 <= nstripes_l)) break;
        fetch(x0, s + 128); consume(x1);
        if (!__ballot(s + 64 <= nstripes_l)) break;
        fetch(x1, s + 128); consume(x2);
    }
    if (!have) return;
    const uint32_t h = xxh32_finish(p, len, seed, acc, nstripes & ~63u, lane);
    if (chain == 0) {
        if (mode == FOURMC_VERIFY_SRC) blocks[b].result = (h == blk.xxh32) ? 0 : FOURMC_BLK_BADSUM;
        else blocks[b].xxh32 = h;
    }
}

} // namespace

extern "C" hipError_t fourmc_launch_xxh32(const void* d_base, fourmc_block* d_blocks, uint32_t n,
                                          uint32_t seed, int mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint32_t groups = 1;                               // blocks per wavefront: keep about one wave per SIMD (256 CUs x 4)
    while (groups < 16 && n / groups > 1024) groups *= 2;
    const uint8_t* const b8 = static_cast<const uint8_t*>(d_base);
    const dim3 grid((n + groups - 1) / groups);
    if (groups == 1) hipLaunchKernelGGL(xxh32_staged_kernel<1>, grid, dim3(64), 0, stream, b8, d_blocks, n, seed, mode);
    else if (groups == 2) hipLaunchKernelGGL(xxh32_staged_kernel<2>, grid, dim3(64), 0, stream, b8, d_blocks, n, seed, mode);
    else if (groups == 4) hipLaunchKernelGGL(xxh32_staged_kernel<4>, grid, dim3(64), 0, stream, b8, d_blocks, n, seed, mode);
    else hipLaunchKernelGGL(xxh32_kernel, grid, dim3(64), 0, stream, b8, d_blocks, n, seed, mode, groups);
    return hipGetLastError();
}
