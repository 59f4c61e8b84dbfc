// 4mc_amd/csrc/xxh32.hip — K5: batched XXH32 (seeded, 32-bit) over block payloads on gfx950.
//
// Replaces the per-block XXH32(payload, n, 0) calls of the reference container loops
// (native/4mc.c:311,323 on encode, :637,645 on decode; JNI xxhash32 native/jniCompressor.c:183)
// -> native/lz4/xxhash.c:392-415 (stripe loop :352-389, tail :291-345).
//
// XXH32 is four independent multiply-rotate chains over 16-byte stripes; a chain step depends on
// the previous one and the round function is not associative, so there is no intra-block
// parallel prefix.  Parallelism is ACROSS blocks: one wavefront per block, lanes 0..3 carry the
// four accumulators; the other lanes only stream.  The payload is read once with coalesced
// 16 B/lane loads (one 1 KiB granule in flight ahead of use) into a 4 KiB LDS ring, from which
// the accumulator lanes pick their 4-byte words (payloads start at arbitrary byte offsets inside a
// .4mc file, so words are re-aligned with v_alignbyte).  Bound: HBM read, `len` bytes per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"

namespace {

constexpr uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
constexpr int kRing = 4096, kChunk = 1024;

__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ uint32_t xxh32_block(const uint8_t* p, uint32_t len, uint32_t seed, uint8_t* ring, int lane)
{
    const uint32_t delta = uint32_t(reinterpret_cast<uintptr_t>(p) & 15);
    const uint8_t* abase = p - delta;          // pointer arithmetic keeps the global address space (no flat_load)
    const uint32_t qend = delta + len;                 // end in aligned coordinates
    const uint32_t nstripes = len >> 4;
    const uint32_t sh = delta & 3;

    auto fetch = [&](uint32_t q) {
        const uint32_t g = q + 16u * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < qend) v = *reinterpret_cast<const uint4*>(abase + g);
        return v;
    };
    uint32_t fill_hi = 0;                              // ring holds aligned positions [.., fill_hi)
    uint4 pend = fetch(0);
    uint32_t acc = (lane == 0) ? seed + P1 + P2 : (lane == 1) ? seed + P2 : (lane == 2) ? seed : seed - P1;

    uint32_t s = 0;                                    // next stripe
    while (s < nstripes) {
        // stage one more granule, then consume every stripe that is now complete in the ring
        *reinterpret_cast<uint4*>(ring + ((fill_hi + 16u * lane) & (kRing - 1))) = pend;
        fill_hi += kChunk;
        pend = fetch(fill_hi);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        uint32_t s_hi = (fill_hi >= delta + 16) ? (fill_hi - delta) >> 4 : 0;   // stripes fully staged
        if (s_hi > nstripes) s_hi = nstripes;
        if (lane < 4) {
            uint32_t q = delta + 16u * s + 4u * lane;  // aligned-space byte position of my word
            const uint32_t* r32 = reinterpret_cast<const uint32_t*>(ring);
            for (; s + 4 <= s_hi; s += 4, q += 64) {
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t w = (q + 16u * u) >> 2;
                    lo[u] = r32[w & (kRing / 4 - 1)];
                    hi[u] = r32[(w + 1) & (kRing / 4 - 1)];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t x = (sh == 0) ? lo[u] : __builtin_amdgcn_alignbyte(hi[u], lo[u], sh);
                    acc = rotl(acc + x * P2, 13) * P1;
                }
            }
            for (; s < s_hi; s++, q += 16) {
                const uint32_t w = q >> 2;
                const uint32_t lo = r32[w & (kRing / 4 - 1)], hi = r32[(w + 1) & (kRing / 4 - 1)];
                const uint32_t x = (sh == 0) ? lo : __builtin_amdgcn_alignbyte(hi, lo, sh);
                acc = rotl(acc + x * P2, 13) * P1;
            }
        }
        s = s_hi;
    }
    uint32_t h;
    if (len >= 16) {
        const uint32_t a0 = __builtin_amdgcn_readlane(acc, 0), a1 = __builtin_amdgcn_readlane(acc, 1);
        const uint32_t a2 = __builtin_amdgcn_readlane(acc, 2), a3 = __builtin_amdgcn_readlane(acc, 3);
        h = rotl(a0, 1) + rotl(a1, 7) + rotl(a2, 12) + rotl(a3, 18);
    } else {
        h = seed + P5;
    }
    h += len;
    // tail (< 16 bytes): lane j reads byte j of the tail straight from memory
    const uint32_t tail = len & 15;
    const uint32_t tb = (uint32_t(lane) < tail) ? p[(len & ~15u) + lane] : 0u;
    uint32_t i = 0;
    for (; i + 4 <= tail; i += 4) {
        const uint32_t w = uint32_t(__builtin_amdgcn_readlane(tb, i)) | (uint32_t(__builtin_amdgcn_readlane(tb, i + 1)) << 8) |
                           (uint32_t(__builtin_amdgcn_readlane(tb, i + 2)) << 16) | (uint32_t(__builtin_amdgcn_readlane(tb, i + 3)) << 24);
        h = rotl(h + w * P3, 17) * P4;
    }
    for (; i < tail; i++) h = rotl(h + uint32_t(__builtin_amdgcn_readlane(tb, i)) * P5, 11) * P1;
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(64)
void xxh32_kernel(const uint8_t* __restrict__ base, fourmc_block* blocks, uint32_t nblocks,
                  uint32_t seed, int mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = blocks[b];
    uint64_t off; uint32_t len;
    if (mode == FOURMC_HASH_DST_RESULT) { off = blk.dst_off; len = blk.result > 0 ? uint32_t(blk.result) : 0u; }
    else                                { off = blk.src_off; len = blk.src_len; }
    const uint32_t h = xxh32_block(base + off, len, seed, ring, threadIdx.x);
    if (threadIdx.x == 0) {
        if (mode == FOURMC_VERIFY_SRC) blocks[b].result = (h == blk.xxh32) ? 0 : FOURMC_BLK_BADSUM;
        else blocks[b].xxh32 = h;
    }
}

} // namespace

extern "C" hipError_t fourmc_launch_xxh32(const void* d_base, fourmc_block* d_blocks, uint32_t n,
                                          uint32_t seed, int mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(xxh32_kernel, dim3(n), dim3(64), 0, stream,
                       static_cast<const uint8_t*>(d_base), d_blocks, n, seed, mode);
    return hipGetLastError();
}
