// 4mc_amd/csrc/devcopy.h - wave-wide byte movers shared by the LZ4 and ZSTD decode kernels.
// (LZ77 sequence execution is the same operation in both formats: native/lz4/lz4.c:2300-2325,
//  native/zstd/decompress/zstd_decompress_block.c:755-1050.)
#ifndef FOURMC_DEVCOPY_H
#define FOURMC_DEVCOPY_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 ld16u(const uint8_t* p) {          // 16 B load, any alignment
    const U16B t = *reinterpret_cast<const U16B*>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}

// dst[0..n) = src[0..n), non-overlapping, any alignment: byte head up to a 16 B boundary of dst,
// then 16 B per lane (unaligned loads are legal on gfx950, stores are aligned), byte tail.
__device__ __forceinline__ void wave_copy(uint8_t* dst, const uint8_t* src, int n, int lane)
{
    const int head = min(n, int((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15));
    if (lane < head) dst[lane] = src[lane];
    int k = head;
    for (; k + 4096 <= n; k += 4096) {              // 4 x 1 KiB in flight
        const uint4 v0 = ld16u(src + k + 16 * lane), v1 = ld16u(src + k + 1024 + 16 * lane);
        const uint4 v2 = ld16u(src + k + 2048 + 16 * lane), v3 = ld16u(src + k + 3072 + 16 * lane);
        *reinterpret_cast<uint4*>(dst + k + 16 * lane) = v0;
        *reinterpret_cast<uint4*>(dst + k + 1024 + 16 * lane) = v1;
        *reinterpret_cast<uint4*>(dst + k + 2048 + 16 * lane) = v2;
        *reinterpret_cast<uint4*>(dst + k + 3072 + 16 * lane) = v3;
    }
    for (; k + 1024 <= n; k += 1024) {
        const uint4 v = ld16u(src + k + 16 * lane);
        *reinterpret_cast<uint4*>(dst + k + 16 * lane) = v;
    }
    for (; k < n; k += 64) { const int i = k + lane; if (i < n) dst[i] = src[i]; }
}

// dst[op .. op+n) = dst[op-off ..] with LZ4 (byte-serial forward) semantics; 1 <= off <= op.
__device__ __forceinline__ void copy_match(uint8_t* dst, int op, int off, int n, int lane)
{
    if (off >= 64 || off >= n) {
        // no step reads a byte written by the same step
        const uint8_t* from = dst + op - off;
        for (int k = 0; k < n; k += 64) {
            const int i = k + lane;
            if (i < n) { const uint8_t v = from[i]; dst[op + i] = v; }
        }
        return;
    }
    // overlapping: output is periodic with period `off`; all of it derives from [op-off, op).
    // step 1: first P bytes, P = off << s the smallest such multiple >= 32 (P < 64)
    int P = off; while (P < 32) P <<= 1;
    {
        int r = lane;                                   // r = lane mod off, by binary reduction
        for (int t = P; t >= off; t >>= 1) if (r >= t) r -= t;
        const int n1 = min(n, P);
        if (lane < n1) { const uint8_t v = dst[op - off + r]; dst[op + lane] = v; }
    }
    // step 2: bytes [P, 2P) copy from P behind; afterwards distance 2P >= 64 gives full steps
    int k = P;
    if (k < n) {
        const int n2 = min(n - k, P);
        if (lane < n2) { const uint8_t v = dst[op + k + lane - P]; dst[op + k + lane] = v; }
        k += n2;
    }
    const int D = 2 * P;
    for (; k < n; k += 64) {
        const int i = k + lane;
        if (i < n) { const uint8_t v = dst[op + i - D]; dst[op + i] = v; }
    }
}


} // namespace
#endif
