// 4mc_amd/csrc/devenc.h - wave-wide emit helpers shared by the LZ4 encoders (fast and HC):
// unaligned loads, literal copy, length continuation bytes (native/lz4/lz4.c:1083-1107,1184-1196;
// native/lz4/lz4hc.c:467-548 write the same bytes).
#ifndef FOURMC_DEVENC_H
#define FOURMC_DEVENC_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

struct __attribute__((packed, aligned(1))) U8B  { uint64_t v; };
struct __attribute__((packed, aligned(1))) U4B  { uint32_t v; };
struct __attribute__((packed, aligned(1))) U16B { uint64_t a, b; };
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) { return reinterpret_cast<const U8B*>(p)->v; }
__device__ __forceinline__ uint32_t ld4(const uint8_t* p) { return reinterpret_cast<const U4B*>(p)->v; }

// wave-wide byte copy, non-overlapping
__device__ __forceinline__ void copy_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, int lane)
{
    if (n <= 64) { if (uint32_t(lane) < n) dst[lane] = src[lane]; return; }
    const uint32_t head = min(n, uint32_t((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15));
    if (uint32_t(lane) < head) dst[lane] = src[lane];
    uint32_t k = head;
    for (; k + 1024 <= n; k += 1024) {
        const U16B v = *reinterpret_cast<const U16B*>(src + k + 16 * lane);
        *reinterpret_cast<uint4*>(dst + k + 16 * lane) =
            make_uint4(uint32_t(v.a), uint32_t(v.a >> 32), uint32_t(v.b), uint32_t(v.b >> 32));
    }
    for (; k < n; k += 64) { const uint32_t i = k + lane; if (i < n) dst[i] = src[i]; }
}

// emits the length continuation bytes for value `rest` (>= 0): rest/255 bytes of 255, then rest%255
__device__ __forceinline__ uint32_t emit_len(uint8_t* op, uint32_t rest, int lane)
{
    const uint32_t n255 = rest / 255;
    for (uint32_t k = 0; k < n255; k += 64) if (k + lane < n255) op[k + lane] = 255;
    if (lane == 0) op[n255] = uint8_t(rest - n255 * 255);
    return n255 + 1;
}


} // namespace
#endif
