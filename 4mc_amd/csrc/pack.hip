// 4mc_amd/csrc/pack.hip — lays encoded blocks down as a contiguous .4mc/.4mz file image in HBM.
//
// The reference writes "12-byte block header + payload" per block straight after the codec call
// (native/4mc.c:309-315 / :321-327).  On the device the payload sizes are only known after the
// batch has been encoded, so blocks are encoded into fixed 4 MiB staging slots and then packed:
// block b goes to image offset off[b] = 12 + sum_{j<b}(12 + csize_j) (native/4mc.c:293 — the
// same numbers the footer index stores).  One wavefront per block; the 12 header bytes are
// big-endian u32 (usize, csize, xxh32); the payload copy is 16 B/lane with aligned stores.
// Traffic: csize read + (12 + csize) written per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"

namespace {

struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };

__global__ __launch_bounds__(256)
void pack_image_kernel(const uint8_t* __restrict__ staging, uint8_t* __restrict__ image,
                       const fourmc_block* __restrict__ blocks, const uint64_t* __restrict__ image_off,
                       uint32_t nblocks)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = blocks[b];
    const uint32_t csize = blk.result > 0 ? uint32_t(blk.result) : 0u;
    uint8_t* out = image + image_off[b];
    const uint8_t* in = staging + blk.dst_off;
    const uint32_t t = threadIdx.x;
    if (t < 12) {
        const uint32_t field = t < 4 ? blk.src_len : (t < 8 ? csize : blk.xxh32);
        out[t] = uint8_t(field >> (8 * (3 - (t & 3))));
    }
    out += 12;
    // head bytes up to a 16 B boundary of the destination, 16 B body, byte tail
    const uint32_t head = min(csize, uint32_t((16 - (reinterpret_cast<uintptr_t>(out) & 15)) & 15));
    if (t < head) out[t] = in[t];
    const uint32_t body = (csize - head) & ~15u;
    for (uint32_t k = 16 * t; k < body; k += 16 * 256) {
        const U16B v = *reinterpret_cast<const U16B*>(in + head + k);
        *reinterpret_cast<uint4*>(out + head + k) = make_uint4(v.x, v.y, v.z, v.w);
    }
    const uint32_t tail0 = head + body;
    if (tail0 + t < csize) out[tail0 + t] = in[tail0 + t];
}

} // namespace

extern "C" hipError_t fourmc_launch_pack_image(const void* d_staging, void* d_image, const fourmc_block* d_blocks,
                                               const uint64_t* d_image_off, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_image_kernel, dim3(n), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(d_staging), static_cast<uint8_t*>(d_image), d_blocks, d_image_off, n);
    return hipGetLastError();
}
