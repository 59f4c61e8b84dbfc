// 4mc_amd/csrc/kernels.h — launchers of the gfx950 kernels (internal to the engine).
#ifndef FOURMC_KERNELS_H
#define FOURMC_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif
// The launchers are the engine's own: C linkage for the objects of this library, but not part of its dynamic symbol table
// (the boundary is include/fourmc_gpu.h, include/fourmc.h and the JNI names).
#pragma GCC visibility push(hidden)

// what a hash launch covers for each block descriptor
enum {
    FOURMC_HASH_SRC        = 0,   // XXH32(src + src_off, src_len)          -> blocks[b].xxh32
    FOURMC_HASH_DST_RESULT = 1,   // XXH32(dst + dst_off, max(result,0))    -> blocks[b].xxh32
    FOURMC_VERIFY_SRC      = 2    // compare XXH32(src..) with blocks[b].xxh32; mismatch -> result = BADSUM,
                                  // match -> result = 0
};

/* what an LZ4 decode launch of n blocks runs: resolved once per call by fourmc_lz4_decode_plan, handed to the launcher */
typedef struct fourmc_lz4_plan {
    int      path;          /* decode path after "auto" has been resolved (lz4_decode.hip) */
    uint32_t batch;         /* blocks per piece of the launch (paths with a workspace) */
    size_t   work_bytes;    /* device workspace to lease: 0 for the paths that need none */
    int      ok;            /* 0: the pieces cannot get smaller (the workspace cannot be had) */
} fourmc_lz4_plan;
fourmc_lz4_plan fourmc_lz4_decode_plan(uint32_t n, uint32_t shrink);
size_t     fourmc_lz4_parse_work_bytes(uint32_t n);
size_t     fourmc_lz4_decode_tok_offset(void);
hipError_t fourmc_launch_lz4_decode(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                    uint32_t n, int container_mode, const fourmc_lz4_plan* plan, void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_rows(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                  uint32_t n, int container_mode, hipStream_t stream);
hipError_t fourmc_launch_lz4_lanes(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                  uint32_t n, int container_mode, hipStream_t stream);
hipError_t fourmc_launch_lz4_wx(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                  uint32_t n, int container_mode, hipStream_t stream, const uint32_t* pick, uint32_t want);
size_t     fourmc_lz4_seg_work_bytes(uint32_t n);
uint32_t   fourmc_lz4_seg_batch(void);              /* blocks per launch of the segment-parallel path (bounds its workspace) */
size_t     fourmc_lz4_tile_work_bytes(uint32_t n);
uint32_t   fourmc_lz4_tile_batch(void);             /* blocks per launch of the tile path */
hipError_t fourmc_launch_lz4_tile(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                  int container_mode, void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_seg(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                 int container_mode, void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_seg_walk(const void* d_src, fourmc_block* d_blocks, uint32_t n, int container_mode,
                                      void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_ring(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                  int container_mode, void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_parse(const void* d_src, const void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                   int container_mode, void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_exec(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                  const void* d_work, hipStream_t stream);
hipError_t fourmc_launch_lz4_encode_fast(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                         uint32_t n, int container_mode, hipStream_t stream);
size_t     fourmc_lz4_par_work_bytes(uint32_t n);
hipError_t fourmc_launch_lz4_encode_par(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                        int container_mode, void* d_work, hipStream_t stream);
hipError_t fourmc_launch_pack_image(const void* d_staging, void* d_image, const fourmc_block* d_blocks,
                                    const uint64_t* d_image_off, uint32_t n, hipStream_t stream);
size_t     fourmc_lz4hc_work_bytes(uint32_t n);
hipError_t fourmc_launch_lz4hc_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                      void* d_work, int level, int container_mode, hipStream_t stream);
hipError_t fourmc_launch_lz4mc_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                      void* d_work, int container_mode, hipStream_t stream);
#ifdef FOURMC_RESEARCH      /* the research side build exports these two: tools/zstd_timing.py and tools/k7x_prof.py size their read-backs with them */
#pragma GCC visibility push(default)
#endif
size_t     fourmc_zstd_scratch_bytes(uint32_t n);
size_t     fourmc_zstd_dec_counter_offset(void);
#ifdef FOURMC_RESEARCH
#pragma GCC visibility pop
#endif
hipError_t fourmc_launch_zstd_decode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                     void* d_scratch, int container_mode, hipStream_t stream);
size_t     fourmc_zstd_enc_work_bytes(uint32_t n, int level);
int        fourmc_zstd_enc_level_ok(int level);                 /* 1: the device has every strategy the level's rows name (levels 1..12) */
hipError_t fourmc_launch_zstd_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                     void* d_work, int container_mode, int level, int serial, hipStream_t stream);
hipError_t fourmc_launch_xxh32(const void* d_base, fourmc_block* d_blocks, uint32_t n,
                               uint32_t seed, int mode, hipStream_t stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif

#ifdef __HIPCC__
// A block descriptor arrives through a vector load.  Everything a wave-per-block kernel derives from it (lengths, limits, loop
// counters, pointers) is wave-uniform: pin the fields to scalar registers, or those values live in vector registers and the
// kernel's scalar control flow is computed on the vector unit.
__device__ __forceinline__ fourmc_block uniform_block(const fourmc_block& v)
{
    auto u32 = [](uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); };
    auto u64 = [&](uint64_t x) { return (uint64_t(u32(uint32_t(x >> 32))) << 32) | u32(uint32_t(x)); };
    fourmc_block r;
    r.src_off = u64(v.src_off); r.dst_off = u64(v.dst_off); r.src_len = u32(v.src_len); r.dst_cap = u32(v.dst_cap);
    r.result = int32_t(u32(uint32_t(v.result))); r.xxh32 = u32(v.xxh32);
    return r;
}
#endif
#endif
