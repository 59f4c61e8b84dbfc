/*
 * include/fourmc_gpu.h — C ABI of the MI355X block engine behind 4mc's codec call sites.
 *
 * This is the drop-in boundary for the hot path of fingltd/4mc (SURVEY.md §8(a),(b)): every place
 * where the reference calls ONE codec function on ONE <=4 MiB block
 *     native/4mc.c:301,311,323  (compress loop),  :637,645,661 (decode loop), 4mz twins :467,:810
 *     native/jniCompressor.c:91,124,157   native/jniDecompressor.c:88
 *     native/jniZstdCompressor.c:93,126,159   native/jniZstdDecompressor.c:90
 * is served here, with independent blocks batched into single HIP launches on gfx950.
 *
 * Plain pointers and sizes only; no C++ or torch types.  `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  All `d_` pointers are device (HBM) pointers.
 * Every function returns FOURMC_OK or a negative FOURMC_E* code; nothing here ever falls back to
 * a CPU codec: without a usable gfx950 device the calls fail with FOURMC_ENODEV.
 */
#ifndef FOURMC_GPU_H
#define FOURMC_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOURMC_BLOCKSIZE   (4u * 1024u * 1024u)   /* native/4mc.c:116                          */
#define FOURMC_MAGIC_4MC   0x344D4300u            /* native/4mc.c:111                          */
#define FOURMC_MAGIC_4MZ   0x344D5A00u            /* native/4mc.c:112                          */

enum {
    FOURMC_OK       =  0,
    FOURMC_ENODEV   = -1,   /* no HIP device / wrong architecture                              */
    FOURMC_EHIP     = -2,   /* a HIP runtime call failed (fourmc_gpu_last_error() has the text) */
    FOURMC_EINVAL   = -3,
    FOURMC_ENOMEM   = -4,
    FOURMC_EUNSUP   = -5    /* codec/level not implemented on the device yet                   */
};

/* Codec selectors: the (function, level) pairs 4mc can reach (native/4mc.c:243-253,:411-419). */
enum {
    FOURMC_CODEC_LZ4_FAST = 0,   /* LZ4_compress_default            4mc -1  (Lz4Compressor)    */
    FOURMC_CODEC_LZ4_MC   = 1,   /* LZ4_compressMC                  4mc -2                     */
    FOURMC_CODEC_LZ4_HC   = 2,   /* LZ4_compress_HC(level 4 / 8)    4mc -3 / -4                */
    FOURMC_CODEC_ZSTD     = 3    /* ZSTD_compress(level 1..12)      4mz -1..-4 = 1/3/6/12      */
};

/* One independent block.  Offsets are relative to the base pointers given to the batch call, so
 * one descriptor array describes a whole file image resident in HBM.  32 bytes, no padding. */
typedef struct fourmc_block {
    uint64_t src_off;   /* in : byte offset of the block's input                                */
    uint64_t dst_off;   /* in : byte offset of the block's output                               */
    uint32_t src_len;   /* in : input bytes                                                     */
    uint32_t dst_cap;   /* in : output capacity in bytes                                        */
    int32_t  result;    /* out: codec return value (reference convention, see each call)        */
    uint32_t xxh32;     /* out (encode/hash) or in (4mc decode: expected checksum)              */
} fourmc_block;

/* ---- device management ------------------------------------------------------------------- */
int         fourmc_gpu_device_count(void);           /* >=0, or FOURMC_ENODEV                    */
int         fourmc_gpu_init(int device);             /* select device, check gfx950              */
const char* fourmc_gpu_last_error(void);
const char* fourmc_gpu_arch(void);                   /* "gfx950:..." of the selected device      */

/* ---- raw block codecs, device resident (one launch for `n` blocks) ------------------------ */
/* result = LZ4_decompress_safe(src, dst, src_len, dst_cap)        native/lz4/lz4.c:2345        */
int fourmc_gpu_lz4_decompress(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                              uint32_t n, void* stream);
/* result = LZ4_compress_default(src, dst, src_len, dst_cap)       native/lz4/lz4.c:1435
 * (0 = does not fit dst_cap).  Payload bytes identical to the reference 64-bit LE build.      */
int fourmc_gpu_lz4_compress_fast(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                 uint32_t n, void* stream);
/* result = LZ4_compress_HC(src, dst, src_len, dst_cap, level), hash-chain levels 1..8 (4mc High = 4,
 * Ultra = 8); byte-identical payloads                            native/lz4/lz4hc.c:958-973       */
int fourmc_gpu_lz4_compress_hc(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                               uint32_t n, int level, void* stream);
/* result = LZ4_compressMC_limitedOutput(src, dst, src_len, dst_cap), or LZ4_compressMC (no limit)
 * when dst_cap == 0xFFFFFFFF; byte-identical payloads           native/lz4/lz4mc.c:582-606         */
int fourmc_gpu_lz4_compress_mc(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                               uint32_t n, void* stream);
/* LZ4 fast encoder of fourmc_gpu_lz4_compress_fast and of the LZ4-fast container encode: 0 (default) the reference parse, payload
 * bytes identical to the reference build; 1 the ratio-tolerance encoder (lz4_par_encode.hip): every payload is one valid LZ4 block
 * that LZ4_decompress_safe decodes to the input, but NOT the reference's bytes - sizes within 3 % of the reference parse on the
 * S-mix (north_star: "otherwise compression ratio is reported within a stated tolerance").  Blocks up to 4 MiB are compressed as
 * 64 KiB segments in parallel; what lies beyond 4 MiB of a block goes out as literals.  env FOURMC_LZ4_ENCODE = exact | parallel. */
void fourmc_gpu_set_lz4_encode_mode(int mode);
int  fourmc_gpu_get_lz4_encode_mode(void);
/* Tuning knob (not part of the reference boundary): which LZ4 decode path serves the launches - 6 auto (default: the tile path
 * up to 1536 blocks per launch, the segment-parallel path above), 2 the exact walker alone (what an automatic choice ends at when
 * no workspace can be had), 11 the segment-parallel path, 13 the tile path; results are identical (env FOURMC_DECODE = auto | exact |
 * seg | tile).  Any other value selects auto: the designs measured and not kept are sources under tools/research/. */
void fourmc_gpu_set_lz4_decode_path(int path);
int  fourmc_gpu_get_lz4_decode_path(void);
/* Tuning knob: 4mz decode as entropy kernel + execute kernel (1, default; FOURMC_ZDECODE=split) or all in the one-wave kernel
 * (0; FOURMC_ZDECODE=single); results are identical. */
void fourmc_gpu_set_zstd_decode_split(int on);
int  fourmc_gpu_get_zstd_decode_split(void);
/* Statistics (read-only, not part of the reference boundary): one-block host calls made so far (the LZ4_* / ZSTD_* twins and the
 * JNI entry points: one call = one block) and the launches that served them - calls that arrive while a launch is in flight
 * share the next one (engine.hip: host_one), so launches < calls under concurrency. */
void fourmc_gpu_one_block_stats(unsigned long long* calls, unsigned long long* launches);
/* The engine keeps one device workspace per stream and reuses it (the segment-parallel LZ4 decode of 8192 blocks needs 92 GB, the zstd
 * level-12 encoder 48 MiB per block).  This frees them all, synchronizing each stream first; the next call allocates again. */
int fourmc_gpu_release_workspaces(void);
/* result = ZSTD_compress(dst + dst_off, dst_cap, src + src_off, src_len, level) as int: frame bytes, or
 * -(ZSTD error number), e.g. -70 = dstSize_tooSmall        native/zstd/compress/zstd_compress.c:4806
 * Levels 1 .. 12 are on the device, byte-identical for every input size: every strategy their rows of the level table name (clevels.h:25-130:
 * fast, dfast, greedy, lazy, lazy2 with the row-hash or hash-chain finder, and - for inputs of 256 KiB and less - btlazy2 / btopt).  4mz -1 .. -4
 * use 1, 3, 6, 12.  Levels 13 and above (btlazy2 on full blocks, btultra) and levels below 1 return FOURMC_EUNSUP - never different bytes. */
int fourmc_gpu_zstd_compress(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                             uint32_t n, int level, void* stream);

/* result = ZSTD_decompress(dst, dst_cap, src, src_len) as int: decoded bytes, or < 0 where the
 * reference returns an error code (ZSTD_isError)            native/zstd/decompress/zstd_decompress.c:1112 */
int fourmc_gpu_zstd_decompress(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                               uint32_t n, void* stream);
/* xxh32 = XXH32(src + src_off, src_len, seed)                     native/lz4/xxhash.c:392      */
int fourmc_gpu_xxh32(const void* d_src, fourmc_block* d_blocks, uint32_t n, uint32_t seed,
                     void* stream);

/* ---- container-level block ops (what one iteration of the 4mc.c loops does) --------------- */
/* Encode: codec with capacity src_len-1 (native/4mc.c:301); result<=0 => the block is stored
 * raw (:318-329): payload = input, result = src_len.  xxh32 = XXH32(stored payload) (:311,:323).
 * dst_cap must be >= src_len.  `codec`/`level` as FOURMC_CODEC_*.                               */
int fourmc_gpu_4mc_encode_blocks(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                 uint32_t n, int codec, int level, void* stream);
/* Decode: verify XXH32(payload)==xxh32 (native/4mc.c:637,645); src_len==dst_cap => stored copy
 * (:635-642) else codec decode (:661).  result = decoded bytes, or
 * FOURMC_BLK_BADSUM / FOURMC_BLK_CORRUPT (the two exit-4 conditions of the reference CLI).      */
#define FOURMC_BLK_BADSUM   (-1000000001)
#define FOURMC_BLK_CORRUPT  (-1000000002)
int fourmc_gpu_4mc_decode_blocks(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                 uint32_t n, int codec, void* stream);

/* Pack: after fourmc_gpu_4mc_encode_blocks, copy each block's 12-byte big-endian header
 * (usize = src_len, csize = result, xxh32; native/4mc.c:309-312) and its payload from the staging
 * slot (d_staging + dst_off) to d_image + d_image_off[b].  d_image_off[b] is the absolute file
 * offset of block b's header = 12 + sum_{j<b}(12 + csize_j) (native/4mc.c:293), i.e. exactly the
 * footer-index entries; the caller (host or RCCL-gathered prefix sum) supplies them.            */
int fourmc_gpu_4mc_pack_image(const void* d_staging, void* d_image, const fourmc_block* d_blocks,
                              const uint64_t* d_image_off, uint32_t n, void* stream);

/* ---- host-buffer conveniences with the reference's per-block signatures ------------------- */
/* These stage one block through HBM (H2D, one launch, D2H).  They exist so the JNI entry points
 * keep their exact one-call-one-block contract (SURVEY.md §8(b) "Batching constraint").        */
int      fourmc_LZ4_compressBound(int inputSize);                       /* lz4.h:212            */
int      fourmc_LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity);
int      fourmc_LZ4_compressMC(const char* src, char* dst, int srcSize);
int      fourmc_LZ4_compressMC_limitedOutput(const char* src, char* dst, int srcSize, int maxOutputSize);
int      fourmc_LZ4_compress_HC(const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel);
int      fourmc_LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);
size_t   fourmc_ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);
size_t   fourmc_ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);   /* zstd.h:156 */
size_t   fourmc_ZSTD_compressBound(size_t srcSize);                                                             /* zstd.h:206 */
unsigned fourmc_XXH32(const void* input, size_t len, unsigned seed);    /* host scalar: framing bytes, JNI xxhash32 */

/* Host batch: `n` blocks described by host-side descriptors over host buffers; the engine does
 * one H2D of the inputs, one launch, one D2H of outputs + descriptors.  Used by the file API.  */
/* Page-locked host memory for staging buffers handed to the two calls below (NULL if it cannot be had: use malloc then). */
void* fourmc_host_alloc(size_t bytes);
void  fourmc_host_free(void* p);
int fourmc_host_4mc_encode(const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                           fourmc_block* blocks, uint32_t n, int codec, int level);
int fourmc_host_4mc_decode(const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                           fourmc_block* blocks, uint32_t n, int codec);
/* Encode `n` blocks and return the finished piece of the file image ("12-byte block header + payload" per block, back to
 * back: native/4mc.c:309-315, :321-327) in ONE device-to-host transfer into `image` - which may be the output file itself
 * (a shared mapping).  image_off[b]: where block b's header sits in the piece; *image_bytes: the piece's length. */
int fourmc_host_4mc_encode_image(const void* src, size_t src_bytes, fourmc_block* blocks, uint32_t n, int codec, int level,
                                 void* image, size_t image_cap, uint64_t* image_off, size_t* image_bytes);

/* ---- debug / profiling exports: the research side build only (make -C 4mc_amd/csrc research -> libhadoop-4mc-research.so,
 * -DFOURMC_RESEARCH); the drop-in library does not export them ------------------------------------------------------------ */
#ifdef FOURMC_RESEARCH
/* Profiling aid (not part of the reference boundary): copy a range of the engine's per-block device
 * workspace to the host; the zstd kernels leave per-phase cycle counters there (tools/zstd_timing.py). */
int fourmc_gpu_debug_read_workspace(void* host, size_t offset, size_t bytes);
/* Test aid: blocks the 4mz execute kernel completed / handed back to the one-wave kernel since the last call. */
int  fourmc_gpu_debug_zstd_exec_counts(unsigned long long* executed, unsigned long long* handed_back);
/* Test aid: runs only the parser kernel of the block-parallel LZ4 decoder on `n` blocks and copies the first `bytes` of
 * the workspace (block 0's slot first: header, window descriptors, token positions) to `host`; layout[0..2] = slot bytes,
 * descriptor offset, token offset. */
int fourmc_gpu_debug_lz4_parse(const void* d_src, const void* d_dst, fourmc_block* d_blocks, uint32_t n, int container_mode,
                               void* host, size_t bytes, size_t* layout);
/* one-block host calls (LZ4_* / ZSTD_* twins, JNI) made so far, and the launches that served them (concurrent calls share one) */
void fourmc_debug_one_block_counters(unsigned long long* calls, unsigned long long* launches);

#endif

#ifdef __cplusplus
}
#endif
#endif
