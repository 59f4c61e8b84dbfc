/*
 * include/fourmc.h — file-level API and container framing of the MI355X 4mc build.
 *
 * The four file functions keep the exact signatures and behaviour of the reference library API
 * (native/4mc.h:36-41): messages on stderr gated by displayLevel, errors leave through exit()
 * with the reference's codes (1 generic / 2 input / 3 output / 4 content, native/4mc.c:135-161),
 * return value 0.  Internally they batch independent blocks into single HIP launches through
 * include/fourmc_gpu.h.
 *
 * The fourmc_frame_* / fourmc_index_* functions are the byte-exact framing (header, block
 * header, end mark, footer index) factored out so that it can be tested without a GPU and
 * reused by bindings; they restate native/4mc.c:264-274,:309-312,:336-362 (writer),
 * :575-585,:670-688 (reader) and the footer consumer
 * java/hadoop-4mc/src/main/java/com/fing/compression/fourmc/FourMcBlockIndex.java:92-173.
 */
#ifndef FOURMC_H
#define FOURMC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference library API (native/4mc.h:36-41) ------------------------------------------ */
int fourMCcompressFilename  (int displayLevel, int overwrite, char* input_filename, char* output_filename, int compressionlevel);
int fourMcDecompressFileName(int displayLevel, int overwrite, char* input_filename, char* output_filename);
int fourMZcompressFilename  (int displayLevel, int overwrite, char* input_filename, char* output_filename, int compressionlevel);
int fourMZDecompressFileName(int displayLevel, int overwrite, char* input_filename, char* output_filename);

/* stdin / stdout / null markers (native/4mc.h:45-52) */
#define FOURMC_NULL_OUTPUT "null"
#define FOURMC_STDINMARK   "stdin"
#define FOURMC_STDOUTMARK  "stdout"
#define FOURMC_NULMARK     "/dev/null"

/* ---- framing (all fields big-endian u32; 4mc-format-spec, 4mz-format-spec) ---------------- */
#define FOURMC_HEADERSIZE 12u
#define FOURMC_FOOTERSIZE(nblocks) (20u + 4u * (nblocks))          /* native/4mc.c:117        */

unsigned fourmc_XXH32(const void* input, size_t len, unsigned seed); /* host scalar, for framing
                                                                       bytes and JNI xxhash32   */
void     fourmc_frame_header(uint8_t out[12], uint32_t magic);      /* native/4mc.c:264-268    */
/* 0 ok; 1 wrong magic; 2 wrong version; 3 wrong checksum (native/4mc.c:575-585,:870-873) */
int      fourmc_frame_check_header(const uint8_t in[12], uint32_t magic);
void     fourmc_frame_block_header(uint8_t out[12], uint32_t usize, uint32_t csize, uint32_t xxh32);
void     fourmc_frame_parse_block_header(const uint8_t in[12], uint32_t* usize, uint32_t* csize, uint32_t* xxh32);
/* Footer from ABSOLUTE file offsets of each block header (first = 12); writes
 * FOURMC_FOOTERSIZE(n) bytes; native/4mc.c:344-358. */
size_t   fourmc_frame_footer(uint8_t* out, uint32_t magic, const uint64_t* block_offsets, uint32_t nblocks);
/* Parses/validates a footer image; fills absolute offsets (may be NULL).  Returns the number of
 * blocks, or -1 bad size, -2 bad checksum, -3 bad version, -4 bad magic/size echo. */
int64_t  fourmc_frame_parse_footer(const uint8_t* foot, size_t len, uint32_t magic, uint64_t* block_offsets);

/* ---- footer-index queries (FourMcBlockIndex.java:92-173) ---------------------------------- */
/* index of the first block whose offset is >= pos, or -1            (findNextPosition :104)    */
int64_t  fourmc_index_find_next(const uint64_t* offsets, uint32_t n, uint64_t pos);
/* index of the block containing pos, or -1                          (findBelongingBlockIndex)  */
int64_t  fourmc_index_find_block(const uint64_t* offsets, uint32_t n, uint64_t pos);
/* split alignment: start/end rounded forward to block starts        (alignSlice* :142-173)     */
uint64_t fourmc_index_align_start(const uint64_t* offsets, uint32_t n, uint64_t start, uint64_t end);
uint64_t fourmc_index_align_end(const uint64_t* offsets, uint32_t n, uint64_t end, uint64_t file_size);

/* ---- random access through the footer index (what a Hadoop split does: FourMcInputStream.java:163-239) ----
 * Library calls: errors are returned, never exit().  -1 I/O, -2 bad header/footer, -3 block range,
 * -4 corrupt block, -5 dst too small, -6 engine error (fourmc_gpu_last_error()). */
int64_t  fourmc_file_block_count(const char* path, int* is_zstd);           /* blocks in a .4mc/.4mz file   */
/* decodes blocks [first, first+count) into dst; returns the decoded byte count */
int64_t  fourmc_file_decode_blocks(const char* path, uint32_t first, uint32_t count, void* dst, size_t dst_cap);

/* ---- one file written by several ranks, one process per GPU (4mc_amd/csrc/shard.c) ----------------------------
 * Rank r of `world` owns the contiguous block range fourmc_shard_range() gives; the only exchange is one all-gather of
 * the per-block compressed sizes, after which every rank knows the footer index (fourmc_shard_offsets) and pwrite()s its
 * own byte range; rank 0 adds header, end mark and footer (native/4mc.c:264-268,:336-362).  The collective is the
 * caller's: recv_all receives `world` rows of `bytes` bytes in rank order; return 0 on success. */
typedef int (*fourmc_allgather_fn)(void* ctx, const void* send, size_t bytes, void* recv_all);
void fourmc_shard_range(uint64_t nblocks, int rank, int world, uint64_t* first, uint64_t* count);
void fourmc_shard_offsets(const uint32_t* csize_all, uint64_t nblocks, uint64_t* off_all);
int  fourmc_shard_write(int fd, uint32_t magic, int rank, uint64_t first, uint64_t count, uint64_t nblocks, const uint64_t* off_all,
                        const uint32_t* csize_all, const uint32_t* usize, const uint32_t* xxh32, const uint8_t* payloads, const uint64_t* payload_off);
/* 0 ok; -1 input, -2 output, -3 engine (fourmc_gpu_last_error()), -4 collective, -5 memory, -6 another rank failed.
 * The row a rank sends through `allgather` is {status, its blocks' compressed sizes}: a rank that failed still takes part in the
 * exchange, and every rank leaves with an error - before anything is written - when any status is not 0. */
int  fourmc_file_compress_sharded(const char* in_name, const char* out_name, int level, uint32_t magic, int rank, int world,
                                  fourmc_allgather_fn allgather, void* ctx);

/* The other direction, no exchange at all: rank r decodes its block range through the footer index and pwrite()s it at
 * block index * FOURMC_BLOCKSIZE; the owner of the last block sets the final size.  0 ok; -1 input, -2 output, -3 engine / corrupt
 * block (*detail = the code of fourmc_file_decode_blocks), -5 memory. */
int  fourmc_file_decompress_sharded(const char* in_name, const char* out_name, int rank, int world, long long* detail);

#ifdef __cplusplus
}
#endif
#endif
