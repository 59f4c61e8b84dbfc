/*
 * oracle/oracle.h — CPU restatement of the 4mc hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * nothing under 4mc_amd/ links, imports or calls it.
 *
 * Every function restates, in plain scalar C written for this repository, the algorithm of the
 * reference function named beside it (paths relative to /root/reference).  Parity status:
 * PINNED — checked against (a) the reference's own compiled sources (oracle/_ref, built by
 * oracle/Makefile from the tree where it lies) on seeded and fuzzed inputs, and (b) the golden
 * vectors of SURVEY.md §8(c), committed under tests/golden/ with their generator.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BLOCKSIZE   (4u * 1024u * 1024u)   /* native/4mc.c:116  FOURMC_BLOCKSIZE        */
#define ORC_MAGIC_4MC   0x344D4300u            /* native/4mc.c:111                           */
#define ORC_MAGIC_4MZ   0x344D5A00u            /* native/4mc.c:112                           */

/* XXH32 — native/lz4/xxhash.c:392-415 (round :276, finalize :291-345, avalanche :283). */
uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed);

/* LZ4_compressBound — native/lz4/lz4.h:212 (LZ4_COMPRESSBOUND). */
int orc_lz4_compress_bound(int n);

/* LZ4_compress_default (acceleration 1, 64-bit little-endian build) —
 * native/lz4/lz4.c:1435 -> :1416 -> :1346-1367 -> :910-1302.
 * Returns bytes written, or 0 when the output does not fit `cap` (limitedOutput). */
int orc_lz4_compress_fast(const uint8_t* src, uint8_t* dst, int n, int cap);

/* LZ4_compress_HC, hash-chain levels 1..8 (4mc uses 4 and 8) - native/lz4/lz4hc.c:958-973 -> :553-788.
 * Returns bytes written, 0 when it does not fit `cap`, -2 for levels this port does not cover. */
int orc_lz4hc_compress(const uint8_t* src, uint8_t* dst, int n, int cap, int level);

/* LZ4_decompress_safe — native/lz4/lz4.c:2345-2350 -> :1936-2339 (noDict, full block).
 * Returns decoded size (>=0) or a negative error. */
int orc_lz4_decompress_safe(const uint8_t* src, uint8_t* dst, int csize, int cap);
int orc_lz4_sequences(const uint8_t* src, int csize, int cap, uint32_t* tokpos, uint32_t* outpos, int maxseq, int* total);
int orc_lz4_sequences_ex(const uint8_t* src, int csize, int cap, uint32_t* tokpos, uint32_t* outpos, uint32_t* fields, int maxseq, int* total);

/* ---- container (native/4mc.c:220-386 writer, :560-707 reader; format spec 4mc-format-spec) */
typedef int (*orc_block_codec_fn)(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);

size_t  orc_container_bound(size_t n);
/* Writes header + blocks + end mark + footer.  `compress` is called once per <=4 MiB block with
 * cap = n-1 (native/4mc.c:301); a return <= 0 means "store raw" (native/4mc.c:318-329). */
int64_t orc_container_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                               uint32_t magic, orc_block_codec_fn compress, void* ctx);
/* Reads ONE stream (header..footer).  Returns decoded bytes (>=0) and sets *consumed, or
 * -(exit code) as the reference CLI would exit (2 input / 4 content; native/4mc.c:135-161). */
int64_t orc_container_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                                 uint32_t magic, orc_block_codec_fn decompress, void* ctx,
                                 size_t* consumed);

/* ZSTD_decompress for 4mz payloads (zstd_port.cpp) - native/zstd/decompress/zstd_decompress.c:1112.
 * Returns decoded bytes or < 0 where the reference reports an error. */
int64_t orc_zstd_decompress(const uint8_t* src, size_t csize, uint8_t* dst, size_t cap);
int     orc_codec_zstd_decode(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);

/* ZSTD_compress(dst, cap, src, n, level) for the 4mz encode path (zstd_enc_port.c) -
 * native/zstd/compress/zstd_compress.c:4806 as called by native/4mc.c:467.  Returns the frame size,
 * a negative ZSTD error number (-70 = dstSize_tooSmall), or ORC_ZSTD_UNSUPPORTED for levels the
 * port does not restate (only level 1 / strategy "fast" so far). */
#define ORC_ZSTD_UNSUPPORTED (-1000)
int64_t orc_zstd_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int level);
int     orc_codec_zstd1(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);

/* codec adaptors with the orc_block_codec_fn shape (ctx unused) */
int orc_codec_lz4_fast(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);
int orc_codec_lz4_decode(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);
int orc_codec_lz4hc4(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);
int orc_codec_lz4hc8(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap);

#ifdef __cplusplus
}
#endif
#endif
