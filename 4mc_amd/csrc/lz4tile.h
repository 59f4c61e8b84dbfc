// 4mc_amd/csrc/lz4tile.h - constants shared by the tile LZ4 decoder (lz4_tile.hip) and the exact walker that finishes its blocks
// (lz4_decode.hip: lz4_decode_resume_kernel).
//
// Per block the walk kernel leaves, in the block's slot of the device workspace (32-bit words):
//   [kMetaStatus]  1: walked, the executor runs; 0: not eligible (stored / failed checksum / sizes beyond the fast path)
//   [kMetaTailIp]  stream position of the first token the fast path does not take (its bytes end inside the last kMargin bytes)
//   [kMetaResIp], [kMetaResOp]   written by the executor: where the exact walker resumes (token position, output position)
//   [kMetaWords ..]  the TOKEN BITMAP: bit p & 31 of word p >> 5 says "a token of the block's chain starts at stream byte p"
//                    (exact below kMetaTailIp; nothing is promised at and above it) - 1/8 byte per stream byte: 0.53 MB for the
//                    largest stream a 4 MiB block can have (the record lists of lz4_seg.hip: 11.3 MB)
#ifndef FOURMC_LZ4TILE_H
#define FOURMC_LZ4TILE_H
#include <stdint.h>
#include <stddef.h>
#include "lz4par.h"

namespace lz4tile {

constexpr int      kSegs    = 64;            // stream segments per block = lanes of the walk wave
constexpr uint32_t kMargin  = 64;            // tokens whose bytes end beyond csize - kMargin are the exact walker's
constexpr uint32_t kOMargin = 128;           // sequences whose output ends beyond cap - kOMargin are the exact walker's
constexpr uint32_t kMinSeg  = 1024;
constexpr uint32_t kMinSrc  = 256, kMinCap = 256;
constexpr uint32_t kMaxSrc  = lz4par::kSrcMax;

constexpr int      kThreads = 512;           // executor workgroup: one thread per sequence of a chunk, one per output byte of a tile row
constexpr uint32_t kTile    = 4096;          // tile coordinates (8 per thread); a tile produces at most kTile - 8 bytes
constexpr uint32_t kSeqs    = 384;           // sequences per chunk (one per thread of the first six waves)
constexpr uint32_t kChunk   = 1536;          // stream bytes whose tokens one chunk takes: 48 bitmap words, one per lane of a wave
constexpr uint32_t kStage   = kChunk + 32 + 208;     // staged stream bytes per chunk (a multiple of 16)
constexpr uint32_t kEntries = 2 * kSeqs + 3; // source entries of a tile: 1 = bytes in front of the tile, 2 + 2 i / 3 + 2 i = literals / match of sequence i, last = bytes behind it
constexpr uint32_t kRing    = 65536;         // output window in LDS: everything an LZ4 offset can reach
constexpr uint32_t kFinal   = 0x8000u;       // state of a tile byte: kFinal | value, or (below kFinal) twice the tile coordinate of the byte it copies
constexpr uint32_t kSzClamp = (4u << 20) + 1u;

constexpr uint32_t kMetaStatus = 0, kMetaTailIp = 1, kMetaResIp = 2, kMetaResOp = 3;
constexpr uint32_t kMetaProf   = 4;                                      // 44 words: cycle counters of profiling builds
constexpr uint32_t kMetaWords  = 48;
// (+ the words segments rounded up to 32 bytes may start and end beyond the stream: segment j starts at j * seglen <= limit + 32 j, so
// with one segment per thread of the fused walk the last word written lies below nwords + kThreads; ADVICE r5: 96 words covered the
// 64 segments of the one-wave walk only, and an incompressible 4 MiB block - csize 4210754 - wrote 154 words into its neighbour's slot)
constexpr uint32_t kBmWords    = (kMaxSrc + 31) / 32 + kThreads + 32;
static_assert(kThreads >= kSegs, "bitmap slack is sized for the walk with the most segments");
constexpr uint32_t kWsWords    = (kMetaWords + kBmWords + 3) & ~3u;      // 131.6 K words = 526 KB per block
constexpr int kResumeCode = -1000000004;     // blocks[b].result while a block waits for the exact walker to finish it (= lz4seg's)

} // namespace lz4tile
#endif
