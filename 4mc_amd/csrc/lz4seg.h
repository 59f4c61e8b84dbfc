// 4mc_amd/csrc/lz4seg.h - constants shared by the segment-parallel LZ4 decoder (lz4_seg.hip) and the exact walker that finishes
// its blocks (lz4_decode.hip: lz4_decode_resume_kernel).
//
// Per block the walk kernel leaves, in the block's slot of the device workspace (32-bit words):
//   [kMetaStatus]  1: walked, the executor runs; 0: not eligible (stored / failed checksum / sizes beyond the fast path)
//   [kMetaNLive]   live segments (the ones the true token chain passes through), in chain order
//   [kMetaTailIp]  stream position of the first token the fast path does not take (its bytes end inside the last kMargin bytes)
//   [kMetaResIp], [kMetaResOp]   written by the executor: where the exact walker resumes (token position, output position)
//   [kMetaLive + 4 i ..]  live segment i: {word offset of its lists, fix records f, first true recorded record k, records c = f + n - k}
//   lists of segment j at kMetaWords + j * stride, stride = 2 * ((kFixCap + seglen / 3 + 7) & ~3) words:  kFixCap fix records, then the recorded chain.
//   A record is two words:  token position | min(literal length, kEscLL) << 23,  match offset | match length << 16  (what the executor
//   would otherwise fetch from the stream with two more loads per sequence);  kEscLL marks a sequence the executor decodes on its own
//   (literal run of 511 bytes and more, or a match length with more than two extension bytes).
#ifndef FOURMC_LZ4SEG_H
#define FOURMC_LZ4SEG_H
#include <stdint.h>
#include <stddef.h>
#include "lz4par.h"

namespace lz4seg {

constexpr int      kSegs    = 64;            // segments per block = lanes of the walk wave
constexpr uint32_t kFixCap  = 128;           // hops a re-entered segment may take before it meets its recorded chain (else: walked again)
constexpr uint32_t kMargin  = 64;            // tokens whose bytes end beyond csize - kMargin are the exact walker's
constexpr uint32_t kOMargin = 128;           // sequences whose output ends beyond cap - kOMargin are the exact walker's
constexpr uint32_t kMinSeg  = 1024;
constexpr uint32_t kMinSrc  = 256, kMinCap = 256;
constexpr uint32_t kMaxSrc  = lz4par::kSrcMax; // token positions are 23-bit fields of a record
constexpr uint32_t kEscLL   = 511;
constexpr uint32_t kPosBits = 23, kPosMask = (1u << kPosBits) - 1u;
constexpr int      kCapB    = 4032;          // bytes a batch may produce (staging buffer)
constexpr int      kPro     = 32;            // the 32 output bytes in front of a batch are kept in front of it in the buffer
constexpr int      kStage   = 4224;          // 64 + 16 + kCapB + slack, a multiple of 16

constexpr uint32_t kMetaStatus = 0, kMetaNLive = 1, kMetaTailIp = 2, kMetaResIp = 3, kMetaResOp = 4, kMetaLive = 16;
constexpr uint32_t kMetaProf   = kMetaLive + 4 * kSegs;      // 48 words: cycle counters of profiling builds (walk: 24, executor: 24)
constexpr uint32_t kMetaWords  = kMetaProf + 48;                                              // 320
constexpr uint32_t kRecWords   = 2;                                                            // words per record
constexpr uint32_t kWsWords    = (kMetaWords + kRecWords * (kSegs * (kFixCap + 8) + kMaxSrc / 3 + 512) + 3) & ~3u;
constexpr int kResumeCode = -1000000004;     // blocks[b].result while a block waits for the exact walker to finish it

} // namespace lz4seg
#endif
