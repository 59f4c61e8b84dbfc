// 4mc_amd/csrc/lz4par.h - layout shared by the two kernels of the block-parallel LZ4 decoder
// (lz4_parse.hip: token chain -> sequence records; lz4_exec.hip: records -> output) and their launcher.
//
// Per block the parser leaves, in the block's slot of the device workspace:
//   ParHdr              status, number of sequences, decoded size, number of output windows
//   wdesc[w] (2 x uint4) for output window w, about the sequence that covers the window's first byte:
//                       {index, output start, token position, literal start}, {literal length, match length, offset, 0}
//                       (decoded once here: a sequence that spans many windows is not parsed again in each of them)
//   rec[i]  (uint2)     sequence i decoded: x = literal start | (ll & 511) << 23,  y = offset | min(ml, 2047) << 16 | (min(ll, 16383) >> 9) << 27
//                       (a saturated field means the sequence is at least a window long: it is the last of its window and the
//                       next window's descriptor has it exact)
// Output windows are kWin bytes of the block's output counted from a 128-byte aligned ADDRESS at or below the block's
// first output byte (a0 = address & 127 is the shift), so that a window is flushed with aligned 16-byte stores and a
// 128-byte line of the output belongs to exactly one window.
#ifndef FOURMC_LZ4PAR_H
#define FOURMC_LZ4PAR_H
#include <stdint.h>
#include <stddef.h>

namespace lz4par {

constexpr int      kWinLog = 10;
constexpr int      kWin    = 1 << kWinLog;                  // output bytes per window
constexpr uint32_t kDstMax = 4u << 20;                      // blocks beyond these sizes go to the exact kernel
constexpr uint32_t kSrcMax = 4210768u + 32u;                // LZ4_compressBound(4 MiB) + slack
constexpr uint32_t kMaxWin = (kDstMax + 127u) / kWin + 2u;  // windows of the largest block (+ shift, + terminator)
constexpr uint32_t kMaxSeq = kSrcMax / 3u + 8u;             // a sequence with a match takes >= 3 stream bytes

struct ParHdr {
    int32_t  status;        // kParsed: the executor runs; kDone: nothing left to do; kRetry: the exact kernel decides
    uint32_t nseq;
    uint32_t total;         // decoded bytes
    uint32_t nwin;          // windows [0, nwin) hold output
    uint32_t a0;            // shift: window w covers block output [w * kWin - a0, (w + 1) * kWin - a0)
    uint32_t dbg[11];
};
enum : int32_t { kParsed = 0, kDone = 1, kRetry = 2 };

constexpr size_t kHdrBytes   = 64;
constexpr size_t kWdescOff   = kHdrBytes;
constexpr size_t kWdescBytes = size_t(kMaxWin + 2) * 32;
constexpr size_t kTokOff     = (kWdescOff + kWdescBytes + 255) & ~size_t(255);
constexpr size_t kTokBytes   = size_t(kMaxSeq) * 8;
constexpr size_t kDbgOff     = (kTokOff + kTokBytes + 255) & ~size_t(255);   // cycle counters of profiling builds
constexpr size_t kDbgBytes   = 1024;
constexpr size_t kSlotBytes  = kDbgOff + kDbgBytes;

constexpr uint32_t kMaxBatch = 2048;         // blocks per launch pair: bounds the workspace (kMaxBatch * kSlotBytes)
constexpr uint32_t kRecLLSat = 16383, kRecMLSat = 2047;
constexpr int kRetryCode = -1000000003;     // blocks[b].result while a block waits for the exact kernel

} // namespace lz4par
#endif
