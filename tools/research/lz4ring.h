// tools/research/lz4ring.h - constants of the group executor of the LZ4 decode (lz4_ring.hip).  The walk that finds the tokens and
// the workspace layout are lz4_seg.hip's (lz4seg.h); the exact walker finishes its blocks the same way (kResumeCode).
#ifndef FOURMC_LZ4RING_H
#define FOURMC_LZ4RING_H
#include <stdint.h>
#include <stddef.h>

namespace lz4ring {

constexpr int      kThreads   = 512;                    // one workgroup per block, one sequence per thread and step
constexpr uint32_t kWin       = 65536;                  // everything an LZ4 offset can reach
constexpr uint32_t kZero      = 8192;                   // ring bytes zeroed in front of a step (16 per thread)
constexpr uint32_t kStepB     = kZero - 16;             // bytes a step may produce
constexpr uint32_t kR         = kWin + kZero + 16;      // the ring: the window, the step's region, and the 15 bytes a 16-byte aligned zeroing starts late
constexpr uint32_t kMirror    = 64;                     // the ring's first bytes again behind its end: a string is read without wrapping
constexpr uint32_t kVbWords   = kZero / 32 + 2;         // valid bits of the step's region
constexpr uint32_t kPatBytes  = 192, kPatMax = 128;     // long matches with offsets below kPatMax are written from their period
constexpr uint32_t kMaxRounds = 4096;                   // (a step's dependency depth is at most its sequences: 512)
static_assert(kR % 16 == 0 && kStepB + 15 < kZero + 16, "ring geometry");
static_assert(kPatMax + 20 <= kPatBytes, "period scratch");

} // namespace lz4ring
#endif
