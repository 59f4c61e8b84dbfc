// 4mc_amd/csrc/zstd_encode.hip — K6: batched ZSTD frame encode on gfx950, BYTE-IDENTICAL to the
// reference's ZSTD_compress(dst, cap, src, n, level) (zstd 1.5.3; levels 1 / 3 / 6 / 12), the call 4mz makes per
// 4 MiB block at its "fast" level (native/4mc.c:411-412,:467) and Java makes through
// ZstdCompressor.compressBytesDirect (native/jniZstdCompressor.c:93).
//
//   parameters   ZSTD_getCParams_internal / ZSTD_adjustCParams_internal   compress/zstd_compress.c:6465-6488,:1335-1399
//   frame        ZSTD_writeFrameHeader :4065, ZSTD_compress_frameChunk :3983, ZSTD_compressBlock_internal :3812
//   match finder ZSTD_compressBlock_fast_noDict_generic                   compress/zstd_fast.c:95-365
//   literals     ZSTD_compressLiterals + HUF_compress_internal            compress/zstd_compress_literals.c:100, huf_compress.c:1250
//   sequences    ZSTD_buildSequencesStatistics / ZSTD_encodeSequences     compress/zstd_compress.c:2489, zstd_compress_sequences.c:302
//   FSE          normalizeCount / writeNCount / buildCTable               compress/fse_compress.c:68-520
//
// One wavefront owns one 4mc block (one zstd frame of up to 32 blocks of 128 KiB).  Per 128 KiB
// block it (1) runs the greedy match finder, (2) entropy-codes literals and sequences.
//   * Match finder (levels 1 and 3): the reference walks positions one or two at a time with a repcode test ahead; every
//     tested position reads then overwrites its hash slot(s).  Two shapes reproduce that walk exactly.  The DENSE WINDOW (see
//     fast_block / dfast_block): lane l takes position sp + l whatever role the walk will give it; one table round trip and
//     one candidate round trip serve 64 positions; byte-equality masks of 48 candidate bytes give the tests, the lengths and
//     the repcode tests behind a chosen match; a scalar walk (a readlane chain for runs of plain hash hits) only chooses,
//     sequences and table writes follow from the chosen lanes for all lanes at once.  The BATCHED SEARCH: lane j
//     speculatively executes pair / position j of the current search, a ballot picks the first event in serial order, only
//     lanes up to it commit their table writes, lanes that touch the same slot are ordered by cutting the batch (LDS
//     atomic-min scoreboard) - one sequence per batch; it takes what the window hands over (steps > 2, block ends, long
//     catch-ups, slots written twice inside a window) and is the cross-check path (FOURMC_ZSTD_SERIAL=1).  The hash tables
//     (<= 2^17 x u32) live in the block's HBM workspace slot; LDS per block is 19 984 bytes so that eight blocks share a CU.
//   * Entropy stage: histograms are wave-parallel (LDS atomics); Huffman bit packing is wave-parallel
//     (8 symbols per lane, DPP prefix sum of code lengths, LDS atomic-or staging); table construction
//     (Huffman tree, FSE normalisation/spread) and the three interleaved FSE state chains are serial
//     by nature and run wave-uniform, fed 64 sequences at a time through readlane.
// Per block HBM traffic: n bytes read (+ candidate re-reads, mostly L2/MALL hits), the sequence
// store and literal buffer (<= 0.7 MiB, L2 resident), csize written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devenc.h"

namespace {

constexpr int      kErrGeneric = -1, kErrTooSmall = -70;     // ZSTD_error_GENERIC / _dstSize_tooSmall
constexpr uint32_t kSub    = 128 * 1024;                     // ZSTD_BLOCKSIZE_MAX
constexpr uint32_t kSeqCap = 32768 + 64;
constexpr size_t   kSeqBytes  = size_t(kSeqCap) * 4;
constexpr size_t   kCodeBytes = kSeqCap;
constexpr size_t   kLitPad    = 64;
constexpr size_t   kLitBytes  = kLitPad + kSub + 256;
// per-block workspace: [sequence store | codes | literal buffer (+ profile counters) | previous FSE tables | scratch] [tables]
constexpr size_t   kOffCodes  = 3 * kSeqBytes;
constexpr size_t   kOffLit    = kOffCodes + 3 * kCodeBytes;
constexpr size_t   kOffPrev   = kOffLit + kLitBytes;                 // 3 x FseCt of the last confirmed block (lazy strategies)
constexpr size_t   kOffTmp    = kOffPrev + 4736;                     // 512 B: FSE_writeNCount trial output (ZSTD_NCountCost)
constexpr size_t   kStoreBytes = kOffTmp + 512 + 192;
static_assert(kStoreBytes % 64 == 0, "tables start 64-byte aligned");
// clevels.h:25-130, levels 1..12: {windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy} for inputs > 256 KB, <= 256 KB,
// <= 128 KB, <= 16 KB.  Strategies as the reference numbers them: 1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2, 6 btlazy2, 7 btopt (levels 13+ need
// btultra / btlazy2 on full blocks: not on the device).  4mz uses 1, 3, 6, 12; the JNI name compressBytesDirectHC(level) may pass any.
struct LevelRow { uint8_t wlog, clog, hlog, slog, mml, tlen, strat; };
constexpr int kMaxLevel = 12;
#define FOURMC_ZSTD_LEVEL_ROWS { \
    {{19, 13, 14, 1, 7, 0, 1}, {18, 13, 14, 1, 6, 0, 1}, {17, 12, 13, 1, 6, 0, 1}, {14, 14, 15, 1, 5, 0, 1}}, \
    {{20, 15, 16, 1, 6, 0, 1}, {18, 14, 14, 1, 5, 0, 2}, {17, 13, 15, 1, 5, 0, 1}, {14, 14, 15, 1, 4, 0, 1}}, \
    {{21, 16, 17, 1, 5, 0, 2}, {18, 16, 16, 1, 4, 0, 2}, {17, 15, 16, 2, 5, 0, 2}, {14, 14, 15, 2, 4, 0, 2}}, \
    {{21, 18, 18, 1, 5, 0, 2}, {18, 16, 17, 3, 5, 2, 3}, {17, 17, 17, 2, 4, 0, 2}, {14, 14, 14, 4, 4, 2, 3}}, \
    {{21, 18, 19, 3, 5, 2, 3}, {18, 17, 18, 5, 5, 2, 3}, {17, 16, 17, 3, 4, 2, 3}, {14, 14, 14, 3, 4, 4, 4}}, \
    {{21, 18, 19, 3, 5, 4, 4}, {18, 18, 19, 3, 5, 4, 4}, {17, 16, 17, 3, 4, 4, 4}, {14, 14, 14, 4, 4, 8, 5}}, \
    {{21, 19, 20, 4, 5, 8, 4}, {18, 18, 19, 4, 4, 4, 4}, {17, 16, 17, 3, 4, 8, 5}, {14, 14, 14, 6, 4, 8, 5}}, \
    {{21, 19, 20, 4, 5, 16, 5}, {18, 18, 19, 4, 4, 8, 5}, {17, 16, 17, 4, 4, 8, 5}, {14, 14, 14, 8, 4, 8, 5}}, \
    {{22, 20, 21, 4, 5, 16, 5}, {18, 18, 19, 5, 4, 8, 5}, {17, 16, 17, 5, 4, 8, 5}, {14, 15, 14, 5, 4, 8, 6}}, \
    {{22, 21, 22, 5, 5, 16, 5}, {18, 18, 19, 6, 4, 8, 5}, {17, 16, 17, 6, 4, 8, 5}, {14, 15, 14, 9, 4, 8, 6}}, \
    {{22, 21, 22, 6, 5, 16, 5}, {18, 18, 19, 5, 4, 12, 6}, {17, 17, 17, 5, 4, 8, 6}, {14, 15, 14, 3, 4, 12, 7}}, \
    {{22, 22, 23, 6, 5, 32, 5}, {18, 19, 19, 7, 4, 12, 6}, {17, 18, 17, 7, 4, 12, 6}, {14, 15, 14, 4, 3, 24, 7}}}
__device__ __constant__ const LevelRow kLevelRowsDev[kMaxLevel][4] = FOURMC_ZSTD_LEVEL_ROWS;     // what the kernels read
constexpr LevelRow kLevelRows[kMaxLevel][4] = FOURMC_ZSTD_LEVEL_ROWS;                             // the same rows for the host (launcher, table sizes)
// per-block table area by level.  The levels 4mz uses keep the sizes they always had: 1 -> hash table (<= 2^15 x u32); 3 -> long 2^17 +
// short 2^16; 6 -> rows 2^19 x u32 + tags 2^19 x u16; 12 -> rows 2^23 x u32 + tags 2^23 x u16 (inputs > 256 KiB), hash 2^19 x u32 + binary
// tree 2^19 x u32 (btlazy2, smaller inputs).  Any other level: room for the largest hash table of its four rows (entries + tags, or the long
// table of dfast at its fixed place), the largest chain / short / tree table behind it, and the optimal parser's arrays.
__host__ __device__ constexpr size_t table_bytes(int level)
{
    // largest hashLog / chainLog of a level's four rows (checked against the rows below)
    constexpr uint8_t hmax[kMaxLevel] = {15, 16, 17, 18, 19, 19, 20, 20, 21, 22, 22, 23}, cmax[kMaxLevel] = {14, 15, 16, 18, 18, 18, 19, 19, 20, 21, 21, 22};
    return level == 12 ? (size_t(4) << 23) + (size_t(2) << 23) : level == 6 ? (size_t(4) << 19) + (size_t(2) << 19) : level == 3 ? (size_t(4) << 17) + (size_t(4) << 16)
         : level == 1 ? (size_t(4) << 15)
         : level >= 1 && level <= kMaxLevel ? (size_t(6) << (hmax[level - 1] < 17 ? 17 : hmax[level - 1])) + (size_t(8) << cmax[level - 1]) + (size_t(1) << 20) : 0;
}

constexpr bool level_maxima_ok()
{
    constexpr uint8_t hmax[kMaxLevel] = {15, 16, 17, 18, 19, 19, 20, 20, 21, 22, 22, 23}, cmax[kMaxLevel] = {14, 15, 16, 18, 18, 18, 19, 19, 20, 21, 21, 22};
    for (int l = 0; l < kMaxLevel; l++) {
        uint8_t h = 0, c = 0;
        for (int k = 0; k < 4; k++) { if (kLevelRows[l][k].hlog > h) h = kLevelRows[l][k].hlog; if (kLevelRows[l][k].clog > c) c = kLevelRows[l][k].clog; }
        if (h != hmax[l] || c != cmax[l]) return false;
    }
    return true;
}
static_assert(level_maxima_ok(), "table_bytes: the maxima do not match the level rows");

struct __attribute__((packed, aligned(1))) S4B { uint32_t v; };
__device__ __forceinline__ void st4(uint8_t* p, uint32_t v) { reinterpret_cast<S4B*>(p)->v = v; }
__device__ __forceinline__ uint32_t U(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
__device__ __forceinline__ int Ui(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long U64(unsigned long long v) { return (uint64_t(U(uint32_t(v >> 32))) << 32) | U(uint32_t(v)); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t l) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(l))); }
__device__ __forceinline__ int hibit(uint32_t v) { return 31 - __clz(v); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t wave_max(uint32_t v)          // wave64 maximum (values >= 0) on the DPP network, uniform result
{
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v)); v = max(v, dpp0<0x143, 0xc>(v));
    return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}
__device__ __forceinline__ uint32_t scan_add(uint32_t v)          // wave64 inclusive prefix sum
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v);
    v += dpp0<0x143, 0xc>(v);
    return v;
}

struct FseCt { uint16_t next[512]; uint32_t dbits[64]; int32_t dfind[64]; uint32_t log, maxsym; };

struct ZLds {
    uint32_t count[256];
    uint32_t ncount[516];                 // Huffman nodes, index + 1 (entry 0 is the barrier node)
    uint16_t nparent[516];
    uint8_t  nbyte[516], nbits[516];
    uint16_t rank_base[192], rank_curr[192];
    uint32_t qstack[256];
    uint32_t huf[2][256];                 // literal tables of the previous / next block: code | nbits << 16
    uint32_t fresh[256];
    FseCt    ct[3];                       // ll, of, ml   (ct[0] doubles as the Huffman-weight table)
    int16_t  norm[64];
    uint16_t cumul[72];
    uint8_t  symbol_at[512];
    uint8_t  weight[264];
    uint32_t wcount[16];
    uint32_t rank_last[16];
    uint16_t per_rank[16], val[16];
    uint32_t score[1024];
};                                       // 19 984 B: eight blocks per CU (160 KiB of LDS); the bit packers stage their output in `count` (free by then)

// ------------------------------------------------------------------------------------------------ bit writer
// serial writer (wave-uniform state, lane 0 stores); overflow rule of BIT_CStream_t (bitstream.h:153-240):
// the stream does not fit when floor(total_bits / 8) + 8 >= cap.
struct BitW {
    uint8_t* p; uint32_t cap, pos, nb, total; uint64_t acc;
    __device__ __forceinline__ bool init(uint8_t* p_, uint32_t cap_) { p = p_; cap = cap_; pos = 0; nb = 0; total = 0; acc = 0; return cap_ > 8; }
    __device__ __forceinline__ void put(uint64_t v, uint32_t n, int lane)       // n <= 32
    {
        acc |= (v & ((1ull << n) - 1)) << nb; nb += n; total += n;
        if (nb >= 32) { if (lane == 0 && pos + 4 <= cap) st4(p + pos, uint32_t(acc)); pos += 4; acc >>= 32; nb -= 32; }
    }
    __device__ __forceinline__ uint32_t close(int lane)
    {
        put(1, 1, lane);
        for (uint32_t k = 0; 8 * k < nb; k++) if (lane == 0 && pos + k < cap) p[pos + k] = uint8_t(acc >> (8 * k));
        return ((total >> 3) + 8 < cap) ? (total + 7) >> 3 : 0u;
    }
};

// ------------------------------------------------------------------------------------------------ histograms
__device__ __forceinline__ void hist_finish(ZLds& L, uint32_t& largest, uint32_t& max_sym, int lane)
{
    uint32_t lg = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t c = L.count[4 * lane + k]; lg = max(lg, c); if (c) hi = uint32_t(4 * lane + k + 1); }
    largest = wave_max(lg);
    hi = wave_max(hi);
    max_sym = hi ? hi - 1 : 0;
}

// byte histogram of s[0, n) (HBM) into L.count (hist.c:31-60)
__device__ __forceinline__ void hist_bytes(ZLds& L, const uint8_t* s, uint32_t n, uint32_t& largest, uint32_t& max_sym, int lane)
{
    for (int i = lane; i < 256; i += 64) L.count[i] = 0;
    uint32_t i = uint32_t(lane) * 16;
    for (; i + 16 <= n; i += 1024) {
        const U16B q = *reinterpret_cast<const U16B*>(s + i);
#pragma unroll
        for (int k = 0; k < 8; k++) { atomicAdd(&L.count[(q.a >> (8 * k)) & 255], 1u); atomicAdd(&L.count[(q.b >> (8 * k)) & 255], 1u); }
    }
    if (i < n) for (uint32_t k = i; k < n; k++) atomicAdd(&L.count[s[k]], 1u);
    hist_finish(L, largest, max_sym, lane);
}

// ------------------------------------------------------------------------------------------------ FSE (wave-uniform serial)
__device__ __forceinline__ uint32_t fse_min_log(uint32_t n, uint32_t max_sym)
{ return min(uint32_t(hibit(n)) + 1, uint32_t(hibit(max_sym)) + 2); }

__device__ __forceinline__ uint32_t fse_optimal_log(uint32_t max_log, uint32_t n, uint32_t max_sym, uint32_t minus)
{
    const uint32_t src_bits = uint32_t(hibit(n - 1)) - minus;
    uint32_t log = max_log;
    if (src_bits < log) log = src_bits;
    log = max(log, fse_min_log(n, max_sym));
    return min(max(log, 5u), 12u);
}

// FSE_normalizeCount on L.count -> L.norm; 0 or kErrGeneric
__device__ __forceinline__ int fse_normalize(ZLds& L, uint32_t log, uint32_t total0, uint32_t max_sym, bool use_low)
{
    const int16_t low = use_low ? int16_t(-1) : int16_t(1);
    const uint64_t scale = 62 - log, step = (1ull << 62) / total0, vstep = 1ull << (scale - 20);
    const uint32_t low_thr = total0 >> log;
    int remaining = 1 << log;
    uint32_t largest = 0; int largest_p = 0;
    if (log < fse_min_log(total0, max_sym)) return kErrGeneric;
    for (uint32_t s = 0; s <= max_sym; s++) {
        const uint32_t c = L.count[s];
        if (c == total0) return 0;
        if (!c) { L.norm[s] = 0; continue; }
        if (c <= low_thr) { L.norm[s] = low; remaining--; continue; }
        int p = int(int16_t((c * step) >> scale));
        if (p < 8) {
            const uint32_t rtb = p == 0 ? 0u : p == 1 ? 473195u : p == 2 ? 504333u : p == 3 ? 520860u : p == 4 ? 550000u : p == 5 ? 700000u : p == 6 ? 750000u : 830000u;
            p += ((c * step) - (uint64_t(p) << scale) > vstep * rtb) ? 1 : 0;
        }
        if (p > largest_p) { largest_p = p; largest = s; }
        L.norm[s] = int16_t(p); remaining -= p;
    }
    if (-remaining < (int(L.norm[largest]) >> 1)) { L.norm[largest] = int16_t(L.norm[largest] + remaining); return 0; }
    // FSE_normalizeM2 (fse_compress.c:373-455)
    uint64_t total = total0;
    uint32_t distributed = 0, todo;
    uint32_t low_one = uint32_t((total * 3) >> (log + 1));
    for (uint32_t s = 0; s <= max_sym; s++) {
        const uint32_t c = L.count[s];
        if (!c) { L.norm[s] = 0; continue; }
        if (c <= low_thr) { L.norm[s] = low; distributed++; total -= c; continue; }
        if (c <= low_one) { L.norm[s] = 1; distributed++; total -= c; continue; }
        L.norm[s] = -2;
    }
    todo = (1u << log) - distributed;
    if (!todo) return 0;
    if (total / todo > low_one) {
        low_one = uint32_t((total * 3) / (uint64_t(todo) * 2));
        for (uint32_t s = 0; s <= max_sym; s++)
            if (L.norm[s] == -2 && L.count[s] <= low_one) { L.norm[s] = 1; distributed++; total -= L.count[s]; }
        todo = (1u << log) - distributed;
    }
    if (distributed == max_sym + 1) {
        uint32_t best = 0, bc = 0;
        for (uint32_t s = 0; s <= max_sym; s++) if (L.count[s] > bc) { best = s; bc = L.count[s]; }
        L.norm[best] = int16_t(L.norm[best] + int(todo));
        return 0;
    }
    if (!total) {
        for (uint32_t s = 0; todo > 0; s = (s + 1) % (max_sym + 1)) if (L.norm[s] > 0) { todo--; L.norm[s] = int16_t(L.norm[s] + 1); }
        return 0;
    }
    {
        const uint64_t vlog = 62 - log, mid = (1ull << (vlog - 1)) - 1;
        const uint64_t rstep = (((1ull << vlog) * todo) + mid) / uint32_t(total);
        uint64_t acc = mid;
        for (uint32_t s = 0; s <= max_sym; s++) if (L.norm[s] == -2) {
            const uint64_t end = acc + L.count[s] * rstep;
            const uint32_t w = uint32_t(end >> vlog) - uint32_t(acc >> vlog);
            if (w < 1) return kErrGeneric;
            L.norm[s] = int16_t(w); acc = end;
        }
    }
    return 0;
}

// FSE_writeNCount of L.norm; bytes written or kErr*
__device__ __forceinline__ int fse_write_ncount(ZLds& L, uint8_t* out, uint32_t cap, uint32_t max_sym, uint32_t log, int lane)
{
    const uint32_t bound = max_sym ? (((max_sym + 1) * log + 4 + 2) / 8) + 1 + 2 : 512u;
    const bool safe = cap >= bound;
    const uint32_t alphabet = max_sym + 1;
    uint32_t o = 0, bs = log - 5, sym = 0;
    int nbits = int(log) + 1, remaining = (1 << log) + 1, threshold = 1 << log, bc = 4;
    bool prev0 = false;
    auto flush16 = [&]() -> bool {
        if (!safe && o + 2 > cap) return false;
        if (lane == 0) { out[o] = uint8_t(bs); out[o + 1] = uint8_t(bs >> 8); }
        o += 2; bs >>= 16; return true;
    };
    while (sym < alphabet && remaining > 1) {
        if (prev0) {
            uint32_t start = sym;
            while (sym < alphabet && !L.norm[sym]) sym++;
            if (sym == alphabet) break;
            while (sym >= start + 24) { start += 24; bs += 0xFFFFu << bc; if (!flush16()) return kErrTooSmall; }
            while (sym >= start + 3) { start += 3; bs += 3u << bc; bc += 2; }
            bs += (sym - start) << bc; bc += 2;
            if (bc > 16) { if (!flush16()) return kErrTooSmall; bc -= 16; }
        }
        {
            int c = L.norm[sym++];
            const int mx = (2 * threshold - 1) - remaining;
            remaining -= c < 0 ? -c : c;
            c++;
            if (c >= threshold) c += mx;
            bs += uint32_t(c) << bc;
            bc += nbits; bc -= (c < mx) ? 1 : 0;
            prev0 = (c == 1);
            if (remaining < 1) return kErrGeneric;
            while (remaining < threshold) { nbits--; threshold >>= 1; }
        }
        if (bc > 16) { if (!flush16()) return kErrTooSmall; bc -= 16; }
    }
    if (remaining != 1) return kErrGeneric;
    if (!safe && o + 2 > cap) return kErrTooSmall;
    if (lane == 0) { out[o] = uint8_t(bs); out[o + 1] = uint8_t(bs >> 8); }
    o += uint32_t(bc + 7) / 8;
    return int(o);
}

// FSE_buildCTable_wksp from norm[] (LDS `norm` or the default distributions passed through L.norm)
__device__ __forceinline__ void fse_build(ZLds& L, FseCt& ct, uint32_t max_sym, uint32_t log)
{
    const uint32_t size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint32_t high = size - 1, pos = 0, total = 0;
    ct.log = log; ct.maxsym = max_sym;
    L.cumul[0] = 0;
    for (uint32_t s = 0; s <= max_sym; s++) {
        const int n = L.norm[s];
        if (n == -1) { L.cumul[s + 1] = uint16_t(L.cumul[s] + 1); L.symbol_at[high--] = uint8_t(s); }
        else L.cumul[s + 1] = uint16_t(L.cumul[s] + n);
    }
    for (uint32_t s = 0; s <= max_sym; s++) {
        const int n = L.norm[s];
        for (int i = 0; i < n; i++) {
            L.symbol_at[pos] = uint8_t(s);
            do pos = (pos + step) & mask; while (pos > high);
        }
    }
    for (uint32_t u = 0; u < size; u++) {
        const uint32_t s = L.symbol_at[u];
        const uint32_t c = L.cumul[s];
        ct.next[c] = uint16_t(size + u);
        L.cumul[s] = uint16_t(c + 1);
    }
    for (uint32_t s = 0; s <= max_sym; s++) {
        const int n = L.norm[s];
        if (n == 0) { ct.dbits[s] = ((log + 1) << 16) - size; ct.dfind[s] = 0; }
        else if (n == -1 || n == 1) { ct.dbits[s] = (log << 16) - size; ct.dfind[s] = int32_t(total) - 1; total++; }
        else {
            const uint32_t max_out = log - uint32_t(hibit(uint32_t(n) - 1));
            ct.dbits[s] = (max_out << 16) - (uint32_t(n) << max_out);
            ct.dfind[s] = int32_t(total) - n;
            total += uint32_t(n);
        }
    }
}

__device__ __forceinline__ uint32_t fse_first_state(const FseCt& ct, uint32_t sym)
{
    const uint32_t d = ct.dbits[sym], nb = (d + (1u << 15)) >> 16, v = (nb << 16) - d;
    return ct.next[int32_t(v >> nb) + ct.dfind[sym]];
}

// ------------------------------------------------------------------------------------------------ Huffman
__device__ __forceinline__ uint32_t huf_bucket(uint32_t c) { return c < 165 ? c : uint32_t(hibit(c)) + 158; }

__device__ __forceinline__ void node_swap(ZLds& L, int a, int b)     // a, b: node indices (+1 applied by caller)
{
    const uint32_t c = L.ncount[a]; const uint8_t y = L.nbyte[a];
    L.ncount[a] = L.ncount[b]; L.nbyte[a] = L.nbyte[b];
    L.ncount[b] = c; L.nbyte[b] = y;
}

// HUF_simpleQuickSort on nodes [base+lo, base+hi] (descending); deferred sub-ranges are disjoint, so
// processing them from a stack gives the same array as the reference's recursion order.
__device__ __forceinline__ void huf_qsort(ZLds& L, int base, int lo0, int hi0)
{
    int sp = 0;
    L.qstack[sp++] = uint32_t(lo0) | (uint32_t(hi0) << 16);
    while (sp > 0) {
        const uint32_t f = L.qstack[--sp];
        int lo = int(f & 0xFFFF), hi = int(f >> 16);
        if (hi - lo < 8) {
            for (int i = lo + 1; i <= hi; i++) {
                const uint32_t kc = L.ncount[base + i]; const uint8_t kb = L.nbyte[base + i];
                int j = i - 1;
                while (j >= lo && L.ncount[base + j] < kc) { L.ncount[base + j + 1] = L.ncount[base + j]; L.nbyte[base + j + 1] = L.nbyte[base + j]; j--; }
                L.ncount[base + j + 1] = kc; L.nbyte[base + j + 1] = kb;
            }
            continue;
        }
        while (lo < hi) {
            const uint32_t pivot = L.ncount[base + hi];
            int i = lo - 1;
            for (int j = lo; j < hi; j++) if (L.ncount[base + j] > pivot) { i++; node_swap(L, base + i, base + j); }
            node_swap(L, base + i + 1, base + hi);
            const int idx = i + 1;
            if (idx - lo < hi - idx) { if (idx - 1 > lo) L.qstack[sp++] = uint32_t(lo) | (uint32_t(idx - 1) << 16); lo = idx + 1; }
            else                     { if (hi > idx + 1) L.qstack[sp++] = uint32_t(idx + 1) | (uint32_t(hi) << 16); hi = idx - 1; }
        }
    }
}

// HUF_buildCTable_wksp on L.count -> L.fresh (code | nbits << 16); returns the table log
__device__ __forceinline__ uint32_t huf_build(ZLds& L, uint32_t max_sym, uint32_t max_bits, int lane)
{
    constexpr int N0 = 1;                                     // node k lives at array index k + N0
    for (int i = lane; i < 516; i += 64) { L.ncount[i] = 0; L.nparent[i] = 0; L.nbyte[i] = 0; L.nbits[i] = 0; }
    for (int i = lane; i < 192; i += 64) { L.rank_base[i] = 0; L.rank_curr[i] = 0; }
    // HUF_sort (:604-640)
    for (uint32_t s = 0; s <= max_sym; s++) { const uint32_t r = huf_bucket(L.count[s]); L.rank_base[r] = uint16_t(L.rank_base[r] + 1); }
    for (int r = 191; r > 0; r--) { const uint16_t v = uint16_t(L.rank_base[r - 1] + L.rank_base[r]); L.rank_base[r - 1] = v; L.rank_curr[r - 1] = v; }
    for (uint32_t s = 0; s <= max_sym; s++) {
        const uint32_t c = L.count[s], r = huf_bucket(c) + 1;
        const uint32_t pos = L.rank_curr[r];
        L.rank_curr[r] = uint16_t(pos + 1);
        L.ncount[N0 + pos] = c; L.nbyte[N0 + pos] = uint8_t(s);
    }
    for (int r = 165; r < 191; r++) {
        const int len = int(L.rank_curr[r]) - int(L.rank_base[r]);
        if (len > 1) huf_qsort(L, N0 + int(L.rank_base[r]), 0, len - 1);
    }
    // HUF_buildTree (:665-705)
    int last = int(max_sym);
    while (!L.ncount[N0 + last]) last--;
    int low_s = last, nb = 256, low_n = 256;
    const int root = nb + low_s - 1;
    L.ncount[N0 + nb] = L.ncount[N0 + low_s] + L.ncount[N0 + low_s - 1];
    L.nparent[N0 + low_s] = uint16_t(nb); L.nparent[N0 + low_s - 1] = uint16_t(nb);
    nb++; low_s -= 2;
    for (int n = nb; n <= root; n++) L.ncount[N0 + n] = 1u << 30;
    L.ncount[0] = 1u << 31;
    while (nb <= root) {
        const int n1 = L.ncount[N0 + low_s] < L.ncount[N0 + low_n] ? low_s-- : low_n++;
        const int n2 = L.ncount[N0 + low_s] < L.ncount[N0 + low_n] ? low_s-- : low_n++;
        L.ncount[N0 + nb] = L.ncount[N0 + n1] + L.ncount[N0 + n2];
        L.nparent[N0 + n1] = uint16_t(nb); L.nparent[N0 + n2] = uint16_t(nb);
        nb++;
    }
    L.nbits[N0 + root] = 0;
    for (int n = root - 1; n >= 256; n--) L.nbits[N0 + n] = uint8_t(L.nbits[N0 + L.nparent[N0 + n]] + 1);
    for (int n = 0; n <= last; n++) L.nbits[N0 + n] = uint8_t(L.nbits[N0 + L.nparent[N0 + n]] + 1);
    // HUF_setMaxHeight (:360-470)
    {
        const uint32_t largest = L.nbits[N0 + last], target = max_bits;
        if (largest > target) {
            int cost = 0, n = last;
            const int base = 1 << (largest - target);
            while (L.nbits[N0 + n] > target) { cost += base - (1 << (largest - L.nbits[N0 + n])); L.nbits[N0 + n] = uint8_t(target); n--; }
            while (L.nbits[N0 + n] == target) n--;
            cost >>= (largest - target);
            for (int i = 0; i < 16; i++) L.rank_last[i] = 0xF0F0F0F0u;
            {
                uint32_t cur = target;
                for (int pos = n; pos >= 0; pos--) {
                    const uint32_t b = L.nbits[N0 + pos];
                    if (b >= cur) continue;
                    cur = b;
                    L.rank_last[target - cur] = uint32_t(pos);
                }
            }
            while (cost > 0) {
                uint32_t dec = uint32_t(hibit(uint32_t(cost))) + 1;
                for (; dec > 1; dec--) {
                    const uint32_t hp = L.rank_last[dec], lp = L.rank_last[dec - 1];
                    if (hp == 0xF0F0F0F0u) continue;
                    if (lp == 0xF0F0F0F0u) break;
                    if (L.ncount[N0 + hp] <= 2 * L.ncount[N0 + lp]) break;
                }
                while (dec <= 12 && L.rank_last[dec] == 0xF0F0F0F0u) dec++;
                cost -= 1 << (dec - 1);
                { const uint32_t q = L.rank_last[dec]; L.nbits[N0 + q] = uint8_t(L.nbits[N0 + q] + 1); }
                if (L.rank_last[dec - 1] == 0xF0F0F0F0u) L.rank_last[dec - 1] = L.rank_last[dec];
                if (L.rank_last[dec] == 0) L.rank_last[dec] = 0xF0F0F0F0u;
                else {
                    const uint32_t q = L.rank_last[dec] - 1;
                    L.rank_last[dec] = (L.nbits[N0 + q] != target - dec) ? 0xF0F0F0F0u : q;
                }
            }
            while (cost < 0) {
                if (L.rank_last[1] == 0xF0F0F0F0u) {
                    while (L.nbits[N0 + n] == target) n--;
                    L.nbits[N0 + n + 1] = uint8_t(L.nbits[N0 + n + 1] - 1);
                    L.rank_last[1] = uint32_t(n + 1);
                    cost++;
                    continue;
                }
                { const uint32_t q = L.rank_last[1] + 1; L.nbits[N0 + q] = uint8_t(L.nbits[N0 + q] - 1); L.rank_last[1] = q; }
                cost++;
            }
            max_bits = target;
        } else max_bits = largest;
    }
    // HUF_buildCTableFromTree (:714-735)
    for (int i = 0; i < 16; i++) { L.per_rank[i] = 0; L.val[i] = 0; }
    for (int n = 0; n <= last; n++) { const uint32_t b = L.nbits[N0 + n]; L.per_rank[b] = uint16_t(L.per_rank[b] + 1); }
    {
        uint32_t mn = 0;
        for (int r = int(max_bits); r > 0; r--) { L.val[r] = uint16_t(mn); mn = (mn + L.per_rank[r]) >> 1; }
    }
    for (int i = lane; i < 256; i += 64) L.fresh[i] = 0;
    for (uint32_t n = 0; n <= max_sym; n++) L.fresh[L.nbyte[N0 + n]] = uint32_t(L.nbits[N0 + n]) << 16;
    for (uint32_t s = 0; s <= max_sym; s++) {
        const uint32_t b = L.fresh[s] >> 16;
        if (b) { const uint32_t v = L.val[b]; L.val[b] = uint16_t(v + 1); L.fresh[s] = (b << 16) | v; }
    }
    return max_bits;
}

// HUF_writeCTable_wksp of L.fresh; bytes or kErr*
__device__ __forceinline__ int huf_write_table(ZLds& L, uint8_t* dst, uint32_t cap, uint32_t max_sym, uint32_t log, int lane)
{
    for (uint32_t s = lane; s < max_sym; s += 64) { const uint32_t b = L.fresh[s] >> 16; L.weight[s] = b ? uint8_t(log + 1 - b) : uint8_t(0); }
    if (cap < 1) return kErrTooSmall;
    int h = 0;
    if (max_sym > 1) {                                        // HUF_compressWeights (:147-186)
        uint8_t* const out = dst + 1;
        const uint32_t ocap = cap - 1;
        if (lane < 16) L.wcount[lane] = 0;
        for (uint32_t s = lane; s < max_sym; s += 64) atomicAdd(&L.wcount[L.weight[s]], 1u);
        uint32_t top = 0, wmax = 0;
        for (uint32_t w = 0; w <= 12; w++) { const uint32_t c = L.wcount[w]; top = max(top, c); if (c) wmax = w; }
        if (top == max_sym) h = 1;
        else if (top == 1) h = 0;
        else {
            const uint32_t wlog = fse_optimal_log(6, max_sym, wmax, 2);
            if (lane < 13) L.count[lane] = L.wcount[lane];            // the literal histogram is restored by the caller
            int r = fse_normalize(L, wlog, max_sym, wmax, false);
            if (r < 0) return r;
            const int nc = fse_write_ncount(L, out, ocap, wmax, wlog, lane);
            if (nc < 0) return nc;
            FseCt& wt = L.ct[0];
            fse_build(L, wt, wmax, wlog);
            // FSE_compress_usingCTable, two interleaved states (:560-620)
            uint32_t c = 0;
            BitW w;
            if (max_sym > 2 && w.init(out + nc, ocap - uint32_t(nc))) {
                uint32_t i = max_sym, s1, s2;
                auto enc = [&](uint32_t st, uint32_t sym) -> uint32_t {
                    const uint32_t nb = (st + wt.dbits[sym]) >> 16;
                    w.put(st, nb, lane);
                    return wt.next[int32_t(st >> nb) + wt.dfind[sym]];
                };
                if (max_sym & 1) { s1 = fse_first_state(wt, L.weight[--i]); s2 = fse_first_state(wt, L.weight[--i]); s1 = enc(s1, L.weight[--i]); }
                else             { s2 = fse_first_state(wt, L.weight[--i]); s1 = fse_first_state(wt, L.weight[--i]); }
                while (i > 0) { s2 = enc(s2, L.weight[--i]); s1 = enc(s1, L.weight[--i]); }
                w.put(s2, wt.log, lane); w.put(s1, wt.log, lane);
                c = w.close(lane);
            }
            h = c ? nc + int(c) : 0;
        }
    }
    if (h > 1 && uint32_t(h) < max_sym / 2) { if (lane == 0) dst[0] = uint8_t(h); return h + 1; }
    if (max_sym > 128) return kErrGeneric;
    if (((max_sym + 1) / 2) + 1 > cap) return kErrTooSmall;
    if (lane == 0) { dst[0] = uint8_t(128 + (max_sym - 1)); L.weight[max_sym] = 0; }
    for (uint32_t s = 2 * lane; s < max_sym; s += 128) dst[s / 2 + 1] = uint8_t((L.weight[s] << 4) + (s + 1 < max_sym ? L.weight[s + 1] : 0));
    return int((max_sym + 1) / 2) + 1;
}

// one Huffman stream (HUF_compress1X_usingCTable_internal_body, :1029-1094): symbols lit[a, b), last first.
// Wave-parallel: 8 symbols per lane and round, prefix sum of code lengths, LDS atomic-or staging.
__device__ __forceinline__ uint32_t huf_stream(ZLds& L, const uint32_t* tab, uint8_t* dst, uint32_t cap,
                                               const uint8_t* lit, uint32_t a, uint32_t b, int lane)
{
    if (cap <= 8) return 0;
    uint32_t* const stage = L.count;                                // (the histogram has served: every estimate is made before the first stream)
    uint32_t total = 0, pos = 0, carry = 0, carry_nb = 0;
    for (uint32_t hi = b; hi > a; ) {
        const int32_t top = int32_t(hi) - 1 - 8 * lane;
        uint64_t v0 = 0, v1 = 0; uint32_t len = 0;
        if (top >= int32_t(a)) {
            const uint64_t w = ld8(lit + top - 7);                  // the literal buffer has a front pad
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (top - k >= int32_t(a)) {
                    const uint32_t e = tab[(w >> (8 * (7 - k))) & 255];
                    const uint64_t code = e & 0xFFFF; const uint32_t nbt = e >> 16;
                    if (len < 64) { v0 |= code << len; if (len + nbt > 64) v1 |= code >> (64 - len); }
                    else v1 |= code << (len - 64);
                    len += nbt;
                }
            }
        }
        const uint32_t incl = scan_add(len);
        const uint32_t round_bits = rl(incl, 63);
        const uint32_t off = incl - len + carry_nb;
        for (int i = lane; i < 200; i += 64) stage[i] = (i == 0) ? carry : 0u;
        if (len) {
            const uint32_t w0 = off >> 5, sh = off & 31;
            const uint64_t lo = v0 << sh;
            const uint64_t mid = (sh ? (v0 >> (64 - sh)) : 0ull) | (v1 << sh);
            if (uint32_t(lo)) atomicOr(&stage[w0], uint32_t(lo));
            if (uint32_t(lo >> 32)) atomicOr(&stage[w0 + 1], uint32_t(lo >> 32));
            if (uint32_t(mid)) atomicOr(&stage[w0 + 2], uint32_t(mid));
            if (uint32_t(mid >> 32)) atomicOr(&stage[w0 + 3], uint32_t(mid >> 32));
        }
        const uint32_t bits = carry_nb + round_bits, nbytes = bits >> 3;
        for (uint32_t wi = lane; 4 * wi < nbytes; wi += 64) {
            const uint32_t v = stage[wi];
            if (4 * wi + 4 <= nbytes && pos + 4 * wi + 4 <= cap) st4(dst + pos + 4 * wi, v);
            else for (uint32_t k = 0; k < 4; k++) if (4 * wi + k < nbytes && pos + 4 * wi + k < cap) dst[pos + 4 * wi + k] = uint8_t(v >> (8 * k));
        }
        carry_nb = bits & 7;
        carry = (stage[nbytes >> 2] >> (8 * (nbytes & 3))) & ((1u << carry_nb) - 1);
        pos += nbytes; total += round_bits;
        hi = hi - a > 512 ? hi - 512 : a;
    }
    carry |= 1u << carry_nb; carry_nb++; total++;                 // HUF_endMark
    if (lane == 0 && pos < cap) dst[pos] = uint8_t(carry);
    return ((total >> 3) + 8 < cap) ? (total + 7) >> 3 : 0u;
}

// HUF_compressCTable_internal (:1208-1224); `hdr` table-description bytes already sit in front of dst
__device__ __forceinline__ uint32_t huf_encode(ZLds& L, const uint32_t* tab, uint8_t* dst, uint32_t cap, uint32_t hdr,
                                               const uint8_t* lit, uint32_t n, bool four, int lane)
{
    uint32_t c;
    if (!four) c = huf_stream(L, tab, dst, cap, lit, 0, n, lane);
    else {
        const uint32_t seg = (n + 3) / 4;
        uint32_t o = 6;
        if (cap < 6 + 1 + 1 + 1 + 8 || n < 12) return 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t a = k * seg, b = k < 3 ? a + seg : n;
            const uint32_t s = huf_stream(L, tab, dst + o, cap - o, lit, a, b, lane);
            if (s == 0 || s > 65535) return 0;
            if (k < 3 && lane == 0) { dst[2 * k] = uint8_t(s); dst[2 * k + 1] = uint8_t(s >> 8); }
            o += s;
        }
        c = o;
    }
    if (!c) return 0;
    if (hdr + c >= n - 1) return 0;
    return hdr + c;
}

enum { kRepNone = 0, kRepCheck = 1, kRepValid = 2 };

// HUF_compress_internal (:1250-1360).  `table` = L.huf[next] holds the previous block's table on entry.
__device__ __forceinline__ int huf_compress(ZLds& L, uint32_t* table, uint8_t* dst, uint32_t cap, const uint8_t* lit, uint32_t n,
                                            bool four, int& repeat, bool prefer_repeat, bool suspect, int lane)
{
    uint32_t largest, max_sym;
    if (!n || !cap) return 0;
    if (prefer_repeat && repeat == kRepValid) return int(huf_encode(L, table, dst, cap, 0, lit, n, four, lane));
    if (suspect && n >= 4096 * 10) {
        uint32_t a, b, m;
        hist_bytes(L, lit, 4096, a, m, lane);
        hist_bytes(L, lit + n - 4096, 4096, b, m, lane);
        if (a + b <= ((2 * 4096) >> 7) + 4) return 0;
    }
    hist_bytes(L, lit, n, largest, max_sym, lane);
    if (largest == n) { if (lane == 0) dst[0] = lit[0]; return 1; }
    if (largest <= (n >> 7) + 4) return 0;
    if (repeat == kRepCheck) {
        bool bad = false;
        for (uint32_t s = lane; s <= max_sym; s += 64) bad |= (L.count[s] != 0) && ((table[s] >> 16) == 0);
        if (__ballot(bad)) repeat = kRepNone;
    }
    if (prefer_repeat && repeat != kRepNone) return int(huf_encode(L, table, dst, cap, 0, lit, n, four, lane));
    uint32_t log = fse_optimal_log(11, n, max_sym, 1);
    log = huf_build(L, max_sym, log, lane);
    // bit estimates before the weight coder borrows L.count
    uint32_t old_bits = 0, new_bits = 0;
    for (uint32_t s = lane; s <= max_sym; s += 64) { old_bits += (table[s] >> 16) * L.count[s]; new_bits += (L.fresh[s] >> 16) * L.count[s]; }
    old_bits = rl(scan_add(old_bits), 63); new_bits = rl(scan_add(new_bits), 63);
    const int h = huf_write_table(L, dst, cap, max_sym, log, lane);
    if (h < 0) return h;
    if (repeat != kRepNone) {
        if ((old_bits >> 3) <= uint32_t(h) + (new_bits >> 3) || uint32_t(h) + 12 >= n)
            return int(huf_encode(L, table, dst, cap, 0, lit, n, four, lane));
    }
    if (uint32_t(h) + 12 >= n) return 0;
    repeat = kRepNone;
    for (int i = lane; i < 256; i += 64) table[i] = L.fresh[i];
    return int(huf_encode(L, L.fresh, dst + h, cap - uint32_t(h), uint32_t(h), lit, n, four, lane));
}

// ZSTD_noCompressLiterals / ZSTD_compressRleLiteralsBlock
__device__ __forceinline__ int raw_literals(uint8_t* dst, uint32_t cap, const uint8_t* lit, uint32_t n, bool rle, int lane)
{
    const uint32_t fl = 1 + (n > 31) + (n > 4095);
    const uint32_t type = rle ? 1u : 0u;
    if (!rle && n + fl > cap) return kErrTooSmall;
    const uint32_t v = fl == 1 ? type + (n << 3) : fl == 2 ? type + (1u << 2) + (n << 4) : type + (3u << 2) + (n << 4);
    if (uint32_t(lane) < fl) dst[lane] = uint8_t(v >> (8 * lane));
    if (rle) { if (lane == 0) dst[fl] = lit[0]; return int(fl) + 1; }
    copy_bytes(dst + fl, lit, n, lane);
    return int(n + fl);
}

struct Entropy { int huf_repeat; uint32_t rep[3]; int fse_repeat[3]; };       // per-block state besides the LDS literal table and the FSE tables

// ZSTD_compressLiterals
__device__ __forceinline__ int compress_literals(ZLds& L, int prev, const Entropy& pe, Entropy& ne, uint8_t* dst, uint32_t cap,
                                                 const uint8_t* lit, uint32_t n, bool suspect, uint32_t strat, int lane)
{
    const uint32_t min_gain = (n >> 6) + 2, lh = 3 + (n >= 1024) + (n >= 16384);
    bool single = n < 256;
    int repeat = pe.huf_repeat, type = 2;
    uint32_t* const ptab = L.huf[prev];
    uint32_t* const ntab = L.huf[prev ^ 1];
    for (int i = lane; i < 256; i += 64) ntab[i] = ptab[i];
    ne.huf_repeat = pe.huf_repeat;
    if (n <= uint32_t(pe.huf_repeat == kRepValid ? 6 : 63)) return raw_literals(dst, cap, lit, n, false, lane);
    if (cap < lh + 1) return kErrTooSmall;
    if (repeat == kRepValid && lh == 3) single = true;
    const int c = huf_compress(L, ntab, dst + lh, cap - lh, lit, n, !single, repeat, strat < 4 && n <= 1024, suspect, lane);
    if (repeat != kRepNone) type = 3;
    if (c <= 0 || uint32_t(c) >= n - min_gain) { for (int i = lane; i < 256; i += 64) ntab[i] = ptab[i]; return raw_literals(dst, cap, lit, n, false, lane); }
    if (c == 1) { for (int i = lane; i < 256; i += 64) ntab[i] = ptab[i]; return raw_literals(dst, cap, lit, n, true, lane); }
    if (type == 2) ne.huf_repeat = kRepCheck;
    uint64_t v;
    if (lh == 3) v = uint32_t(type) + (uint32_t(!single) << 2) + (n << 4) + (uint32_t(c) << 14);
    else if (lh == 4) v = uint32_t(type) + (2u << 2) + (n << 4) + (uint32_t(c) << 18);
    else v = uint64_t(uint32_t(uint32_t(type) + (3u << 2) + (n << 4) + (uint32_t(c) << 22))) | (uint64_t(uint32_t(c) >> 10) << 32);
    if (uint32_t(lane) < lh) dst[lane] = uint8_t(v >> (8 * lane));
    return int(lh) + c;
}

// ------------------------------------------------------------------------------------------------ sequences
__device__ __forceinline__ uint32_t ll_code(uint32_t v)
{
    if (v > 63) return uint32_t(hibit(v)) + 19;
    if (v < 16) return v;
    if (v < 24) return 16 + ((v - 16) >> 1);
    if (v < 32) return 20 + ((v - 24) >> 2);
    if (v < 48) return 22 + ((v - 32) >> 3);
    return 24;
}
__device__ __forceinline__ uint32_t ml_code(uint32_t v)
{
    if (v > 127) return uint32_t(hibit(v)) + 36;
    if (v < 32) return v;
    if (v < 40) return 32 + ((v - 32) >> 1);
    if (v < 48) return 36 + ((v - 40) >> 2);
    if (v < 64) return 38 + ((v - 48) >> 3);
    if (v < 96) return 40 + ((v - 64) >> 4);
    return 42;
}
__device__ __forceinline__ uint32_t ll_bits(uint32_t c) { return c < 16 ? 0u : c < 26 ? uint32_t((0x6433221111ull >> (4 * (c - 16))) & 15) : c - 19; }
__device__ __forceinline__ uint32_t ml_bits(uint32_t c) { return c < 32 ? 0u : c < 43 ? uint32_t((0x54433221111ull >> (4 * (c - 32))) & 15) : c - 36; }

__device__ __forceinline__ int default_norm(int which, uint32_t s)       // LL_defaultNorm / OF_ / ML_ (zstd_internal.h)
{
    if (which == 0) return s == 0 ? 4 : (s == 1 || s == 25) ? 3 : (s >= 13 && s <= 15) ? 1 : (s >= 27 && s <= 31) ? 1 : s >= 32 ? -1 : 2;
    if (which == 1) return (s >= 6 && s <= 8) ? 2 : s >= 24 ? -1 : 1;
    return s == 0 ? 1 : s == 1 ? 4 : s == 2 ? 3 : (s >= 3 && s <= 8) ? 2 : s >= 46 ? -1 : 1;
}

// floor(-log2(i / 256) * 256), i in 1..255 (kInverseProbabilityLog256, zstd_compress_sequences.c:19-42), computed at
// compile time: exponent + 24 fractional bits of log2 by repeated squaring
struct InvLog { uint16_t v[256]; };
constexpr InvLog make_inv_log()
{
    InvLog t{};
    for (unsigned i = 1; i < 256; i++) {
        unsigned long long x = (unsigned long long)i << 56; unsigned ip = 0, r = 0;
        while (!(x >> 63)) { x <<= 1; ip++; }
        for (int k = 0; k < 24; k++) {
            const __uint128_t sq = (__uint128_t)x * x;
            r <<= 1;
            if ((sq >> 127) & 1) { r |= 1; x = (unsigned long long)(sq >> 64); } else x = (unsigned long long)(sq >> 63);
        }
        t.v[i] = uint16_t((((unsigned long long)(ip + 1) << 24) - r) >> 16);
    }
    return t;
}
__constant__ InvLog kInvLog = make_inv_log();

constexpr uint32_t kCostErr = 0xFFFFFFFFu;

// ZSTD_fseBitCost of the previous block's table (kept in the HBM workspace) for the histogram in L.count
__device__ __forceinline__ uint32_t fse_bit_cost(ZLds& L, const FseCt* prev, uint32_t max, int lane)
{
    if (prev->maxsym < max) return kCostErr;
    const uint32_t log = prev->log, badc = (log + 1) << 8;
    uint32_t cost = 0; bool bad = false;
    for (uint32_t sy = lane; sy <= max; sy += 64) {
        const uint32_t d = prev->dbits[sy], c = L.count[sy];
        const uint32_t min_bits = d >> 16, threshold = (min_bits + 1) << 16;
        const uint32_t delta = threshold - (d + (1u << log));
        const uint32_t bits = (min_bits + 1) * 256 - ((delta << 8) >> log);
        if (c) { if (bits >= badc) bad = true; cost += c * bits; }
    }
    if (__ballot(bad)) return kCostErr;
    cost = rl(scan_add(cost), 63);
    return cost >> 8;
}

// ZSTD_selectEncodingType (zstd_compress_sequences.c:153-239), no dictionary: 0 basic, 1 rle, 2 compressed, 3 repeat
__device__ __forceinline__ int select_type(ZLds& L, int& repeat, int which, uint32_t most, uint32_t max, uint32_t nseq, uint32_t fse_log,
                                           const FseCt* prev, uint8_t* tmp, uint32_t def_log, bool def_ok, uint32_t strat, int lane)
{
    if (most == nseq) { repeat = kRepNone; return (def_ok && nseq <= 2) ? 0 : 1; }
    if (strat < 4) {
        if (def_ok) {
            const uint32_t dyn_min = ((1u << def_log) * (10u - strat)) >> 3;      // ZSTD_fast = 1, ZSTD_dfast = 2
            if (repeat == kRepValid && nseq < 1000) return 3;
            if (nseq < dyn_min || most < (nseq >> (def_log - 1))) { repeat = kRepNone; return 0; }
        }
    } else {
        uint32_t basic = kCostErr, rep = kCostErr, comp = 0;
        if (def_ok) {                                              // ZSTD_crossEntropyCost
            uint32_t c = 0;
            for (uint32_t sy = lane; sy <= max; sy += 64) {
                const int dn = default_norm(which, sy);
                c += L.count[sy] * kInvLog.v[uint32_t(dn != -1 ? dn : 1) << (8 - def_log)];
            }
            c = rl(scan_add(c), 63);
            basic = c >> 8;
        }
        if (repeat != kRepNone) rep = fse_bit_cost(L, prev, max, lane);
        {                                                          // ZSTD_NCountCost + ZSTD_entropyCost
            const uint32_t log = fse_optimal_log(fse_log, nseq, max, 2);
            fse_normalize(L, log, nseq, max, nseq >= 2048);
            const int nc = fse_write_ncount(L, tmp, 512, max, log, lane);
            uint32_t c = 0;
            for (uint32_t sy = lane; sy <= max; sy += 64) {
                const uint32_t cnt = L.count[sy];
                uint32_t q = uint32_t((256ull * cnt) / nseq);
                if (cnt && !q) q = 1;
                c += cnt * kInvLog.v[q];
            }
            c = rl(scan_add(c), 63);
            comp = (uint32_t(nc) << 3) + (c >> 8);
        }
        if (basic <= rep && basic <= comp) { repeat = kRepNone; return 0; }
        if (rep <= comp) return 3;
    }
    repeat = kRepCheck;
    return 2;
}

// ZSTD_buildCTable for table `which` (0 ll, 1 of, 2 ml): bytes of description or kErr*
__device__ __forceinline__ int build_seq_table(ZLds& L, int which, uint8_t* dst, uint32_t cap, uint32_t fse_log, int type, uint32_t max,
                                               const uint8_t* codes, uint32_t nseq, uint32_t def_log, uint32_t def_max, const FseCt* prev, int lane)
{
    FseCt& ct = L.ct[which];
    if (type == 3) {                                               // set_repeat: the previous block's table
        const uint32_t* from = reinterpret_cast<const uint32_t*>(prev);
        uint32_t* to = reinterpret_cast<uint32_t*>(&ct);
        for (uint32_t i = lane; i < sizeof(FseCt) / 4; i += 64) to[i] = from[i];
        return 0;
    }
    if (type == 1) {
        if (lane == 0) { ct.log = 0; ct.maxsym = max; ct.next[0] = 0; ct.next[1] = 0; ct.dbits[max] = 0; ct.dfind[max] = 0; }
        if (!cap) return kErrTooSmall;
        if (lane == 0) dst[0] = codes[0];
        return 1;
    }
    if (type == 0) {
        if (uint32_t(lane) <= def_max) L.norm[lane] = int16_t(default_norm(which, uint32_t(lane)));
        fse_build(L, ct, def_max, def_log);
        return 0;
    }
    uint32_t n1 = nseq;
    const uint32_t log = fse_optimal_log(fse_log, nseq, max, 2);
    const uint32_t lastc = codes[nseq - 1];
    if (L.count[lastc] > 1) { L.count[lastc] = L.count[lastc] - 1; n1--; }
    int r = fse_normalize(L, log, n1, max, n1 >= 2048);
    if (r < 0) return r;
    r = fse_write_ncount(L, dst, cap, max, log, lane);
    if (r < 0) return r;
    fse_build(L, ct, max, log);
    return r;
}

struct SeqStore { uint32_t *ll, *ml, *off; uint8_t *llc, *ofc, *mlc; uint8_t* lit; uint32_t nseq, nlit;
#ifdef Z1_PROF
    uint64_t prof[20];      // 0..10 event counts, 11..15 cycles per phase, 19 last time stamp
#endif
};

// sequences section (tail of ZSTD_entropyCompressSeqStore_internal); bytes, 0 or kErr*
__device__ __forceinline__ int encode_sequences(ZLds& L, uint8_t* dst, uint32_t cap, const SeqStore& S, uint32_t strat,
                                                const Entropy& pe, Entropy& ne, const FseCt* prevfse, uint8_t* tmp, bool& tables_built, int lane)
{
    tables_built = false;
    ne.fse_repeat[0] = pe.fse_repeat[0]; ne.fse_repeat[1] = pe.fse_repeat[1]; ne.fse_repeat[2] = pe.fse_repeat[2];
    const uint32_t nseq = S.nseq;
    uint32_t o = 0, last_count = 0;
    if (cap < 4) return kErrTooSmall;
    if (nseq < 128) { if (lane == 0) dst[0] = uint8_t(nseq); o = 1; }
    else if (nseq < 0x7F00) { if (lane == 0) { dst[0] = uint8_t((nseq >> 8) + 0x80); dst[1] = uint8_t(nseq); } o = 2; }
    else { if (lane == 0) { dst[0] = 0xFF; dst[1] = uint8_t(nseq - 0x7F00); dst[2] = uint8_t((nseq - 0x7F00) >> 8); } o = 3; }
    if (!nseq) return int(o);
    for (uint32_t i = lane; i < nseq; i += 64) {
        S.llc[i] = uint8_t(ll_code(S.ll[i])); S.ofc[i] = uint8_t(hibit(S.off[i])); S.mlc[i] = uint8_t(ml_code(S.ml[i]));
    }
    {
        uint8_t* const head = dst + o++;
        uint32_t most, max;
        hist_bytes(L, S.llc, nseq, most, max, lane);
        const int tll = select_type(L, ne.fse_repeat[0], 0, most, max, nseq, 9, prevfse + 0, tmp, 6, true, strat, lane);
        int r = build_seq_table(L, 0, dst + o, cap - o, 9, tll, max, S.llc, nseq, 6, 35, prevfse + 0, lane);
        if (r < 0) return r;
        if (tll == 2) last_count = uint32_t(r);
        o += uint32_t(r);
        hist_bytes(L, S.ofc, nseq, most, max, lane);
        const int tof = select_type(L, ne.fse_repeat[1], 1, most, max, nseq, 8, prevfse + 1, tmp, 5, max <= 28, strat, lane);
        r = build_seq_table(L, 1, dst + o, cap - o, 8, tof, max, S.ofc, nseq, 5, 28, prevfse + 1, lane);
        if (r < 0) return r;
        if (tof == 2) last_count = uint32_t(r);
        o += uint32_t(r);
        hist_bytes(L, S.mlc, nseq, most, max, lane);
        const int tml = select_type(L, ne.fse_repeat[2], 2, most, max, nseq, 9, prevfse + 2, tmp, 6, true, strat, lane);
        r = build_seq_table(L, 2, dst + o, cap - o, 9, tml, max, S.mlc, nseq, 6, 52, prevfse + 2, lane);
        tables_built = true;
        if (r < 0) return r;
        if (tml == 2) last_count = uint32_t(r);
        o += uint32_t(r);
        if (lane == 0) *head = uint8_t((tll << 6) + (tof << 4) + (tml << 2));
    }
    {   // ZSTD_encodeSequences_body (zstd_compress_sequences.c:302-400): last sequence first, 64 sequences per chunk, lane l
        // holds sequence top - l.  What is serial are the three FSE state chains (a table look-up per sequence and stream):
        // they run on lanes 0..2 (of, ml, ll - the order their bits are written in), fed from and answering into LDS.
        // Putting the bits together (state bits, then the extra bits of ll, ml, of), their positions (prefix sum) and
        // the stream itself (LDS atomic-or staging, as the Huffman streams do) are wave-parallel.
        if (cap - o <= 8) return kErrTooSmall;                           // BIT_initCStream
        uint8_t* const out = dst + o; const uint32_t ocap = cap - o;
        struct SeqStage { uint32_t d[3][64]; int32_t f[3][64]; uint32_t o[3][64]; };                 // 2304 B over the Huffman builder's scratch (idle here)
        static_assert(sizeof(SeqStage) <= sizeof(L.ncount) + sizeof(L.nparent), "stage fits");
        SeqStage& st = *reinterpret_cast<SeqStage*>(&L.ncount[0]);
        uint32_t* const stage = L.count;                                 // (the three tables are built: the histogram has served)
        const FseCt& cll = L.ct[0]; const FseCt& cof = L.ct[1]; const FseCt& cml = L.ct[2];
        const FseCt* const myct = &L.ct[lane == 0 ? 1 : lane == 1 ? 2 : 0];                         // lane 0: of, 1: ml, 2: ll
        const int me = lane < 3 ? lane : 0;
        uint32_t sst = 0;                                                // lanes 0..2: the state of their stream
        uint32_t total = 0, pos = 0, carry = 0, carry_nb = 0;
        bool first = true;
        for (int32_t top = int32_t(nseq) - 1; top >= 0; top -= 64) {
            const int32_t i = top - lane;
            uint32_t cl = 0, co = 0, cm = 0, x0 = 0, x1 = 0, n0 = 0, n1 = 0;
            if (i >= 0) {
                cl = S.llc[i]; co = S.ofc[i]; cm = S.mlc[i];
                const uint32_t bl = ll_bits(cl), bm = ml_bits(cm);
                const uint64_t ext = uint64_t(S.ll[i] & ((1u << bl) - 1)) | (uint64_t(S.ml[i] & ((1u << bm) - 1)) << bl);
                x0 = uint32_t(ext); n0 = bl + bm;                      // <= 32 bits
                x1 = S.off[i] & ((1u << co) - 1); n1 = co;              // <= 31 bits
                st.d[0][lane] = cof.dbits[co]; st.f[0][lane] = cof.dfind[co];
                st.d[1][lane] = cml.dbits[cm]; st.f[1][lane] = cml.dfind[cm];
                st.d[2][lane] = cll.dbits[cl]; st.f[2][lane] = cll.dfind[cl];
            }
            const int cnt = top >= 63 ? 64 : top + 1;
            int k0 = 0;
            if (first) {                                                 // FSE_initCState2 with the last sequence's symbols: no bits
                const uint32_t sym = lane == 0 ? rl(co, 0) : lane == 1 ? rl(cm, 0) : rl(cl, 0);
                if (lane < 3) { sst = fse_first_state(*myct, sym); st.o[me][0] = 0; }
                first = false; k0 = 1;
            }
            if (lane < 3) {
                uint32_t dn = st.d[me][k0 & 63]; int32_t fn = st.f[me][k0 & 63];
                for (int k = k0; k < cnt; k++) {
                    const uint32_t d = dn; const int32_t f = fn;
                    dn = st.d[me][(k + 1) & 63]; fn = st.f[me][(k + 1) & 63];                        // (next step's, off the state chain)
                    const uint32_t nb = (sst + d) >> 16;
                    st.o[me][k] = (sst & ((1u << nb) - 1)) | (nb << 16);
                    sst = myct->next[int32_t(sst >> nb) + f];
                }
            }
            // this lane's sequence: state bits of, ml, ll; extra bits ll | ml; extra bits of
            uint64_t v0 = 0, v1 = 0; uint32_t len = 0;
            if (i >= 0) {
                const uint32_t o0 = st.o[0][lane], o1 = st.o[1][lane], o2 = st.o[2][lane];
                const uint32_t b0 = o0 >> 16, b1 = o1 >> 16, b2 = o2 >> 16;
                const uint32_t nbf = b0 + b1 + b2;                                                   // <= 8 + 9 + 9
                v0 = uint64_t((o0 & 0xFFFFu) | ((o1 & 0xFFFFu) << b0) | ((o2 & 0xFFFFu) << (b0 + b1))) | (uint64_t(x0) << nbf);
                const uint32_t p = nbf + n0;                                                         // <= 58
                v0 |= uint64_t(x1) << p;
                if (p + n1 > 64) v1 = uint64_t(x1) >> (64 - p);
                len = p + n1;
            }
            const uint32_t incl = scan_add(len);
            const uint32_t round_bits = rl(incl, 63);
            const uint32_t off = incl - len + carry_nb;
            for (int w = lane; w < 200; w += 64) stage[w] = (w == 0) ? carry : 0u;
            if (len) {
                const uint32_t w0 = off >> 5, sh = off & 31;
                const uint64_t lo = v0 << sh;
                const uint64_t mid = (sh ? (v0 >> (64 - sh)) : 0ull) | (v1 << sh);
                if (uint32_t(lo)) atomicOr(&stage[w0], uint32_t(lo));
                if (uint32_t(lo >> 32)) atomicOr(&stage[w0 + 1], uint32_t(lo >> 32));
                if (uint32_t(mid)) atomicOr(&stage[w0 + 2], uint32_t(mid));
                if (uint32_t(mid >> 32)) atomicOr(&stage[w0 + 3], uint32_t(mid >> 32));
            }
            const uint32_t bits = carry_nb + round_bits, nbytes = bits >> 3;
            for (uint32_t wi = lane; 4 * wi < nbytes; wi += 64) {
                const uint32_t v = stage[wi];
                if (4 * wi + 4 <= nbytes && pos + 4 * wi + 4 <= ocap) st4(out + pos + 4 * wi, v);
                else for (uint32_t k = 0; k < 4; k++) if (4 * wi + k < nbytes && pos + 4 * wi + k < ocap) out[pos + 4 * wi + k] = uint8_t(v >> (8 * k));
            }
            carry_nb = bits & 7;
            carry = (stage[nbytes >> 2] >> (8 * (nbytes & 3))) & ((1u << carry_nb) - 1);
            pos += nbytes; total += round_bits;
        }
        {   // FSE_flushCState of ml, of, ll, then the end mark (BIT_closeCStream)
            const uint32_t sof = rl(sst, 0), sml = rl(sst, 1), sll = rl(sst, 2);
            const uint32_t lm = cml.log, lo = cof.log, lg = cll.log;
            const uint32_t tb = lm + lo + lg + 1;                                                    // <= 28
            const uint64_t t = uint64_t((sml & ((1u << lm) - 1)) | ((sof & ((1u << lo) - 1)) << lm) | ((sll & ((1u << lg) - 1)) << (lm + lo)) | (1u << (lm + lo + lg)));
            const uint64_t v = uint64_t(carry) | (t << carry_nb);
            const uint32_t nb = carry_nb + tb, nbytes = (nb + 7) >> 3;
            if (uint32_t(lane) < nbytes && pos + lane < ocap) out[pos + lane] = uint8_t(v >> (8 * lane));
            total += tb;
        }
        const uint32_t bytes = ((total >> 3) + 8 < ocap) ? (total + 7) >> 3 : 0u;
        if (!bytes) return kErrTooSmall;
        o += bytes;
        if (last_count && last_count + bytes < 4) return 0;             // zstd <= 1.3.4 decoder workaround
    }
    return int(o);
}

// ------------------------------------------------------------------------------------------------ match finder
struct Params { uint32_t wlog, hlog, clog, slog, mml, tlen, strat; };

__device__ __forceinline__ uint32_t zhash(uint64_t v, uint32_t hlog, uint32_t mls)
{
    if (mls == 7) return uint32_t(((v << 8) * 58295818150454627ull) >> (64 - hlog));
    if (mls == 6) return uint32_t(((v << 16) * 227718039650203ull) >> (64 - hlog));
    if (mls == 5) return uint32_t(((v << 24) * 889523592379ull) >> (64 - hlog));
    return (uint32_t(v) * 2654435761u) >> (32 - hlog);
}

// bytes equal from s[a..] vs s[b..] (a > b), a stops at lim; wave-wide
__device__ __forceinline__ uint32_t count_fwd(const uint8_t* s, uint32_t a, uint32_t b, uint32_t lim, int lane)
{
    uint32_t n = 0;
    for (;;) {
        if (a + 1024 <= lim) {
            const U16B x = *reinterpret_cast<const U16B*>(s + a + 16 * lane);
            const U16B y = *reinterpret_cast<const U16B*>(s + b + 16 * lane);
            const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
            const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
            const unsigned long long bad = __ballot(eq < 16);
            if (bad) { const int l = __builtin_ctzll(bad); return n + 16 * l + rl(eq, l); }
            n += 1024; a += 1024; b += 1024;
        } else {
            const uint32_t i = a + lane;
            const bool same = (i < lim) && s[i] == s[b + lane];
            const unsigned long long bad = ~__ballot(same);
            if (bad) return n + __builtin_ctzll(bad);
            n += 64; a += 64; b += 64;
        }
    }
}

// ZSTD_storeSeq (zstd_compress_internal.h:643-686).  Only the lengths are recorded while parsing: the literal bytes are
// gathered once per block (gather_literals), off the parser's chain of dependent memory round trips.
__device__ __forceinline__ void store_seq(SeqStore& S, const uint8_t* s, uint32_t anchor, uint32_t ll, uint32_t off_base, uint32_t ml, int lane)
{
    (void)s; (void)anchor;
    S.nlit += ll;
    if (lane == 0) { S.ll[S.nseq] = ll; S.ml[S.nseq] = ml - 3; S.off[S.nseq] = off_base; }
    S.nseq++;
}

struct __attribute__((packed, aligned(1))) LP8 { uint64_t v; };
struct __attribute__((packed, aligned(1))) LP4 { uint32_t v; };
struct __attribute__((packed, aligned(1))) LP2 { uint16_t v; };
__device__ __forceinline__ void lane_copy32(uint8_t* b, const uint8_t* a, uint32_t n)   // exact length, n <= 32, one lane
{
    if (n & 32) { const U16B x = *reinterpret_cast<const U16B*>(a), y = *reinterpret_cast<const U16B*>(a + 16);
                  *reinterpret_cast<U16B*>(b) = x; *reinterpret_cast<U16B*>(b + 16) = y; return; }
    if (n & 16) { const U16B x = *reinterpret_cast<const U16B*>(a); *reinterpret_cast<U16B*>(b) = x; a += 16; b += 16; }
    if (n & 8) { const LP8 x = *reinterpret_cast<const LP8*>(a); *reinterpret_cast<LP8*>(b) = x; a += 8; b += 8; }
    if (n & 4) { const LP4 x = *reinterpret_cast<const LP4*>(a); *reinterpret_cast<LP4*>(b) = x; a += 4; b += 4; }
    if (n & 2) { const LP2 x = *reinterpret_cast<const LP2*>(a); *reinterpret_cast<LP2*>(b) = x; a += 2; b += 2; }
    if (n & 1) *b = *a;
}

// the literal runs of the block's sequences, in order, into S.lit: 64 sequences per step, positions by prefix sums of
// the recorded lengths; short runs lane by lane, long ones with the whole wave
__device__ __forceinline__ void gather_literals(const SeqStore& S, const uint8_t* s, uint32_t start, int lane)
{
    uint32_t srcpos = start, dstpos = 0;
    for (uint32_t base = 0; base < S.nseq; base += 64) {
        const uint32_t i = base + lane;
        const bool on = i < S.nseq;
        const uint32_t ll = on ? S.ll[i] : 0u, tot = on ? ll + S.ml[i] + 3 : 0u;
        const uint32_t il = scan_add(ll), it = scan_add(tot);
        const uint32_t my_src = srcpos + it - tot, my_dst = dstpos + il - ll;
        if (ll && ll <= 32) lane_copy32(S.lit + my_dst, s + my_src, ll);
        unsigned long long big = __ballot(ll > 32);
        while (big) {
            const int l = __builtin_ctzll(big);
            copy_bytes(S.lit + rl(my_dst, l), s + rl(my_src, l), rl(ll, l), lane);
            big &= big - 1;
        }
        srcpos += rl(it, 63); dstpos += rl(il, 63);
    }
}

// 16 bytes at p, any alignment, as four dwords (one global_load_dwordx4)
struct Q16 { uint32_t d0, d1, d2, d3; };
__device__ __forceinline__ Q16 ld16(const uint8_t* p) { const U16B t = *reinterpret_cast<const U16B*>(p); return Q16{uint32_t(t.a), uint32_t(t.a >> 32), uint32_t(t.b), uint32_t(t.b >> 32)}; }
__device__ __forceinline__ uint64_t u64(uint32_t lo, uint32_t hi) { return (uint64_t(hi) << 32) | lo; }
// hash table read of the "fast" strategy.  The dense window commits its table writes with atomic max (performed in L2),
// so every read of that table goes to L2 as well (a plain load could hit a stale line of the CU's L1).
constexpr uint32_t kFwHeld = 40;                                                // bytes behind pos + 4 that a dense-window lane holds of itself and of its candidate
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l) { return u64(rl(uint32_t(v), l), rl(uint32_t(v >> 32), l)); }
__device__ __forceinline__ uint32_t tld(const uint32_t* t, uint32_t h) { return __hip_atomic_load(t + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef Z1_PROF   // one-off counters of the dense window (tools/zstd_timing.py reads them behind the phase cycle counters)
#define ZCNT(i) do { S.prof[i]++; } while (0)
#define ZADD(i, v) do { S.prof[i] += (v); } while (0)
#define ZPT(i) do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); const uint64_t zt_ = __builtin_readcyclecounter(); S.prof[i] += zt_ - S.prof[19]; S.prof[19] = zt_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ZPT(i) do { } while (0)
#define ZCNT(i) do { } while (0)
#define ZADD(i, v) do { } while (0)
#endif

// after a match ending at ip0: table refills and the repcode-2 loop (zstd_fast.c:263-281).  `fill_cur` = false when the
// refill of current0 + 2 has been made already (dense window: it was one of the window's lanes).
__device__ __forceinline__ void after_match(SeqStore& S, uint32_t* tab, const Params& P, const uint8_t* s, uint32_t& ip0, uint32_t& anchor,
                                            uint32_t cur0_idx, uint32_t& rep1, uint32_t& rep2, uint32_t end, int64_t ilimit, int lane, bool fill_cur = true, bool fill_end = true)
{
    if (int64_t(ip0) > ilimit) return;
    // the four reads of this step in one round trip, ahead of the table stores
    const uint64_t w_a = fill_cur ? ld8(s + cur0_idx) : 0, w_b = ld8(s + ip0 - 2);
    uint32_t r_cur = ld4(s + ip0), r_rep = ld4(s + ip0 - rep2);         // (rep2 == 0 reads ip0 itself: not used then)
    {
        const uint32_t h_a = zhash(w_a, P.hlog, P.mml), h_b = zhash(w_b, P.hlog, P.mml);
        if (lane == 0) { if (fill_cur) tab[h_a] = cur0_idx + 2; if (fill_end) tab[h_b] = ip0; }
    }
    if (rep2 > 0)
        for (bool first = true; int64_t(ip0) <= ilimit; first = false) {
            if (!first) { r_cur = ld4(s + ip0); r_rep = ld4(s + ip0 - rep2); }
            if (r_cur != r_rep) break;
            const uint32_t rlen = count_fwd(s, ip0 + 4, ip0 + 4 - rep2, end, lane) + 4;
            const uint32_t t = rep2; rep2 = rep1; rep1 = t;
            const uint32_t h = zhash(ld8(s + ip0), P.hlog, P.mml);
            if (lane == 0) tab[h] = ip0 + 2;
            ip0 += rlen;
            store_seq(S, s, anchor, 0, 1, rlen, lane);
            anchor = ip0;
        }
}

// ZSTD_compressBlock_fast_noDict_generic over s[start, end).
//
// Two shapes reproduce the reference's serial walk (pairs of positions ip0, ip0+1 with a repcode test at ip0+step; every tested
// position reads, then overwrites its hash slot; after a match: two refills and the repcode-2 loop):
//   * dense window (step 2, i.e. the first 124 bytes of a search - where nearly all sequences of compressible data are found).
//     Lane l takes position sp+l whatever role the walk will give it and prepares, against the table as it stands: its
//     candidate, the 4-byte test, up to 4 equal bytes backwards and 24 (56) forwards; and, for both repeat offsets, whether its 4
//     bytes repeat at that distance (ballots: the repcode tests of all 64 positions at once, and per-byte equality, whose runs
//     are the lengths of repcode matches).  Lanes sharing a table slot are found with an LDS scoreboard and made exact by a
//     ballot per group.  A scalar walk then only CHOOSES: per sequence it orders the first hash hit against the first repcode
//     hit of the right parity, takes lengths from the prepared lanes, and marks which lanes the reference would have written
//     into the table.  A hash match changes the repeat offset: the repcode ballots are redone for the lanes behind it (one
//     read).  All sequences of the window are stored at once, the marked lanes enter the table with one atomic max each (the
//     latest position of a slot wins).  Whatever the window does not hold (long matches, long catch-up, a lane whose slot an
//     earlier lane of the window also wrote and that could match there, an immediate repcode-2 match) is left to
//   * the batched search: lane j speculatively executes pair j of the running search (16..64 pairs per batch, any step),
//     one event (sequence) per batch; wave-wide extension; the serial tail of the reference (`after_match`).
__device__ __forceinline__ uint32_t fast_block(ZLds& L, SeqStore& S, uint32_t* tab, const Params& P, uint32_t rep[3],
                                               const uint8_t* s, uint32_t start, uint32_t end, uint32_t n_total, bool serial, int lane)
{
    start = U(start); end = U(end);                                            // (wave-uniform: keep what derives from them on the scalar unit)
    const uint32_t hlog = U(P.hlog), wsize = 1u << U(P.wlog), mls = U(P.mml);
    const uint32_t step0 = U(P.tlen > 1 ? P.tlen + 1 : 2);
    const uint32_t prefix_idx = end > wsize ? end + 2 - wsize : 2;
    const uint32_t prefix = prefix_idx - 2;
    const int64_t ilimit = int64_t(end) - 8;
    uint32_t anchor = start, ip0 = start;
    uint32_t rep1 = U(rep[0]), rep2 = U(rep[1]), saved1 = 0, saved2 = 0;

    ip0 += (ip0 == prefix) ? 1 : 0;
    {
        const uint32_t cur = ip0 + 2;
        const uint32_t low = cur - prefix_idx > wsize ? cur - wsize : prefix_idx;
        const uint32_t max_rep = cur - low;
        if (rep2 > max_rep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > max_rep) { saved1 = rep1; rep1 = 0; }
    }
    // Walk state: ip0 is the start of the running search, sp its next pair (sp - ip0 even; sp > ip0 only while the step is still
    // 2).  `owed`: a match has just ended at ip0 == sp and the refill of ip0 - 2 and the repcode-2 loop are still to do.
    uint32_t sp = ip0;
    bool owed = false, owed_fill = false, gen_tail = false, gen_search = false;        // owed_fill: the refill of ip0 - 2 is part of what is owed (not behind an immediate repcode-2 match)
    // the next window's input and repeat-offset bytes, read while this window's sequences and table writes are produced
    bool pf_ok = false; uint32_t pf_sp = 0; Q16 pf_q0 = {0, 0, 0, 0}, pf_q1 = {0, 0, 0, 0}; uint64_t pf_ra = 0, pf_rb = 0;
    for (;;) {
        if (owed && (gen_tail || serial || step0 != 2 || sp < max(rep1, rep2) + 4 || int64_t(sp) + 200 > ilimit)) {
            after_match(S, tab, P, s, ip0, anchor, 0, rep1, rep2, end, ilimit, lane, false, owed_fill);
            sp = ip0; owed = false;
        }
        gen_tail = false;
        if (!owed) owed_fill = false;
        if (!serial && !gen_search && step0 == 2 && sp - ip0 <= 60 && sp >= max(rep1, rep2) + 4 && sp >= 4 && int64_t(sp) + 200 <= ilimit) {
            // ------------------------------------------------------------------------------------------ dense window
            ZPT(16);                                                                    // (everything outside the dense window)
            const uint32_t sp0 = U(sp);
            ip0 = U(ip0); anchor = U(anchor); rep1 = U(rep1); rep2 = U(rep2);
            const uint32_t pos = sp0 + uint32_t(lane);
            Q16 q0, q1;                                                                 // [pos - 4, pos + 28)
            uint64_t ra = 0, rb = 0;                                                    // [pos - rep - 4, pos - rep + 4) for both repeat offsets
            if (pf_ok && pf_sp == sp0) { q0 = pf_q0; q1 = pf_q1; ra = pf_ra; rb = pf_rb; }
            else {
                q0 = ld16(s + pos - 4); q1 = ld16(s + pos + 12);
                if (rep1) ra = ld8(s + pos - rep1 - 4);
                if (rep2) rb = ld8(s + pos - rep2 - 4);
            }
            pf_ok = false;
            ZCNT(0); ZPT(11);
            const uint32_t h = zhash(u64(q0.d1, q0.d2), hlog, mls);
            uint32_t ent = tld(tab, h);
            uint32_t hfill = 0;
            if (owed_fill) {                                                            // the owed refill of sp - 2 comes before every read of this window (written with the window's own)
                const uint32_t lo = __builtin_amdgcn_alignbit(q0.d1, q0.d0, 16), hi = __builtin_amdgcn_alignbit(q0.d2, q0.d1, 16);
                hfill = rl(zhash(u64(lo, hi), hlog, mls), 0);
                if (h == hfill) ent = sp0;
            }
            // lanes that share a table slot: candidates from a folded scoreboard, then one ballot per group makes it exact.
            // first lane of a slot: clean; second: its only predecessor in the window is `pred`; later ones: dirty.
            bool second = false, dirty = false; uint32_t pred = 0;
            unsigned long long grp = 0;                                                 // the lanes of this lane's slot, if it shares it
            {
                uint32_t* const sc = &L.score[h & 1023];
                atomicMin(sc, uint32_t(lane));
                const bool poss = *sc != uint32_t(lane);
                *sc = 0xFFFFFFFFu;
                unsigned long long mp = __ballot(poss);
#ifdef Z1_TRACE
                if (lane == 0) printf("prep sp0 %u poss %llx h9 %u h18 %u\n", sp0, mp, rl(h, 9), rl(h, 18));
#endif
                while (mp) {
                    const int e = __builtin_ctzll(mp);
                    const unsigned long long g = __ballot(h == rl(h, uint32_t(e)));
                    mp &= ~g;
                    if (g & (g - 1)) {
                        if (h == rl(h, uint32_t(e))) grp = g;
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi(uint32_t(g >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(g), 0));
                        if (h == rl(h, uint32_t(e)) && below) {
                            const unsigned long long gb = g & ((1ull << lane) - 1);
                            pred = 63u - uint32_t(__builtin_clzll(gb));
                            second = below == 1; dirty = below > 1;
                        }
                    }
                }
            }
            ZPT(12);
            const bool valid = ent >= prefix_idx;
            const uint32_t c = ent - 2;
            bool hit = false, slow = false;
            uint32_t fw = 0, bk = 0; uint64_t E = 0;                                    // E: which of the 48 bytes at [pos - 4, pos + 44) equal the candidate's (bit i: byte pos - 4 + i)
            if (valid) {
                if (c >= 4) {
                    const Q16 c0 = ld16(s + c - 4), c1 = ld16(s + c + 12), c2 = ld16(s + c + 28), q2 = ld16(s + pos + 28);
                    // zero bytes of the XOR words -> bit 7 of each byte; two words' flags gathered into one byte by a multiply
                    auto zb = [](uint32_t x) -> uint32_t { return ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x)) & 0x80808080u; };
                    auto two = [&](uint32_t xa, uint32_t xb) -> uint32_t { return ((((zb(xa) >> 7) | (zb(xb) >> 3)) * 0x00204081u) >> 21) & 0xFFu; };
                    const uint32_t Elo = two(q0.d0 ^ c0.d0, q0.d1 ^ c0.d1) | (two(q0.d2 ^ c0.d2, q0.d3 ^ c0.d3) << 8) | (two(q1.d0 ^ c1.d0, q1.d1 ^ c1.d1) << 16) | (two(q1.d2 ^ c1.d2, q1.d3 ^ c1.d3) << 24);
                    E = u64(Elo, two(q2.d0 ^ c2.d0, q2.d1 ^ c2.d1) | (two(q2.d2 ^ c2.d2, q2.d3 ^ c2.d3) << 8));
                    hit = ((Elo >> 4) & 15u) == 15u;
                    fw = uint32_t(__builtin_ctzll(~(E >> 8)));                          // (bit 40 of the complement is set: at most kFwHeld)
                    const uint32_t nb = ~Elo & 15u;                                       // bit 3 is position -1
                    bk = nb ? uint32_t(__builtin_clz(nb)) - 28u : 4u;
                } else { hit = ld4(s + c) == q0.d1; slow = true; }
            }
            const uint32_t info = fw | (bk << 8) | (uint32_t(bk == 4 && c - prefix > 4) << 12);   // bit 12: the catch-up may go beyond the 4 bytes held
            const uint32_t wpred = __shfl(q0.d1, int(pred));                            // (every lane takes part: the source lane must be active)
            const bool hitA = second && wpred == q0.d1;
            const unsigned long long m_dirty = __ballot(dirty), m_cond = __ballot(second && (hitA || hit)),
                                     m_slow = __ballot(!second && !dirty && hit && slow), m_hit = __ballot(!second && !dirty && hit && !slow);
            const unsigned long long m_stop = m_dirty | m_cond | m_slow;
#ifdef Z1_TRACE
            { const unsigned long long bs = __ballot(second), ba = __ballot(hitA), bh = __ballot(hit); if (lane == 0) printf("prep2 second %llx hitA %llx hit %llx pred18 %u\n", bs, ba, bh, rl(pred, 18)); }
#endif
            // repcode tests of every position, for both repeat offsets
            const uint64_t own = u64(q0.d0, q0.d1);
            // MR: the 4 bytes repeat; EQB: the byte repeats; EQM1: the byte before repeats.  H4 / HB: last lane for which MR / EQB
            // is known (63 when read from memory; a chosen match knows 28 bytes behind its position)
            unsigned long long MR1 = 0, EQB1 = 0, EQM1 = 0, MR2 = 0, EQB2 = 0, EQM2 = 0;
            int H41 = 63, HB1 = 63, H42 = 63, HB2 = 63;
            if (rep1) { const uint64_t x = ra ^ own; MR1 = __ballot(uint32_t(x >> 32) == 0); EQB1 = __ballot((uint32_t(x >> 32) & 0xFFu) == 0); EQM1 = __ballot((uint32_t(x) >> 24) == 0); }
            if (rep2) { const uint64_t x = rb ^ own; MR2 = __ballot(uint32_t(x >> 32) == 0); EQB2 = __ballot((uint32_t(x >> 32) & 0xFFu) == 0); EQM2 = __ballot((uint32_t(x) >> 24) == 0); }
            auto read_rep = [&](uint32_t off, int lo, unsigned long long& mr, unsigned long long& eqb, unsigned long long& eqm1) {
                uint64_t x = ~0ull;                                                     // (lanes >= lo lie behind a match at that offset: the read stays inside the input)
                if (lane >= lo) x = ld8(s + pos - off - 4) ^ own;
                mr = __ballot(uint32_t(x >> 32) == 0); eqb = __ballot((uint32_t(x >> 32) & 0xFFu) == 0); eqm1 = __ballot((uint32_t(x) >> 24) == 0);
                ZCNT(9);
            };
            // Where the walk goes from a chosen hash hit at this lane, if nothing but another plain hash hit follows: the search
            // behind the match (start e) meets its first event at lane nx, all repcode tests in between (offsets known from E) fail,
            // and the match at nx needs nothing the lanes do not hold.  64: anything else - the walk's general step decides.
            uint32_t nxt = 64;
            {
                const int e = lane + 4 + int(fw);
                const unsigned long long sh = e < 64 ? (m_hit | m_stop) >> e : 0ull;
                const int nx = e + (sh ? __builtin_ctzll(sh) : 64);
                const int a = nx - ((nx - e) & 1), amax = 60 - ((60 - e) & 1);
                const uint32_t infn = __shfl(info, nx & 63);                            // (every lane takes part)
                const bool plain = ((m_hit >> lane) & 1) && fw < kFwHeld && e < 58 && a <= amax && !((m_stop >> (nx & 63)) & 1) && a + 5 - lane <= kFwHeld + 3;
                if (plain) {
                    const uint64_t Es = E >> 4, M4 = Es & (Es >> 1) & (Es >> 2) & (Es >> 3);         // bit d: the 4 bytes at pos + d repeat
                    const uint64_t T = (0x5555555555555555ull << (e + 2 - lane)) & ((2ull << (a + 2 - lane)) - 1); // the tests of that search: e+2, e+4 .. a+2
                    if (!(M4 & T) && !(((infn >> 12) & 1) && nx - e > 4) && (infn & 63) != kFwHeld) nxt = uint32_t(nx);
                }
            }
            // ---- the walk (scalar).  Lanes are window positions; s_l: start of the running search, cur: its next pair,
            // anc: the anchor (both may lie before the window: negative).  The walk only CHOOSES: which hash-hit lanes (selH) and
            // which repcode-hit lanes (selR) start a match; lengths, literal runs, sequence numbers and the table writes follow
            // from the chosen lanes for all lanes at once, behind the walk.
            // Repeat-offset knowledge comes in two forms: masks over the window (k = 0; MR: the 4 bytes repeat, EQB: the byte
            // repeats, EQM1: the byte before repeats; H4 / HB: last lane for which MR / EQB is known - 63 when read from memory),
            // or the 32 byte-equalities E of the hash match at lane m that made the offset (k = 1): masks are built from them
            // only when a step needs them.
            ZPT(13);
            int s_l = -int(sp0 - ip0), cur = 0, anc = -int(sp0 - anchor);
            const int anc0 = anc;
            bool pend = owed, lastI = false;                                            // pend: the repcode-2 test at lane cur == s_l comes first; lastI: the last match was an immediate repcode-2 match
            unsigned long long selH = 0, selR = 0, selI = 0;                            // chosen: hash hits, repcode hits, immediate repcode-2 matches
            uint32_t endv = uint32_t(lane) + 4 + fw, brep = 0;                          // per lane: end of the match starting here; rep lanes: the byte before repeats
            uint32_t r1 = rep1, r2 = rep2;
            int k1 = 0, k2 = 0, m1 = 0, m2 = 0; uint64_t E1 = 0, E2 = 0;
            int endk;                                                                   // 0: search goes on at cur, 1: fresh search at cur, 2: batched search at cur, 3: repcode-2 loop at cur
            const unsigned long long m_ev = m_hit | m_stop;
            // (all of the walk's state is wave-uniform: pinned to scalar registers on every way into the loop head, or the loop runs on the vector unit)
#define WPIN() do { cur = Ui(cur); s_l = Ui(s_l); anc = Ui(anc); r1 = U(r1); r2 = U(r2); k1 = Ui(k1); k2 = Ui(k2); m1 = Ui(m1); m2 = Ui(m2); E1 = U64(E1); E2 = U64(E2); \
                    H41 = Ui(H41); HB1 = Ui(HB1); H42 = Ui(H42); HB2 = Ui(HB2); pend = Ui(int(pend)) != 0; lastI = Ui(int(lastI)) != 0; selH = U64(selH); selR = U64(selR); selI = U64(selI); MR1 = U64(MR1); EQB1 = U64(EQB1); EQM1 = U64(EQM1); MR2 = U64(MR2); EQB2 = U64(EQB2); EQM2 = U64(EQM2); } while (0)
            WPIN();
            for (;;) {
                // ---- general step
                if (k1) { EQB1 = ((E1 >> 4) & ((1ull << (kFwHeld + 4)) - 1)) << m1; EQM1 = EQB1 << 1; MR1 = EQB1 & (EQB1 >> 1) & (EQB1 >> 2) & (EQB1 >> 3); HB1 = min(63, m1 + kFwHeld + 3); H41 = HB1 - 3; k1 = 0; }
                if (k2) { EQB2 = ((E2 >> 4) & ((1ull << (kFwHeld + 4)) - 1)) << m2; EQM2 = EQB2 << 1; MR2 = EQB2 & (EQB2 >> 1) & (EQB2 >> 2) & (EQB2 >> 3); HB2 = min(63, m2 + kFwHeld + 3); H42 = HB2 - 3; k2 = 0; }
                if (pend) {
                    if (cur >= 62) { endk = 1; break; }
                    if (r2) {
                        if (cur > H42) { read_rep(r2, cur, MR2, EQB2, EQM2); H42 = HB2 = 63; }
                        if ((MR2 >> cur) & 1) {                                         // an immediate repcode-2 match (zstd_fast.c:270-281): offsets swap, only its first position enters the table
                            const int q = cur;
                            const unsigned long long t = q + 4 < 64 ? ~(EQB2 >> (q + 4)) : 1ull;
                            const int fwv = t ? __builtin_ctzll(t) : 64;
                            if (q + 4 + fwv > HB2 && HB2 < 63) { read_rep(r2, cur, MR2, EQB2, EQM2); H42 = HB2 = 63; { WPIN(); continue; } }
                            if (q + 4 + fwv >= 64) { endk = 3; break; }                 // it runs to the end of the window: the serial loop takes it
                            const int e = q + 4 + fwv;
                            selI |= 1ull << q; lastI = true;
                            if (lane == q) { endv = uint32_t(e); brep = 0; }
                            ZCNT(1);
                            { const uint32_t tr = r1; r1 = r2; r2 = tr; }
                            { unsigned long long tq; tq = MR1; MR1 = MR2; MR2 = tq; tq = EQB1; EQB1 = EQB2; EQB2 = tq; tq = EQM1; EQM1 = EQM2; EQM2 = tq; }
                            { int ti; ti = H41; H41 = H42; H42 = ti; ti = HB1; HB1 = HB2; HB2 = ti; }
                            anc = s_l = cur = e;
                            { WPIN(); continue; }
                        }
                    }
                    pend = false;
                }
                const int amax = 60 - ((60 - s_l) & 1);                                 // last pair handled here: its refills stay inside the window
                if (cur > amax) { endk = cur == s_l ? 1 : 0; break; }
                const unsigned long long from = ~0ull << cur;
                const unsigned long long pm = (s_l & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
                const unsigned long long k4 = H41 >= 63 ? ~0ull : (2ull << H41) - 1;    // lanes whose repcode test is known
                const unsigned long long evh = m_ev & from, evr = MR1 & pm & (from << 2) & k4, unk = pm & (from << 2) & ~k4;
                const int lh = evh ? __builtin_ctzll(evh) : 64, lr = evr ? __builtin_ctzll(evr) : 999, lu = unk ? __builtin_ctzll(unk) : 999;   // (999: none)
                if (r1 && lr > lh + 2 && lu <= lh + 2 && lu - 2 <= amax) {              // a repcode test that is not known comes first: read it
                    read_rep(r1, cur, MR1, EQB1, EQM1); H41 = HB1 = 63;
                    { WPIN(); continue; }
                }
                if (lr <= lh + 2) {                                                     // the repcode test at lr comes before the hash tests of lr-2, lr-1
                    const int q = lr, a = q - 2;
                    if (a > amax) { cur = amax + 2; endk = 0; break; }
                    const unsigned long long t = q + 4 < 64 ? ~(EQB1 >> (q + 4)) : 1ull;
                    const int fwv = t ? __builtin_ctzll(t) : 64;
                    if (q + 4 + fwv > HB1 && HB1 < 63) { read_rep(r1, cur, MR1, EQB1, EQM1); H41 = HB1 = 63; { WPIN(); continue; } }   // the run reaches the last byte known
                    if (q + 4 + fwv >= 64) {                                            // the match runs to the end of the window: its length is not known here
                        cur = a;
                        if (a > 0) endk = a == s_l ? 1 : 0;                             // a window starting at its pair sees 58 bytes of it
                        else endk = 2;
                        break;
                    }
                    const int e = q + 4 + fwv;
                    selR |= 1ull << q; lastI = false;
                    if (lane == q) { endv = uint32_t(e); brep = uint32_t(EQM1 >> q) & 1u; }
                    ZCNT(1);
                    anc = s_l = cur = e; pend = true;
                    { WPIN(); continue; }
                }
                if (lh >= 64) { cur = amax + 2; endk = 0; break; }
                const int m = lh, a = m - ((m - s_l) & 1);
                if (a > amax) { cur = amax + 2; endk = 0; break; }
                if ((m_stop >> m) & 1) {
                    cur = a;
                    if ((m_dirty >> m) & 1) { endk = a == s_l ? 1 : 0; ZCNT(4); }       // a window starting at its pair reads the table
                    else endk = 2;
                    break;
                }
                const uint32_t inf = rl(info, uint32_t(m));
                if (((inf >> 12) & 1) && m - anc > 4) { cur = a; endk = 2; break; }     // the catch-up goes on in memory
                const uint32_t cm = rl(c, uint32_t(m));
                uint32_t fwm = inf & 63;
                if (fwm == kFwHeld) fwm = kFwHeld + count_fwd(s, sp0 + uint32_t(m) + 4 + kFwHeld, cm + 4 + kFwHeld, end, lane);   // the match runs past the bytes held
                const int e = m + 4 + int(fwm);
                selH |= 1ull << m; lastI = false;
                if (lane == m) endv = uint32_t(e);
                ZCNT(2);
                // repcode-2 test behind this match (offset: the previous repeat offset), as far as it is known here
                bool bad0 = false;
                if (r1) {
                    if (k1) { const int d = e - m1; bad0 = d > int(kFwHeld) || ((E1 >> ((d + 4) & 63)) & 15u) == 15u; }
                    else bad0 = e > H41 || ((MR1 >> (e & 63)) & 1);
                }
                r2 = r1; r1 = sp0 + uint32_t(m) - cm; MR2 = MR1; EQB2 = EQB1; EQM2 = EQM1; H42 = H41; HB2 = HB1; k2 = 0;
                k1 = 1; m1 = m; E1 = rl64(E, uint32_t(m));
                anc = s_l = cur = e; pend = true;
                // ---- the chain of plain hash hits behind it: one readlane per sequence
                unsigned long long selc = 0;
                for (int t = m;;) {
                    t = int(rl(nxt, uint32_t(t)));
                    if (t >= 64) break;
                    selc |= 1ull << t;
                }
                if (selc) {
                    // every link took the repcode-2 test behind its predecessor p for granted (offset: that of p's predecessor pp,
                    // whose bytes know it up to 24 bytes behind pp): check them all at once, cut the chain at the first that fails
                    const unsigned long long C = selc | (1ull << m);
                    const unsigned long long lt = (1ull << lane) - 1;
                    const unsigned long long bl = C & lt;
                    const int P = bl ? 63 - __builtin_clzll(bl) : 0;
                    const uint32_t packP = __shfl(uint32_t(P) | (uint32_t(bl != 0) << 8), P);     // my predecessor's predecessor (| it has one << 8)
                    const int PP = int(packP & 63);
                    const int eP = int(__shfl(endv, P));
                    const uint64_t Epp = u64(__shfl(uint32_t(E), PP), __shfl(uint32_t(E >> 32), PP));
                    const int d = eP - PP;
                    const bool bad = ((selc >> lane) & 1) && (((packP >> 8) & 1) ? (d > int(kFwHeld) || ((Epp >> ((d + 4) & 63)) & 15u) == 15u) : bad0);
                    const unsigned long long badm = __ballot(bad);
                    if (badm) selc &= (1ull << __builtin_ctzll(badm)) - 1;
                    ZADD(10, uint32_t(__builtin_popcountll(selc)));
                    if (selc) {
                        const int Lst = 63 - __builtin_clzll(selc);
                        const unsigned long long bL = (selc | (1ull << m)) & ((1ull << Lst) - 1);
                        const int Lp = 63 - __builtin_clzll(bL);                         // (m at least)
                        selH |= selc;
                        r2 = sp0 + uint32_t(Lp) - rl(c, uint32_t(Lp)); r1 = sp0 + uint32_t(Lst) - rl(c, uint32_t(Lst));
                        k2 = 1; m2 = Lp; E2 = rl64(E, uint32_t(Lp)); m1 = Lst; E1 = rl64(E, uint32_t(Lst));
                        anc = s_l = cur = int(rl(endv, uint32_t(Lst)));
                    }
                }
                { WPIN(); }
            }
#undef WPIN
            ZPT(14);
            {   // where the next window starts is known: its reads go out now
                const uint32_t nsp = sp0 + uint32_t(cur), nip = uint32_t(int(sp0) + s_l);
                if (endk < 2 && nsp - nip <= 60 && nsp >= max(r1, r2) + 4 && int64_t(nsp) + 200 <= ilimit) {
                    const uint32_t np = nsp + uint32_t(lane);
                    pf_q0 = ld16(s + np - 4); pf_q1 = ld16(s + np + 12);
                    pf_ra = r1 ? ld8(s + np - r1 - 4) : 0; pf_rb = r2 ? ld8(s + np - r2 - 4) : 0;
                    pf_ok = true; pf_sp = nsp;
                }
            }
            // ---- behind the walk, all lanes at once: the sequences (ZSTD_storeSeq) and the table writes of the chosen matches
            {
                const unsigned long long chosen = selH | selR | selI;
                const bool isH = (selH >> lane) & 1, isR = ((selR | selI) >> lane) & 1, isC = isH || isR;
                const uint32_t kind = isH ? 1u : ((selI >> lane) & 1) ? 2u : 0u;        // 1: hash hit, 0: repcode hit, 2: immediate repcode-2 match
                const uint32_t nbelow = __builtin_amdgcn_mbcnt_hi(uint32_t(chosen >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(chosen), 0));
                const unsigned long long below = chosen & ((1ull << lane) - 1);
                const int P = below ? 63 - __builtin_clzll(below) : 0;
                const bool tail = isC && int64_t(sp0) + int64_t(endv) > ilimit;       // the match ends behind ilimit: no refills follow it (zstd_fast.c:263)
                const uint32_t endP = __shfl(endv, P), kindP = __shfl(kind | (uint32_t(tail) << 3), P);
                const int ancl = below ? int(endP) : anc0;                              // my anchor: the end of the match chosen before me
                uint32_t ll = 0;
                if (isC) {
                    const uint32_t b = isH ? min(bk, min(uint32_t(lane - ancl), c - prefix)) : brep;
                    ll = uint32_t(lane - ancl) - b;
                    const uint32_t at = S.nseq + nbelow;
                    S.ll[at] = ll; S.ml[at] = 1 + b + (endv - uint32_t(lane) - 4); S.off[at] = isH ? pos - c + 3 : 1u;
                }
                S.nseq += uint32_t(__builtin_popcountll(chosen));
                S.nlit += rl(scan_add(ll), 63);
                // lanes the reference writes into the table: every lane the walk passed as a probe; of a match starting at lane
                // P and ending at e: P (+1, +2 behind a hash hit) and e - 2 (not behind an immediate repcode-2 match)
                const int Pl = isC ? lane : P;
                const bool any = isC || below != 0;
                const int eP = isC ? int(endv) : int(endP);
                const uint32_t kPt = isC ? (kind | (uint32_t(tail) << 3)) : kindP, kP = kPt & 7;
                const bool fills = !(kPt & 8);                                         // (kind 1: P is a probe, P + 1 the pair's second write, P + 2 the refill of current0 + 2; kind 0: P is that refill)
                const int d = lane - Pl;
                const bool visited = lane < cur && (!any || lane >= eP || (d == 0 && (kP != 0 || fills)) || (kP == 1 && (d == 1 || (d == 2 && fills))) || (kP != 2 && fills && lane == eP - 2));
                // plain stores; of the lanes of one slot the last one writes (and the owed refill only if no lane has its slot)
                const unsigned long long vism = __ballot(visited);
#ifdef Z1_TRACE
                { const unsigned long long stm = __ballot(visited && !(grp & vism & ~((2ull << lane) - 1))); if (lane == 0) printf("vis sp0 %u vism %llx storem %llx selI %llx cur %d h46 %u h62 %u ent30 %u\n", sp0, vism, stm, selI, cur, rl(h, 46), rl(h, 62), rl(ent, 30)); }
#endif
                if (visited && !(grp & vism & ~((2ull << lane) - 1))) tab[h] = pos + 2;
#ifdef Z1_DOUBLECOMMIT
                asm volatile("" ::: "memory");
                if (visited && !(grp & vism & ~((2ull << lane) - 1))) __builtin_nontemporal_store(pos + 2, &tab[h]);
#endif
                const unsigned long long fillm = __ballot(visited && h == hfill);
                if (owed_fill && lane == 0 && !fillm) tab[hfill] = sp0;
            }
            anchor = uint32_t(int(sp0) + anc); ip0 = uint32_t(int(sp0) + s_l); sp = sp0 + uint32_t(cur);
            rep1 = r1; rep2 = r2;
            owed = (endk == 1 && pend) || endk == 3;
            owed_fill = owed && ((selH | selR | selI) ? !lastI : owed_fill);            // (no match in this window: what was owed on the way in)
            gen_tail = endk == 3; gen_search = endk == 2;
            ZCNT(5 + endk); ZPT(15);
#ifdef Z1_TRACE
            if (lane == 0) printf("win sp0 %u endk %d cur %d s_l %d anc %d selH %llx selR %llx hit %llx stop %llx dirty %llx cond %llx r1 %u r2 %u\n", sp0, endk, cur, s_l, anc, selH, selR, m_hit, m_stop, m_dirty, m_cond, r1, r2);
#endif
            anchor = U(anchor); ip0 = U(ip0); sp = U(sp); rep1 = U(rep1); rep2 = U(rep2);
            continue;
        }
        gen_search = false; pf_ok = false;
        // ---------------------------------------------------------------------------------------------- one search, one sequence
        uint32_t match0 = 0, mlen = 0, off_base = 0, cur0 = 0;
        bool found = false;
        if (sp == ip0 && int64_t(ip0) + step0 + 1 >= ilimit) break;
        if (serial) {
            // wave-uniform transcription of the reference loop (debug / cross-check path)
            uint32_t step = step0, next_step = ip0 + 128, ip1 = ip0 + 1, ip2 = ip0 + step, ip3 = ip2 + 1;
            uint32_t h0 = zhash(ld8(s + ip0), hlog, mls), h1 = zhash(ld8(s + ip1), hlog, mls);
            uint32_t idx = tab[h0];
            int kind = 0;                                        // 1 repcode, 2 hash hit
            for (;;) {
                const uint32_t rval = ld4(s + ip2 - rep1);
                cur0 = ip0 + 2;
                if (lane == 0) tab[h0] = cur0;
                if ((ld4(s + ip2) == rval) & (rep1 > 0)) { if (lane == 0) tab[h1] = ip1 + 2; kind = 1; break; }
                if (idx >= prefix_idx && ld4(s + idx - 2) == ld4(s + ip0)) { if (lane == 0) tab[h1] = ip1 + 2; kind = 2; break; }
                idx = tab[h1]; h0 = h1; h1 = zhash(ld8(s + ip2), hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip3;
                cur0 = ip0 + 2;
                if (lane == 0) tab[h0] = cur0;
                if (idx >= prefix_idx && ld4(s + idx - 2) == ld4(s + ip0)) { if (step <= 4 && lane == 0) tab[h1] = ip1 + 2; kind = 2; break; }
                idx = tab[h1]; h0 = h1; h1 = zhash(ld8(s + ip2), hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
                if (ip2 >= next_step) { step++; next_step += 128; }
                if (int64_t(ip3) >= ilimit) break;
            }
            if (kind == 0) break;                                // _cleanup
            if (kind == 1) {
                ip0 = ip2; match0 = ip0 - rep1;
                mlen = (s[ip0 - 1] == s[match0 - 1]) ? 1 : 0;
                ip0 -= mlen; match0 -= mlen; off_base = 1; mlen += 4;
            } else {
                match0 = idx - 2;
                rep2 = rep1; rep1 = ip0 - match0; off_base = rep1 + 3; mlen = 4;
                while (ip0 > anchor && match0 > prefix && s[ip0 - 1] == s[match0 - 1]) { ip0--; match0--; mlen++; }
            }
            found = true;
        } else {
            // ---- batched search: lane j owns pair j = positions (b, b+1) and the repcode test at b+sj
            uint32_t B = sp, SJ = step0, V = step0, NS = ip0 + 128, width = 16;
            int ev_kind = 0;                                     // 0 none yet, 1 rep, 2 hit at b, 3 hit at b+1, 4 end of block
            uint32_t ev_b = 0, ev_s = 0, ev_v = 0, ev_idx = 0;
            for (;;) {
                // pair state of this lane and of its successor
                uint32_t b = B, sj = SJ, v = V, ns = NS;
                uint32_t nb_ = 0, nsj = 0, nv = 0, nns = 0;
                if (SJ == V && B + V * (width + 1) < NS) {
                    b = B + V * uint32_t(lane);                  // no step change inside this batch
                } else {
                    for (uint32_t i = 0; i < width; i++) {
                        nb_ = b + sj; nsj = v; nns = ns; nv = v;
                        if (nb_ + v >= ns) { nv = v + 1; nns = ns + 128; }
                        if (uint32_t(lane) > i) { b = nb_; sj = nsj; v = nv; ns = nns; }
                    }
                }
                // successor of (b, sj, v, ns) for this lane
                nb_ = b + sj; nsj = v; nns = ns; nv = v;
                if (nb_ + v >= ns) { nv = v + 1; nns = ns + 128; }
                const bool act = uint32_t(lane) < width;
                const bool term = int64_t(nb_) + 1 + int64_t(nsj) >= ilimit;            // this pair's iteration ends the block
                // speculative lanes may lie beyond the block: keep their reads inside the input
                const bool inb = int64_t(b) + 1 + int64_t(sj) < ilimit;
                const uint32_t rb = inb ? b : start;
                const uint64_t w0 = ld8(s + rb), w1 = ld8(s + rb + 1);
                const uint32_t ip2 = inb ? b + sj : start + 4;
                const uint32_t r_cur = ld4(s + ip2), r_rep = ld4(s + ip2 - (inb ? rep1 : 0));
                const uint32_t h0 = zhash(w0, hlog, mls), h1 = zhash(w1, hlog, mls);
                uint32_t i0 = 0, i1 = 0; bool shared = false;
                if (act && inb) {
                    i0 = tld(tab, h0); i1 = (h1 == h0) ? b + 2 : tld(tab, h1);
                    uint32_t* const sc0 = &L.score[h0 & 1023]; uint32_t* const sc1 = &L.score[h1 & 1023];
                    atomicMin(sc0, uint32_t(lane)); atomicMin(sc1, uint32_t(lane));
                    shared = (*sc0 != uint32_t(lane)) || (*sc1 != uint32_t(lane));
                    *sc0 = 0xFFFFFFFFu; *sc1 = 0xFFFFFFFFu;
                }
                const bool ok0 = act && inb && i0 >= prefix_idx, ok1 = act && inb && i1 >= prefix_idx;
                const uint32_t c0 = ld4(s + (ok0 ? i0 - 2 : 0)), c1 = ld4(s + (ok1 ? i1 - 2 : 0));
                int kind = 0;
                if (act && inb) {
                    if ((r_cur == r_rep) & (rep1 > 0)) kind = 1;
                    else if (ok0 && c0 == uint32_t(w0)) kind = 2;
                    else if (ok1 && c1 == uint32_t(w1)) kind = 3;
                    else if (term) kind = 4;
                }
                const unsigned long long cutm = __ballot(act && (shared || !inb)) & ~1ull;     // lane 0 never conflicts with an earlier lane
                const int cut = cutm ? __builtin_ctzll(cutm) : int(width);
                const unsigned long long evm = __ballot(kind != 0) & ((cut >= 64) ? ~0ull : ((1ull << cut) - 1));
                if (evm) {
                    const int J = __builtin_ctzll(evm);
                    // commit table writes of lanes <= J (both positions; on every exit path the second is inserted too)
                    if (lane <= J) { tab[h0] = b + 2; tab[h1] = b + 3; }
                    ev_kind = int(rl(uint32_t(kind), J)); ev_b = rl(b, J); ev_s = rl(sj, J); ev_v = rl(v, J);
                    ev_idx = ev_kind == 2 ? rl(i0, J) : rl(i1, J);
                    break;
                }
                // nothing in [0, cut): commit them and continue from pair `cut`
                if (lane < cut) { tab[h0] = b + 2; tab[h1] = b + 3; }
                if (cut < int(width)) { B = rl(b, cut); SJ = rl(sj, cut); V = rl(v, cut); NS = rl(ns, cut); }
                else { B = rl(nb_, width - 1); SJ = rl(nsj, width - 1); V = rl(nv, width - 1); NS = rl(nns, width - 1); }
                width = min(64u, width * 2);
            }
            if (ev_kind == 4) break;                             // _cleanup
            ZCNT(3);
#ifdef Z1_TRACE
            if (lane == 0) printf("gen sp %u ip0 %u kind %d b %u idx %u\n", sp, ip0, ev_kind, ev_b, ev_idx);
#endif
            uint32_t room;
            if (ev_kind == 1) {
                cur0 = ev_b + 2;
                ip0 = ev_b + ev_s; match0 = ip0 - rep1;
                off_base = 1;
                room = 1;                                        // mLength = ip0[-1] == match0[-1] (zstd_fast.c:236)
            } else {
                if (ev_kind == 3) {
                    ip0 = ev_b + 1;
                    if (ev_v <= 4) { const uint32_t q = ev_b + ev_s; const uint32_t h = zhash(ld8(s + q), hlog, mls); if (lane == 0) tab[h] = q + 2; }
                } else ip0 = ev_b;
                cur0 = ip0 + 2;
                match0 = ev_idx - 2;
                rep2 = rep1; rep1 = ip0 - match0; off_base = rep1 + 3;
                room = min(ip0 - anchor, match0 - prefix);
            }
            {
                // backward and forward extension in ONE round trip: where the forward count starts does not depend on
                // how far the match reaches back (it is the 4 bytes behind the event position either way)
                const uint32_t fa = ip0 + 4, fb = match0 + 4;
                const bool wide = fa + 1024 <= end;
                const bool bl = uint32_t(lane) < room;
                const uint32_t b_i = bl ? uint32_t(s[ip0 - 1 - lane]) : 0u, b_m = bl ? uint32_t(s[match0 - 1 - lane]) : 1u;
                U16B x = {0, 0}, y = {0, 0}; uint32_t f_i = 0, f_m = 1;
                if (wide) { x = *reinterpret_cast<const U16B*>(s + fa + 16 * lane); y = *reinterpret_cast<const U16B*>(s + fb + 16 * lane); }
                else if (fa + lane < end) { f_i = s[fa + lane]; f_m = s[fb + lane]; }
                {
                    const unsigned long long bad = ~__ballot(bl && b_i == b_m);
                    uint32_t k = bad ? uint32_t(__builtin_ctzll(bad)) : 64u;
                    ip0 -= k; match0 -= k; mlen = 4 + k;
                    while (k == 64) {                            // (only hash hits can get here: room is 1 for a repcode)
                        const uint32_t room2 = min(ip0 - anchor, match0 - prefix);
                        const bool same = uint32_t(lane) < room2 && s[ip0 - 1 - lane] == s[match0 - 1 - lane];
                        const unsigned long long bad2 = ~__ballot(same);
                        k = bad2 ? uint32_t(__builtin_ctzll(bad2)) : 64u;
                        ip0 -= k; match0 -= k; mlen += k;
                    }
                }
                if (wide) {
                    const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
                    const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
                    const unsigned long long bad = __ballot(eq < 16);
                    if (bad) { const int l = __builtin_ctzll(bad); mlen += 16 * l + rl(eq, l); }
                    else mlen += 1024 + count_fwd(s, fa + 1024, fb + 1024, end, lane);
                } else {
                    const unsigned long long bad = ~__ballot(fa + lane < end && f_i == f_m);
                    if (bad) mlen += uint32_t(__builtin_ctzll(bad));
                    else mlen += 64 + count_fwd(s, fa + 64, fb + 64, end, lane);
                }
            }
            found = true;
        }
        if (!found) break;
        if (serial) mlen += count_fwd(s, ip0 + mlen, match0 + mlen, end, lane);
        store_seq(S, s, anchor, ip0 - anchor, off_base, mlen, lane);
        ip0 += mlen; anchor = ip0;
        after_match(S, tab, P, s, ip0, anchor, cur0, rep1, rep2, end, ilimit, lane);
        sp = ip0;
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return end - anchor;
}


// ZSTD_compressBlock_doubleFast_noDict_generic (compress/zstd_double_fast.c:98-330) over s[start, end): level 3.
// Lane j speculates position j of the current search: long (8-byte) and short hash probes, the repcode test at
// ip+1 and the "next long" probe at ip1; same first-event / commit / scoreboard-cut rules as fast_block.
__device__ __forceinline__ uint32_t dfast_block(ZLds& L, SeqStore& S, uint32_t* tl, uint32_t* ts, const Params& P, uint32_t rep[3],
                                                const uint8_t* s, uint32_t start, uint32_t end, bool serial, int lane)
{
    const uint32_t hl_log = P.hlog, hs_log = P.clog, wsize = 1u << P.wlog, mls = P.mml;
    const uint32_t prefix_idx = end > wsize ? end + 2 - wsize : 2;
    const uint32_t prefix = prefix_idx - 2;
    const int64_t ilimit = int64_t(end) - 8;
    uint32_t anchor = start, ip = start;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    auto HL = [&](uint64_t v) -> uint32_t { return uint32_t((v * 0xCF1BBCDCB7A56463ull) >> (64 - hl_log)); };

    ip += (ip == prefix) ? 1 : 0;
    {
        const uint32_t cur = ip + 2;
        const uint32_t low = cur - prefix_idx > wsize ? cur - wsize : prefix_idx;
        const uint32_t max_rep = cur - low;
        if (rep2 > max_rep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > max_rep) { saved1 = rep1; rep1 = 0; }
    }
    // Walk state as in fast_block: ip is the start of the running search, sp its next position (the step is 1 while
    // sp - ip < 192).  `owed`: a match has just ended at ip == sp and its two end refills (long: ip - 2, short: ip - 1) and the
    // repcode-2 loop are still to do.
    uint32_t sp = ip;
    bool owed = false, owed_fill = false, gen_tail = false, gen_search = false;   // owed_fill: the two end refills are part of what is owed (not behind an immediate repcode-2 match)
    // refills behind a match that ended at `at` and the repcode-2 loop (zstd_double_fast.c:262-293); the refill of curr + 2 only
    // when it has not been made (dense window: it is one of the window's lanes)
    auto after_match = [&](uint32_t cur0, bool fill_cur, bool fill_end) {
        if (int64_t(ip) > ilimit) return;
        const uint64_t wa = fill_cur ? ld8(s + cur0) : 0, wb = ld8(s + ip - 2), wc = ld8(s + ip - 1);
        uint32_t r_cur = ld4(s + ip), r_rep = ld4(s + ip - rep2);     // (the reads of this step in one round trip; rep2 == 0: not used)
        if (lane == 0) {
            if (fill_cur) { tl[HL(wa)] = cur0 + 2; ts[zhash(wa, hs_log, mls)] = cur0 + 2; }
            if (fill_end) { tl[HL(wb)] = ip; ts[zhash(wc, hs_log, mls)] = ip + 1; }
        }
        for (bool first = true; int64_t(ip) <= ilimit && rep2 > 0; first = false) {
            if (!first) { r_cur = ld4(s + ip); r_rep = ld4(s + ip - rep2); }
            if (r_cur != r_rep) break;
            const uint32_t rlen = count_fwd(s, ip + 4, ip + 4 - rep2, end, lane) + 4;
            const uint32_t t = rep2; rep2 = rep1; rep1 = t;
            const uint64_t wi = ld8(s + ip);
            if (lane == 0) { ts[zhash(wi, hs_log, mls)] = ip + 2; tl[HL(wi)] = ip + 2; }
            store_seq(S, s, anchor, 0, 1, rlen, lane);
            ip += rlen; anchor = ip;
        }
    };
    for (;;) {                                                   // one dense window, or one search of the batched path, per iteration
        if (owed && (gen_tail || serial || sp < max(rep1, rep2) + 4 || int64_t(sp) + 200 > ilimit)) { after_match(0, false, owed_fill); sp = ip; owed = false; }
        gen_tail = false;
        if (!owed) owed_fill = false;
        if (!serial && !gen_search && sp - ip <= 128 && sp >= max(rep1, rep2) + 4 && sp >= 4 && int64_t(sp) + 200 <= ilimit) {     // (serial: the batched search only - cross-check path)
            // ------------------------------------------------------------------------------------------ dense window
            // As in fast_block (see there), for two tables: lane l takes position sp + l and prepares, against both tables as they
            // stand, its long candidate (8-byte test) and its short one (4-byte test) with 48 bytes of each: byte-equality masks
            // EL / ES give the tests, the lengths, the catch-up and - for the candidate a chosen match used - the repcode tests
            // behind it.  The reference examines position p as: repcode at p+1, long at p, short at p (then long at p+1 decides
            // between p+1 and p).  A scalar walk chooses; sequences and both tables' writes follow from the chosen lanes.
            const uint32_t sp0 = U(sp);
            ip = U(ip); anchor = U(anchor); rep1 = U(rep1); rep2 = U(rep2);
            const uint32_t pos = sp0 + uint32_t(lane);
            const Q16 q0 = ld16(s + pos - 4), q1 = ld16(s + pos + 12), q2 = ld16(s + pos + 28);          // [pos - 4, pos + 44)
            ZCNT(0);
            uint64_t ra = 0, rb = 0;
            if (rep1) ra = ld8(s + pos - rep1 - 4);
            if (rep2) rb = ld8(s + pos - rep2 - 4);
            const uint64_t w8 = u64(q0.d1, q0.d2);
            const uint32_t hl = HL(w8), hs = zhash(w8, hs_log, mls);
            uint32_t entL = tld(tl, hl), entS = tld(ts, hs);
            uint32_t hfL = 0, hfS = 0;
            if (owed_fill) {                                     // the owed refills (long: sp - 2, short: sp - 1) come before every read of this window
                const uint64_t v2 = u64(__builtin_amdgcn_alignbit(q0.d1, q0.d0, 16), __builtin_amdgcn_alignbit(q0.d2, q0.d1, 16));
                const uint64_t v1 = u64(__builtin_amdgcn_alignbit(q0.d1, q0.d0, 24), __builtin_amdgcn_alignbit(q0.d2, q0.d1, 24));
                hfL = rl(HL(v2), 0); hfS = rl(zhash(v1, hs_log, mls), 0);
                if (hl == hfL) entL = sp0;
                if (hs == hfS) entS = sp0 + 1;
            }
            // lanes sharing a slot, per table (folded scoreboard for candidates, one ballot per group makes it exact)
            bool secL = false, dirL = false, secS = false, dirS = false; uint32_t predL = 0, predS = 0;
            unsigned long long grpL = 0, grpS = 0;
            auto groups = [&](uint32_t h, uint32_t* sc, bool& second, bool& dirty, uint32_t& pred, unsigned long long& grp) {
                atomicMin(sc, uint32_t(lane));
                const bool poss = *sc != uint32_t(lane);
                *sc = 0xFFFFFFFFu;
                unsigned long long mp = __ballot(poss);
                while (mp) {
                    const uint32_t he = rl(h, uint32_t(__builtin_ctzll(mp)));
                    const unsigned long long g = __ballot(h == he);
                    mp &= ~g;
                    if (g & (g - 1)) {
                        if (h == he) grp = g;
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi(uint32_t(g >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(g), 0));
                        if (h == he && below) {
                            const unsigned long long gb = g & ((1ull << lane) - 1);
                            pred = 63u - uint32_t(__builtin_clzll(gb));
                            second = below == 1; dirty = below > 1;
                        }
                    }
                }
            };
            groups(hl, &L.score[hl & 511], secL, dirL, predL, grpL);
            groups(hs, &L.score[512 + (hs & 511)], secS, dirS, predS, grpS);
            // candidates: 48 bytes each, byte-equality masks (bit i: byte pos - 4 + i equals the candidate's)
            auto zb = [](uint32_t x) -> uint32_t { return ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x)) & 0x80808080u; };
            auto two = [&](uint32_t xa, uint32_t xb) -> uint32_t { return ((((zb(xa) >> 7) | (zb(xb) >> 3)) * 0x00204081u) >> 21) & 0xFFu; };
            auto emask = [&](uint32_t c) -> uint64_t {
                const Q16 c0 = ld16(s + c - 4), c1 = ld16(s + c + 12), c2 = ld16(s + c + 28);
                const uint32_t lo = two(q0.d0 ^ c0.d0, q0.d1 ^ c0.d1) | (two(q0.d2 ^ c0.d2, q0.d3 ^ c0.d3) << 8) | (two(q1.d0 ^ c1.d0, q1.d1 ^ c1.d1) << 16) | (two(q1.d2 ^ c1.d2, q1.d3 ^ c1.d3) << 24);
                return u64(lo, two(q2.d0 ^ c2.d0, q2.d1 ^ c2.d1) | (two(q2.d2 ^ c2.d2, q2.d3 ^ c2.d3) << 8));
            };
            const uint32_t cL = entL - 2, cS = entS - 2;
            const bool vL = entL > prefix_idx, vS = entS > prefix_idx;
            uint64_t EL = 0, ES = 0; bool slowL = false, slowS = false;                 // slow: a hit on a candidate too close to the start to be read with its 4 bytes in front
            if (vL) { if (cL >= 4) EL = emask(cL); else slowL = ld8(s + cL) == w8; }
            if (vS) { if (cS >= 4) ES = emask(cS); else slowS = ld4(s + cS) == uint32_t(w8); }
            const bool hitL = ((EL >> 4) & 0xFFu) == 0xFFu, hitS = ((ES >> 4) & 0xFu) == 0xFu;
            const uint32_t flL = uint32_t(__builtin_ctzll(~(EL >> 4))), flS = uint32_t(__builtin_ctzll(~(ES >> 4)));   // equal bytes from pos on (at most 44)
            const uint32_t nbL = ~uint32_t(EL) & 15u, nbS = ~uint32_t(ES) & 15u;
            const uint32_t bkL = nbL ? uint32_t(__builtin_clz(nbL)) - 28u : 4u, bkS = nbS ? uint32_t(__builtin_clz(nbS)) - 28u : 4u;
            // info: flL | flS << 6 | bkL << 12 | bkS << 15 | (catch-up may go beyond 4: long) << 18 | (short) << 19
            const uint32_t info = flL | (flS << 6) | (bkL << 12) | (bkS << 15) | (uint32_t(bkL == 4 && cL - prefix > 4) << 18) | (uint32_t(bkS == 4 && cS - prefix > 4) << 19);
            const uint32_t wpL = __shfl(q0.d1, int(predL)), wpL2 = __shfl(q0.d2, int(predL)), wpS = __shfl(q0.d1, int(predS));   // (every lane takes part)
            const bool hitAL = secL && wpL == q0.d1 && wpL2 == q0.d2, hitAS = secS && wpS == q0.d1;
            const unsigned long long m_dirty = __ballot(dirL || dirS),
                                     m_cond = __ballot((secL && (hitAL || hitL)) || (secS && (hitAS || hitS))),
                                     m_unL = __ballot(dirL || (secL && (hitAL || hitL)) || slowL),   // the long test of the lane is not known here
                                     m_Lraw = __ballot(hitL && !secL && !dirL),                      // ... is known and positive
                                     m_slow = __ballot(slowL || slowS);
            const unsigned long long m_stop = m_dirty | m_cond | m_slow;
            const unsigned long long m_L = __ballot(hitL && !secL && !dirL) & ~m_stop, m_S = __ballot(hitS && !secS && !dirS) & ~m_stop & ~m_L;
            // repcode tests of every position, for both repeat offsets (as in fast_block)
            const uint64_t own = u64(q0.d0, q0.d1);
            unsigned long long MR1 = 0, EQB1 = 0, MR2 = 0, EQB2 = 0;
            int H41 = 63, HB1 = 63, H42 = 63, HB2 = 63;
            if (rep1) { const uint64_t x = ra ^ own; MR1 = __ballot(uint32_t(x >> 32) == 0); EQB1 = __ballot((uint32_t(x >> 32) & 0xFFu) == 0); }
            if (rep2) { const uint64_t x = rb ^ own; MR2 = __ballot(uint32_t(x >> 32) == 0); EQB2 = __ballot((uint32_t(x >> 32) & 0xFFu) == 0); }
            auto read_rep = [&](uint32_t off, int lo, unsigned long long& mr, unsigned long long& eqb) {
                uint64_t x = ~0ull;
                if (lane >= lo) x = ld8(s + pos - off - 4) ^ own;
                mr = __ballot(uint32_t(x >> 32) == 0); eqb = __ballot((uint32_t(x >> 32) & 0xFFu) == 0);
            };
            // Where the walk goes from a match that starts at this lane (found with its long candidate: nxtL, with its short one: nxtS)
            // if nothing but another plain table hit follows: every repcode test of the search behind it is known from the
            // candidate's bytes and fails, the next event is a clean hit whose length and catch-up the lanes hold.  lane | kind << 8, or 64.
            uint32_t nxtL = 64, nxtS = 64;
            {
                const unsigned long long m_ev = m_L | m_S | m_stop;
                auto next_of = [&](uint32_t fl, uint64_t E) -> uint32_t {
                    const int e = lane + int(fl);
                    const unsigned long long sh = e < 64 ? m_ev >> e : 0ull;
                    const int nx = e + (sh ? __builtin_ctzll(sh) : 64);
                    if (e > 60 || nx > 61 || ((m_stop >> (nx & 63)) & 1) || nx + 4 - lane > int(kFwHeld) + 3) return 64u;
                    int tm = nx, tk = 1;
                    if (!((m_L >> nx) & 1)) {                                           // a short hit: a long hit at nx + 1 is preferred
                        if ((m_unL >> (nx + 1)) & 1) return 64u;
                        if ((m_Lraw >> (nx + 1)) & 1) { tm = nx + 1; tk = 3; } else tk = 2;
                    }
                    const uint64_t Es = E >> 4, M4 = Es & (Es >> 1) & (Es >> 2) & (Es >> 3);
                    const uint64_t T = ((2ull << (nx + 1 - lane)) - 1) & ~((1ull << (e + 1 - lane)) - 1);   // the repcode tests of that search: e+1 .. nx+1
                    if (M4 & T) return 64u;
                    return uint32_t(tm) | (uint32_t(tk) << 8) | (uint32_t(e) << 16);
                };
                uint32_t tL = 64, tS = 64;
                if (((m_Lraw >> lane) & 1) && !((m_unL >> lane) & 1) && flL < kFwHeld + 4) tL = next_of(flL, EL);
                if (hitS && !secS && !dirS && !slowS && flS < kFwHeld + 4) tS = next_of(flS, ES);
                // the target's own conditions (its catch-up may not go beyond the 4 bytes held, its length is held)
                auto target_ok = [&](uint32_t t) -> uint32_t {
                    const uint32_t tm = t & 63, tk = (t >> 8) & 7, e = t >> 16;
                    const uint32_t it = __shfl(info, int(tm));                          // (every lane takes part)
                    if ((t & 255) >= 64) return 64u;
                    const bool lng = tk != 2;
                    if (((it >> (lng ? 18 : 19)) & 1) && tm - e > 4) return 64u;
                    if ((lng ? it & 63 : (it >> 6) & 63) == kFwHeld + 4) return 64u;
                    return t & 0xFFFFu;
                };
                nxtL = target_ok(tL); nxtS = target_ok(tS);
            }
            // ---- the walk (scalar)
            int s_l = -int(sp0 - ip), cur = 0, anc = -int(sp0 - anchor);
            const int anc0 = anc;
            bool pend = owed, lastI = false;
            unsigned long long sel = 0;                                                  // lanes where a match starts
            uint32_t kindv = 0, endv = 0;                                               // per lane: 1 long, 2 short, 3 long at p+1 (found from p), 4 repcode, 5 immediate repcode-2; end of the match
            uint32_t r1 = rep1, r2 = rep2;
            int k1 = 0, k2 = 0, m1 = 0, m2 = 0; uint64_t E1 = 0, E2 = 0;
            int endk;                                                                   // 0: search goes on at cur, 1: fresh search at cur, 2: batched search at cur, 3: repcode-2 loop at cur
#define DPIN() do { cur = Ui(cur); s_l = Ui(s_l); anc = Ui(anc); r1 = U(r1); r2 = U(r2); k1 = Ui(k1); k2 = Ui(k2); m1 = Ui(m1); m2 = Ui(m2); E1 = U64(E1); E2 = U64(E2); \
                    H41 = Ui(H41); HB1 = Ui(HB1); H42 = Ui(H42); HB2 = Ui(HB2); pend = Ui(int(pend)) != 0; lastI = Ui(int(lastI)) != 0; sel = U64(sel); MR1 = U64(MR1); EQB1 = U64(EQB1); MR2 = U64(MR2); EQB2 = U64(EQB2); } while (0)
            DPIN();
            for (;;) {
                if (k1) { EQB1 = ((E1 >> 4) & ((1ull << (kFwHeld + 4)) - 1)) << m1; MR1 = EQB1 & (EQB1 >> 1) & (EQB1 >> 2) & (EQB1 >> 3); HB1 = min(63, m1 + int(kFwHeld) + 3); H41 = HB1 - 3; k1 = 0; }
                if (k2) { EQB2 = ((E2 >> 4) & ((1ull << (kFwHeld + 4)) - 1)) << m2; MR2 = EQB2 & (EQB2 >> 1) & (EQB2 >> 2) & (EQB2 >> 3); HB2 = min(63, m2 + int(kFwHeld) + 3); H42 = HB2 - 3; k2 = 0; }
                if (pend) {
                    if (cur >= 62) { endk = 1; break; }
                    if (r2) {
                        if (cur > H42) { read_rep(r2, cur, MR2, EQB2); H42 = HB2 = 63; }
                        if ((MR2 >> cur) & 1) {                                         // an immediate repcode-2 match (zstd_double_fast.c:274-290): offsets swap, only its first position enters the tables
                            const int q = cur;
                            const unsigned long long t = q + 4 < 64 ? ~(EQB2 >> (q + 4)) : 1ull;
                            const int fwv = t ? __builtin_ctzll(t) : 64;
                            if (q + 4 + fwv > HB2 && HB2 < 63) { read_rep(r2, cur, MR2, EQB2); H42 = HB2 = 63; { DPIN(); continue; } }
                            if (q + 4 + fwv >= 64) { endk = 3; break; }                 // it runs to the end of the window: the serial loop takes it
                            const int e = q + 4 + fwv;
                            sel |= 1ull << q; lastI = true;
                            if (lane == q) { kindv = 5; endv = uint32_t(e); }
                            { const uint32_t tr = r1; r1 = r2; r2 = tr; }
                            { unsigned long long tq; tq = MR1; MR1 = MR2; MR2 = tq; tq = EQB1; EQB1 = EQB2; EQB2 = tq; }
                            { int ti; ti = H41; H41 = H42; H42 = ti; ti = HB1; HB1 = HB2; HB2 = ti; }
                            anc = s_l = cur = e;
                            { DPIN(); continue; }
                        }
                    }
                    pend = false;
                }
                const int pmax = 61;                                                    // last position examined here: its refills (p + 2) stay inside the window
                if (cur > pmax) { endk = cur == s_l ? 1 : 0; break; }
                const unsigned long long from = ~0ull << cur;
                const unsigned long long k4 = H41 >= 63 ? ~0ull : (2ull << H41) - 1;
                const unsigned long long evt = (m_L | m_S | m_stop) & from, evr = MR1 & (from << 1) & k4, unk = (from << 1) & ~k4;
                const int lt = evt ? __builtin_ctzll(evt) : 64, lr = evr ? __builtin_ctzll(evr) : 999, lu = unk ? __builtin_ctzll(unk) : 999;
                if (r1 && lr > lt + 1 && lu <= lt + 1 && lu - 1 <= pmax) { read_rep(r1, cur, MR1, EQB1); H41 = HB1 = 63; { DPIN(); continue; } }
                if (lr <= lt + 1) {                                                     // the repcode test at lr = p + 1 comes before the table tests of p
                    const int q = lr, p = q - 1;
                    if (p > pmax) { cur = pmax + 1; endk = 0; break; }
                    const unsigned long long t = q + 4 < 64 ? ~(EQB1 >> (q + 4)) : 1ull;
                    const int fwv = t ? __builtin_ctzll(t) : 64;
                    if (q + 4 + fwv > HB1 && HB1 < 63) { read_rep(r1, cur, MR1, EQB1); H41 = HB1 = 63; { DPIN(); continue; } }
                    if (q + 4 + fwv >= 64) {                                            // runs to the end of the window
                        cur = p;
                        if (p > 0) endk = p == s_l ? 1 : 0; else endk = 2;
                        break;
                    }
                    const int e = q + 4 + fwv;
                    sel |= 1ull << q; lastI = false; ZCNT(1);
                    if (lane == q) { kindv = 4; endv = uint32_t(e); }
                    anc = s_l = cur = e; pend = true;
                    { DPIN(); continue; }
                }
                if (lt >= 64 || lt > pmax) { cur = pmax + 1; endk = 0; break; }
                const int p = lt;
                if ((m_stop >> p) & 1) { cur = p; endk = ((m_dirty >> p) & 1) ? (p == s_l ? 1 : 0) : 2; break; }
                int m, kd;
                if ((m_L >> p) & 1) { m = p; kd = 1; }
                else {                                                                  // short hit at p: a long hit at p + 1 is preferred
                    if ((m_unL >> (p + 1)) & 1) { cur = p; endk = 2; break; }
                    if ((m_Lraw >> (p + 1)) & 1) { m = p + 1; kd = 3; }
                    else { m = p; kd = 2; }
                }
                const uint32_t inf = rl(info, uint32_t(m));
                const bool lng = kd != 2;
                if (((inf >> (lng ? 18 : 19)) & 1) && m - anc > 4) { cur = p; endk = 2; break; }     // the catch-up goes on in memory
                const uint32_t cm = lng ? rl(cL, uint32_t(m)) : rl(cS, uint32_t(m));
                uint32_t fl = lng ? inf & 63 : (inf >> 6) & 63;
                if (fl == kFwHeld + 4) fl += count_fwd(s, sp0 + uint32_t(m) + fl, cm + fl, end, lane);
                const int e = m + int(fl);
                sel |= 1ull << m; lastI = false; ZCNT(2);
                if (lane == m) { kindv = uint32_t(kd); endv = uint32_t(e); }
                // repcode-2 test behind this match (offset: the previous repeat offset), as far as it is known here
                bool bad0 = false;
                if (r1) bad0 = e > H41 || ((MR1 >> (e & 63)) & 1);
                r2 = r1; r1 = sp0 + uint32_t(m) - cm; MR2 = MR1; EQB2 = EQB1; H42 = H41; HB2 = HB1; k2 = 0;
                k1 = 1; m1 = m; E1 = lng ? rl64(EL, uint32_t(m)) : rl64(ES, uint32_t(m));
                anc = s_l = cur = e; pend = true;
                // ---- the chain of plain table hits behind it: one readlane per sequence
                unsigned long long selc = 0, chS = 0, ch3 = 0;                          // chained lanes; those found with their short candidate; those found from the position before
                if (fl < kFwHeld + 4)
                    for (int t = m, kk = kd;;) {
                        const uint32_t nx = kk == 2 ? rl(nxtS, uint32_t(t)) : rl(nxtL, uint32_t(t));
                        if ((nx & 255) >= 64) break;
                        t = int(nx & 63); kk = int(nx >> 8);
                        selc |= 1ull << t;
                        if (kk == 2) chS |= 1ull << t; else if (kk == 3) ch3 |= 1ull << t;
                    }
                if (selc) {
                    // every link took the repcode-2 test behind its predecessor p for granted (offset: that of p's predecessor pp,
                    // whose candidate bytes know it up to 40 bytes behind pp): check them all at once, cut at the first that fails
                    const unsigned long long C = selc | (1ull << m);
                    const bool usesS = ((chS >> lane) & 1) || (lane == m && kd == 2);
                    const uint64_t Ech = usesS ? ES : EL;
                    const uint32_t ech = uint32_t(lane) + (usesS ? flS : flL);
                    const unsigned long long bl = C & ((1ull << lane) - 1);
                    const int P = bl ? 63 - __builtin_clzll(bl) : 0;
                    const uint32_t packP = __shfl(uint32_t(P) | (uint32_t(bl != 0) << 8), P);
                    const int PP = int(packP & 63);
                    const int eP = int(__shfl(ech, P));
                    const uint64_t Epp = u64(__shfl(uint32_t(Ech), PP), __shfl(uint32_t(Ech >> 32), PP));
                    const int d = eP - PP;
                    const bool bad = ((selc >> lane) & 1) && (((packP >> 8) & 1) ? (d > int(kFwHeld) || ((Epp >> ((d + 4) & 63)) & 15u) == 15u) : bad0);
                    const unsigned long long badm = __ballot(bad);
                    if (badm) selc &= (1ull << __builtin_ctzll(badm)) - 1;
                    ZADD(10, uint32_t(__builtin_popcountll(selc)));
                    if (selc) {
                        const int Lst = 63 - __builtin_clzll(selc);
                        const unsigned long long bL = (selc | (1ull << m)) & ((1ull << Lst) - 1);
                        const int Lp = 63 - __builtin_clzll(bL);
                        sel |= selc;
                        if ((selc >> lane) & 1) { kindv = ((chS >> lane) & 1) ? 2u : ((ch3 >> lane) & 1) ? 3u : 1u; endv = ech; }
                        const uint32_t candch = usesS ? cS : cL;
                        r2 = sp0 + uint32_t(Lp) - rl(candch, uint32_t(Lp)); r1 = sp0 + uint32_t(Lst) - rl(candch, uint32_t(Lst));
                        k2 = 1; m2 = Lp; E2 = rl64(Ech, uint32_t(Lp)); m1 = Lst; E1 = rl64(Ech, uint32_t(Lst));
                        anc = s_l = cur = int(rl(ech, uint32_t(Lst)));
                    }
                }
                { DPIN(); }
            }
#undef DPIN
            // ---- behind the walk: sequences and the writes of both tables, all lanes at once
            {
                const bool isC = (sel >> lane) & 1;
                const uint32_t nbelow = __builtin_amdgcn_mbcnt_hi(uint32_t(sel >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(sel), 0));
                const unsigned long long below = sel & ((1ull << lane) - 1);
                const int P = below ? 63 - __builtin_clzll(below) : 0;
                const bool tail = isC && int64_t(sp0) + int64_t(endv) > ilimit;       // the match ends behind ilimit: no refills follow it (zstd_double_fast.c:262)
                const uint32_t endP = __shfl(endv, P), kindP = __shfl(kindv | (uint32_t(tail) << 3), P);
                const int ancl = below ? int(endP) : anc0;
                uint32_t ll = 0;
                if (isC) {
                    const bool lng = kindv == 1 || kindv == 3;
                    const uint32_t cand = lng ? cL : cS;
                    const uint32_t b = kindv >= 4 ? 0u : min(lng ? bkL : bkS, min(uint32_t(lane - ancl), cand - prefix));
                    ll = uint32_t(lane - ancl) - b;
                    const uint32_t at = S.nseq + nbelow;
                    S.ll[at] = ll; S.ml[at] = (endv - uint32_t(lane)) + b - 3; S.off[at] = kindv >= 4 ? 1u : pos - cand + 3;
                }
                S.nseq += uint32_t(__builtin_popcountll(sel));
                S.nlit += rl(scan_add(ll), 63);
                // table writes (zstd_double_fast.c:140,227-229,262-270): every position examined enters both tables; of a match
                // starting at lane P (kind K) and ending at e: long P+1 (the hl1 write; K = 3: P itself) and curr+2, e-2; short curr+2, e-1
                const int Pl = isC ? lane : P;
                const bool any = isC || below != 0;
                const int eP = isC ? int(endv) : int(endP);
                const uint32_t kPt = isC ? (kindv | (uint32_t(tail) << 3)) : kindP, kP = kPt & 7;
                const bool fills = !(kPt & 8);
                const int d = lane - Pl;
                const bool probe = !any || lane >= eP;
                const bool inL = (kP <= 2 && (d <= 1 || (d == 2 && fills))) || (kP == 3 && (d == 0 || (d == 1 && fills))) || (kP == 4 && d == 1 && fills) || (kP == 5 && d == 0) || (kP != 5 && fills && lane == eP - 2);
                const bool inS = (kP <= 2 && (d == 0 || (d == 2 && fills))) || ((kP == 3 || kP == 4) && d == 1 && fills) || (kP == 5 && d == 0) || (kP != 5 && fills && lane == eP - 1);
                const bool visL = lane < cur && (probe || inL), visS = lane < cur && (probe || inS);
                const unsigned long long vmL = __ballot(visL), vmS = __ballot(visS);
                if (visL && !(grpL & vmL & ~((2ull << lane) - 1))) tl[hl] = pos + 2;
                if (visS && !(grpS & vmS & ~((2ull << lane) - 1))) ts[hs] = pos + 2;
                const unsigned long long fmL = __ballot(visL && hl == hfL), fmS = __ballot(visS && hs == hfS);
                if (owed_fill && lane == 0) { if (!fmL) tl[hfL] = sp0; if (!fmS) ts[hfS] = sp0 + 1; }
            }
            anchor = uint32_t(int(sp0) + anc); ip = uint32_t(int(sp0) + s_l); sp = sp0 + uint32_t(cur);
            rep1 = r1; rep2 = r2;
            owed = (endk == 1 && pend) || endk == 3;
            owed_fill = owed && (sel ? !lastI : owed_fill);                             // (no match in this window: what was owed on the way in)
            gen_tail = endk == 3; gen_search = endk == 2;
            ZCNT(5 + endk);
            anchor = U(anchor); ip = U(ip); sp = U(sp); rep1 = U(rep1); rep2 = U(rep2);
            continue;
        }
        gen_search = false;
        if (sp == ip && int64_t(ip) + 1 > ilimit) break;
        uint32_t IP = sp, SJ = 1, NS = ip + 256, width = 16;
        int ev_kind = 0;                                         // 1 rep at ip+1, 2 long at ip, 3 long at ip1, 4 short at ip, 5 end of block
        uint32_t ev_ip = 0, ev_s = 0, ev_idx = 0, ev_hl1 = 0;
        for (;;) {
            uint32_t p = IP, sj = SJ, ns = NS, np = 0, nsj = 0, nns = 0;
            if (IP + SJ * width < NS) p = IP + SJ * uint32_t(lane);
            else
                for (uint32_t i = 0; i < width; i++) {
                    np = p + sj; nsj = sj; nns = ns;
                    if (np >= ns) { nsj = sj + 1; nns = ns + 256; }
                    if (uint32_t(lane) > i) { p = np; sj = nsj; ns = nns; }
                }
            np = p + sj; nsj = sj; nns = ns;                     // successor position and its step
            if (np >= ns) { nsj = sj + 1; nns = ns + 256; }
            const bool act = uint32_t(lane) < width;
            const bool inb = int64_t(np) <= ilimit;               // this position is reached only if its own ip1 fits
            const bool term = int64_t(np) + int64_t(nsj) > ilimit;   // its iteration ends the block when nothing matches
            const uint32_t rp = inb ? p : start, rp1 = inb ? np : start;
            const uint64_t w0 = ld8(s + rp), w1 = ld8(s + rp1);
            const uint32_t r_rep = ld4(s + rp + 1 - (inb ? rep1 : 0));
            const uint32_t hl0 = HL(w0), hl1 = HL(w1), hs0 = zhash(w0, hs_log, mls);
            const uint32_t cur = p + 2;
            uint32_t il0 = 0, il1 = 0, is0 = 0; bool shared = false;
            if (act && inb) {
                il0 = tld(tl, hl0); is0 = tld(ts, hs0); il1 = (hl1 == hl0) ? cur : tld(tl, hl1);
                uint32_t* const a = &L.score[hl0 & 511]; uint32_t* const b = &L.score[512 + (hs0 & 511)];
                atomicMin(a, uint32_t(lane)); atomicMin(b, uint32_t(lane));
                shared = (*a != uint32_t(lane)) || (*b != uint32_t(lane)) || (L.score[hl1 & 511] < uint32_t(lane));
                *a = 0xFFFFFFFFu; *b = 0xFFFFFFFFu;
            }
            const bool okl0 = act && inb && il0 > prefix_idx, oks0 = act && inb && is0 > prefix_idx, okl1 = act && inb && il1 > prefix_idx;
            const uint64_t cl0 = ld8(s + (okl0 ? il0 - 2 : 0)), cl1 = ld8(s + (okl1 ? il1 - 2 : 0));
            const uint32_t cs0 = ld4(s + (oks0 ? is0 - 2 : 0));
            int kind = 0;
            if (act && inb) {
                if ((rep1 > 0) & (r_rep == uint32_t(w0 >> 8))) kind = 1;
                else if (okl0 && cl0 == w0) kind = 2;
                else if (oks0 && cs0 == uint32_t(w0)) kind = (okl1 && cl1 == w1) ? 3 : 4;
                else if (term) kind = 5;
            }
            const unsigned long long cutm = __ballot(act && (shared || !inb)) & ~1ull;
            const int cut = cutm ? __builtin_ctzll(cutm) : int(width);
            const unsigned long long evm = __ballot(kind != 0) & ((cut >= 64) ? ~0ull : ((1ull << cut) - 1));
            if (evm) {
                const int J = __builtin_ctzll(evm);
                if (lane <= J) { tl[hl0] = cur; ts[hs0] = cur; }
                ev_kind = int(rl(uint32_t(kind), J)); ev_ip = rl(p, J); ev_s = rl(sj, J); ev_hl1 = rl(hl1, J);
                ev_idx = ev_kind == 2 ? rl(il0, J) : (ev_kind == 3 ? rl(il1, J) : rl(is0, J));
                break;
            }
            if (lane < cut) { tl[hl0] = cur; ts[hs0] = cur; }
            if (cut < int(width)) { IP = rl(p, cut); SJ = rl(sj, cut); NS = rl(ns, cut); }
            else { IP = rl(np, width - 1); SJ = rl(nsj, width - 1); NS = rl(nns, width - 1); }
            width = min(64u, width * 2);
        }
        if (ev_kind == 5) break;                                 // _cleanup
        ZCNT(3);
        const uint32_t cur0 = ev_ip + 2;                          // index of the probed position (`curr`)
        uint32_t mlen, off_base;
        if (ev_kind == 1) {
            ip = ev_ip + 1;
            mlen = count_fwd(s, ip + 4, ip + 4 - rep1, end, lane) + 4;
            off_base = 1;
        } else {
            const uint32_t ip1 = ev_ip + ev_s;
            uint32_t match = ev_idx - 2, base_len = ev_kind == 4 ? 4u : 8u;
            ip = ev_kind == 3 ? ip1 : ev_ip;
            const uint32_t offset = ip - match;
            {
                // forward count and catch-up in ONE round trip (neither depends on the other's result)
                const uint32_t fa = ip + base_len, fb = match + base_len;
                const bool wide = fa + 1024 <= end;
                const uint32_t room = min(ip - anchor, match - prefix);
                const bool bl = uint32_t(lane) < room;
                const uint32_t b_i = bl ? uint32_t(s[ip - 1 - lane]) : 0u, b_m = bl ? uint32_t(s[match - 1 - lane]) : 1u;
                U16B x = {0, 0}, y = {0, 0}; uint32_t f_i = 0, f_m = 1;
                if (wide) { x = *reinterpret_cast<const U16B*>(s + fa + 16 * lane); y = *reinterpret_cast<const U16B*>(s + fb + 16 * lane); }
                else if (fa + lane < end) { f_i = s[fa + lane]; f_m = s[fb + lane]; }
                if (wide) {
                    const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
                    const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
                    const unsigned long long bad = __ballot(eq < 16);
                    if (bad) { const int l = __builtin_ctzll(bad); mlen = base_len + 16 * l + rl(eq, l); }
                    else mlen = base_len + 1024 + count_fwd(s, fa + 1024, fb + 1024, end, lane);
                } else {
                    const unsigned long long bad = ~__ballot(fa + lane < end && f_i == f_m);
                    if (bad) mlen = base_len + uint32_t(__builtin_ctzll(bad));
                    else mlen = base_len + 64 + count_fwd(s, fa + 64, fb + 64, end, lane);
                }
                const unsigned long long badb = ~__ballot(bl && b_i == b_m);
                uint32_t k = badb ? uint32_t(__builtin_ctzll(badb)) : 64u;
                ip -= k; match -= k; mlen += k;
                while (k == 64) {                                // catch up further, 64 bytes per step
                    const uint32_t room2 = min(ip - anchor, match - prefix);
                    const bool same = uint32_t(lane) < room2 && s[ip - 1 - lane] == s[match - 1 - lane];
                    const unsigned long long bad2 = ~__ballot(same);
                    k = bad2 ? uint32_t(__builtin_ctzll(bad2)) : 64u;
                    ip -= k; match -= k; mlen += k;
                }
            }
            rep2 = rep1; rep1 = offset; off_base = offset + 3;
            if (ev_s < 4 && lane == 0) tl[ev_hl1] = ip1 + 2;
        }
        store_seq(S, s, anchor, ip - anchor, off_base, mlen, lane);
        ip += mlen; anchor = ip;
        after_match(cur0, true, true);
        sp = ip;
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return end - anchor;
}


// ------------------------------------------------------------------------------------------------ lazy parsers (level 6)
// ZSTD_compressBlock_lazy_generic (compress/zstd_lazy.c:1486-1737, noDict) with the row-hash match finder
// (ZSTD_RowFindBestMatch :1139-1250, ZSTD_row_update_internal :915-946) or, for windows <= 2^14, hash chains
// (ZSTD_HcFindBestMatch :649-730).  The parse is serial by definition (every search sees the rows as the previous
// insertions left them); the wave parallelises inside a search: a row's 16-64 tags are compared in one step, all
// candidates are measured at once (one per lane), skipped positions enter their rows in conflict-free rounds.
// The reference's 8-entry hash cache only prefetches: the cached value always equals the hash of its position.
struct LazyState {
    uint32_t* tab; uint32_t* chain; uint8_t* tags; uint32_t ntu, low_limit, dict_limit;
    // row of the NEXT position, read ahead while this position's candidates are compared (row_search)
    uint32_t pf_ip, pf_hash, pf_head, pf_tg, pf_e;
    // searches computed ahead (row_batch): the results for positions q_pos, q_pos + 1 as the rows will stand when the walk gets there,
    // and what each needs to enter its row then {row, tag, slot}
    uint32_t q_pos, q_n, q_ml[2], q_ofb[2], q_rel[2], q_tag[2], q_slot[2];
#ifdef ZL_PROF
    unsigned long long pc[20];        // 0 searches from the read-ahead, 1 usual without it, 2 general; 3-5 their cycles; 6 extensions in best_candidate; 7 catch-up insertions; 8 sequences; 9 repcode-after loops
#endif
};
#ifdef ZL_PROF
#define ZLC(i, n) (Z.pc[i] += (n))
#define ZLT0() const unsigned long long zl_t0 = __builtin_readcyclecounter()
#define ZLT(i) (Z.pc[i] += __builtin_readcyclecounter() - zl_t0)
#else
#define ZLC(i, n) do {} while (0)
#define ZLT0() do {} while (0)
#define ZLT(i) do {} while (0)
#endif

__device__ __forceinline__ uint32_t lz_low(const LazyState& Z, const Params& P, uint32_t curr)
{ const uint32_t md = 1u << P.wlog; return curr - Z.low_limit > md ? curr - md : Z.low_limit; }

// common prefix length of s[a..] and s[b..] (a > b), a stops at lim; per lane, for short candidates: up to 32 bytes
__device__ __forceinline__ uint32_t lane_count32(const uint8_t* s, uint32_t a, uint32_t b, uint32_t lim, uint32_t n_total)
{
    const uint32_t room = lim - a;
    if (a + 32 <= n_total) {
        const U16B x0 = *reinterpret_cast<const U16B*>(s + a), y0 = *reinterpret_cast<const U16B*>(s + b);
        const U16B x1 = *reinterpret_cast<const U16B*>(s + a + 16), y1 = *reinterpret_cast<const U16B*>(s + b + 16);
        const uint64_t d0 = x0.a ^ y0.a, d1 = x0.b ^ y0.b, d2 = x1.a ^ y1.a, d3 = x1.b ^ y1.b;
        const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3)
                          : d2 ? 16u + uint32_t(__builtin_ctzll(d2) >> 3) : d3 ? 24u + uint32_t(__builtin_ctzll(d3) >> 3) : 32u;
        return min(eq, room);
    }
    uint32_t k = 0;
    while (k < 32 && k < room && s[a + k] == s[b + k]) k++;
    return k;
}

// best of the candidate list held one per lane (cand != 0): longest, first in list order on ties; returns ml (3 = none)
__device__ __forceinline__ uint32_t best_candidate(const uint8_t* s, uint32_t ip, uint32_t end, uint32_t n_total, uint32_t cand, bool has,
                                                   uint32_t order, uint32_t curr, uint32_t& ofb, int lane)
{
    uint32_t cnt = has ? lane_count32(s, ip, cand - 2, end, n_total) : 0u;
    for (unsigned long long todo = __ballot(has && cnt == 32 && ip + 32 < end); todo; todo &= todo - 1) {
        const int l = __builtin_ctzll(todo);
        const uint32_t extra = count_fwd(s, ip + 32, rl(cand, l) - 2 + 32, end, lane);
        if (lane == l) cnt += extra;
    }
    uint32_t key = (has && cnt > 3) ? ((cnt << 6) | (63u - order)) : 0u;
    key = wave_max(key);
    if (!key) return 3;
    const unsigned long long who = __ballot(has && ((cnt << 6) | (63u - order)) == key);
    ofb = curr - rl(cand, uint32_t(__builtin_ctzll(who))) + 3;
    return key >> 6;
}

// rows: positions [from, to) enter their rows in index order (ZSTD_row_update_internalImpl)
__device__ __forceinline__ void row_insert_range(ZLds& L, LazyState& Z, const Params& P, const uint8_t* s, uint32_t from, uint32_t to, int lane)
{
    const uint32_t rowlog = min(max(P.slog, 4u), 6u), mask = (1u << rowlog) - 1, hbits = P.hlog - rowlog + 8, mls = min(max(P.mml, 4u), 6u);
    for (uint32_t base = from; base < to; base += 64) {
        const uint32_t idx = base + lane;
        bool todo = idx < to;
        const uint32_t hash = todo ? zhash(ld8(s + idx - 2), hbits, mls) : 0u;
        const uint32_t rel = (hash >> 8) << rowlog;
        uint32_t* const sc = &L.score[(hash >> 8) & 1023];
        while (__ballot(todo)) {                                   // same-row insertions keep their order: lowest lane first
            if (todo) atomicMin(sc, uint32_t(lane));
            const bool mine = todo && *sc == uint32_t(lane);
            if (mine) {
                uint8_t* const tag_row = Z.tags + 2 * size_t(rel);
                const uint32_t pos = (uint32_t(tag_row[0]) - 1u) & mask;
                tag_row[0] = uint8_t(pos); tag_row[16 + pos] = uint8_t(hash);
                Z.tab[rel + pos] = idx;
                *sc = 0xFFFFFFFFu;
                todo = false;
            }
        }
    }
}

// K consecutive searches in ONE pair of trips to memory (the lazy walk searches ip, ip + 1, ip + 2 one after the other whatever it
// finds, :1547-1620; each was a row read and a candidate read of its own).  Nothing may be pending (curr == ntu).  The rows of
// p .. p + K - 1 are read together as they stand; what position j < k will have done to the row of position k by the time the walk
// searches k - its insertion, when both hash to the same row - is applied in registers; the candidates of all K are compared in one
// trip.  Only p enters its row here: p + 1, p + 2 are queued with their results and enter when (if) the walk asks for them, in order
// (row_search), so the table never holds a position the reference has not inserted.
template <int K>
__device__ __forceinline__ uint32_t row_batch(LazyState& Z, const Params& P, const uint8_t* s, uint32_t p, uint32_t end, uint32_t& ofb0, int lane)
{
    const uint32_t rowlog = min(max(P.slog, 4u), 6u), entries = 1u << rowlog, mask = entries - 1, mls = min(max(P.mml, 4u), 6u);
    const uint32_t attempts = 1u << min(P.slog, rowlog), hbits = P.hlog - rowlog + 8;
    const bool in_row = uint32_t(lane) < entries;
    const unsigned long long emask = entries == 64 ? ~0ull : ((1ull << entries) - 1);
    uint32_t rel[K], tag[K], headb[K], tg[K], e[K], slot[K];
#ifdef ZL_PROF
    unsigned long long bt = __builtin_readcyclecounter(), bn;
#define ZBT(i) do { bn = __builtin_readcyclecounter(); Z.pc[i] += bn - bt; bt = bn; } while (0)
#else
#define ZBT(i) do {} while (0)
#endif
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t h = zhash(ld8(s + p + k), hbits, mls);
        rel[k] = (h >> 8) << rowlog; tag[k] = h & 255u;
        const uint8_t* const row = Z.tags + 2 * size_t(rel[k]);
        headb[k] = row[0];
        tg[k] = in_row ? uint32_t(row[16 + lane]) : 0u;
        e[k] = in_row ? Z.tab[rel[k] + lane] : 0u;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
        for (int j = 0; j < k; j++)
            if (rel[j] == rel[k]) { headb[k] = slot[j]; if (uint32_t(lane) == slot[j]) { tg[k] = tag[j]; e[k] = p + j + 2; } }
        slot[k] = (headb[k] - 1u) & mask;
    }
    ZBT(10);
    bool has[K]; uint32_t rank[K];
    U16B x0[K], x1[K], y0[K], y1[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t curr = p + k + 2, low = lz_low(Z, P, curr);
        const uint32_t head = headb[k] & mask, ord = (uint32_t(lane) - head) & mask;
        const bool valid = in_row && tg[k] == tag[k];
        auto rot = [&](unsigned long long m) -> unsigned long long {
            m &= emask;
            return head ? ((m >> head) | (m << (entries - head))) & emask : m;
        };
        unsigned long long vm = rot(__ballot(valid));
        const unsigned long long stop = rot(__ballot(valid && e[k] < low));
        if (stop) vm &= (1ull << __builtin_ctzll(stop)) - 1;
        rank[k] = uint32_t(__builtin_popcountll(vm & ((1ull << ord) - 1)));
        has[k] = in_row && ((vm >> ord) & 1) && rank[k] < attempts;
        const uint32_t a = p + k, b = has[k] ? e[k] - 2 : a;
        x0[k] = *reinterpret_cast<const U16B*>(s + a); x1[k] = *reinterpret_cast<const U16B*>(s + a + 16);
        y0[k] = *reinterpret_cast<const U16B*>(s + b); y1[k] = *reinterpret_cast<const U16B*>(s + b + 16);
    }
    ZBT(11);
    uint32_t ml[K], of[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t a = p + k, curr = a + 2;
        const uint64_t d0 = x0[k].a ^ y0[k].a, d1 = x0[k].b ^ y0[k].b, d2 = x1[k].a ^ y1[k].a, d3 = x1[k].b ^ y1[k].b;
        const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3)
                          : d2 ? 16u + uint32_t(__builtin_ctzll(d2) >> 3) : d3 ? 24u + uint32_t(__builtin_ctzll(d3) >> 3) : 32u;
        uint32_t cnt = has[k] ? min(eq, end - a) : 0u;
        ZBT(12);
        for (unsigned long long todo = __ballot(has[k] && cnt == 32 && a + 32 < end); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const uint32_t extra = count_fwd(s, a + 32, rl(e[k], l) - 2 + 32, end, lane);
            if (lane == l) cnt += extra;
            ZLC(14, 1);
        }
        ZBT(13);
        const uint32_t mine = (has[k] && cnt > 3) ? ((cnt << 6) | (63u - rank[k])) : 0u;
        const uint32_t key = wave_max(mine);
        ml[k] = 3; of[k] = 999999999u;
        if (key) {
            const unsigned long long who = __ballot(mine == key);
            of[k] = curr - rl(e[k], uint32_t(__builtin_ctzll(who))) + 3; ml[k] = key >> 6;
        }
    }
    ZBT(15);
    if (lane == 0) {                                                    // p itself goes in (:1229-1234)
        uint8_t* const row = Z.tags + 2 * size_t(rel[0]);
        row[0] = uint8_t(slot[0]); row[16 + slot[0]] = uint8_t(tag[0]); Z.tab[rel[0] + slot[0]] = p + 2;
    }
    Z.ntu = p + 3;
    Z.q_pos = p + 1; Z.q_n = K - 1;
#pragma unroll
    for (int k = 1; k < K; k++) { Z.q_ml[k - 1] = ml[k]; Z.q_ofb[k - 1] = of[k]; Z.q_rel[k - 1] = rel[k]; Z.q_tag[k - 1] = tag[k]; Z.q_slot[k - 1] = slot[k]; }
    if (ml[0] > 3) ofb0 = of[0];
    return ml[0];
}

__device__ __forceinline__ uint32_t row_search(ZLds& L, LazyState& Z, const Params& P, const uint8_t* s, uint32_t ip, uint32_t end,
                                               uint32_t n_total, uint32_t& ofb, int lane)
{
    const uint32_t curr = ip + 2, low = lz_low(Z, P, curr);
    const uint32_t rowlog = min(max(P.slog, 4u), 6u), entries = 1u << rowlog, mask = entries - 1, mls = min(max(P.mml, 4u), 6u);
    const uint32_t attempts = 1u << min(P.slog, rowlog);
    ZLT0();
#ifndef FOURMC_ZLAZY_NOBATCH
    if (Z.q_n && Z.q_pos == ip && curr == Z.ntu) {                     // computed ahead by row_batch: the position enters its row now
        if (lane == 0) {
            uint8_t* const row = Z.tags + 2 * size_t(Z.q_rel[0]);
            row[0] = uint8_t(Z.q_slot[0]); row[16 + Z.q_slot[0]] = uint8_t(Z.q_tag[0]); Z.tab[Z.q_rel[0] + Z.q_slot[0]] = curr;
        }
        const uint32_t ml = Z.q_ml[0];
        if (ml > 3) ofb = Z.q_ofb[0];
        Z.q_ml[0] = Z.q_ml[1]; Z.q_ofb[0] = Z.q_ofb[1]; Z.q_rel[0] = Z.q_rel[1]; Z.q_tag[0] = Z.q_tag[1]; Z.q_slot[0] = Z.q_slot[1];
        Z.q_pos++; Z.q_n--; Z.ntu = curr + 1;
        ZLC(0, 1); ZLT(3);
        return ml;
    }
    Z.q_n = 0;
    const bool batch_ok = ip + 48 <= n_total;
    if (curr == Z.ntu && batch_ok) {
        Z.pf_ip = 0xFFFFFFFFu;
        const uint32_t r_ = P.strat >= 5 ? row_batch<3>(Z, P, s, ip, end, ofb, lane) : row_batch<2>(Z, P, s, ip, end, ofb, lane);
        ZLC(1, 1); ZLT(4);
        return r_;
    }
#else
    const bool batch_ok = false;
#endif
    if (curr - Z.ntu <= 1) {
        // The usual step of the lazy walk: nothing, or exactly one position (the previous one), is still to be inserted.
        // Its row and the row of this search are read in ONE round trip - cursor words first, then both heads, the tags and the
        // entries (unrotated: lane j takes entry j, which is the ((j - head) & mask)-th newest) - instead of five
        // dependent ones; if both are the same row the insertion is applied to the registers.
        const bool ins = curr - Z.ntu == 1;
        const uint32_t idx0 = ins ? Z.ntu : curr, hbits = P.hlog - rowlog + 8;          // (no insertion: the same reads, nothing written)
        const bool in_row = uint32_t(lane) < entries;
        const bool from_pf = !ins && Z.pf_ip == ip;                      // this row was read ahead by the previous search
        const uint64_t wN = ld8(s + ip + 1);                             // next position: its row is read ahead below
        uint32_t hS, head_byte, tg, e;
        uint32_t hA = 0, relA = 0, posA = 0;
        if (from_pf) { hS = Z.pf_hash; head_byte = Z.pf_head; tg = Z.pf_tg; e = Z.pf_e; }
        else {
            const uint64_t wA = ld8(s + idx0 - 2), wS = ld8(s + ip);
            hA = zhash(wA, hbits, mls); hS = zhash(wS, hbits, mls);
            relA = (hA >> 8) << rowlog;
            const uint32_t relS0 = (hS >> 8) << rowlog;
            uint8_t* const rowA = Z.tags + 2 * size_t(relA);
            uint8_t* const rowS0 = Z.tags + 2 * size_t(relS0);
            const uint32_t headA = rowA[0];
            head_byte = rowS0[0];
            tg = in_row ? uint32_t(rowS0[16 + lane]) : 0u; e = in_row ? Z.tab[relS0 + lane] : 0u;
            posA = (headA - 1u) & mask;
            if (ins && lane == 0) { rowA[0] = uint8_t(posA); rowA[16 + posA] = uint8_t(hA); Z.tab[relA + posA] = idx0; }
            if (ins && relA == relS0) { head_byte = posA; if (uint32_t(lane) == posA) { tg = hA & 255; e = idx0; } }
        }
        const uint32_t relS = (hS >> 8) << rowlog, tag = hS & 255;
        uint8_t* const rowS = Z.tags + 2 * size_t(relS);
        // read ahead: the row of ip+1 as it stands now (this search's own insertion is applied to it below)
        const uint32_t hN = zhash(wN, hbits, mls), relN = (hN >> 8) << rowlog;
        uint8_t* const rowN = Z.tags + 2 * size_t(relN);
        uint32_t n_head = rowN[0], n_tg = in_row ? uint32_t(rowN[16 + lane]) : 0u, n_e = in_row ? Z.tab[relN + lane] : 0u;
        if (!from_pf && ins && relA == relN) { n_head = posA; if (uint32_t(lane) == posA) { n_tg = hA & 255; n_e = idx0; } }   // (read before that store landed)
        const uint32_t head = head_byte & mask;
        const uint32_t ord = (uint32_t(lane) - head) & mask;
        const bool valid = in_row && tg == tag;
        const unsigned long long emask = entries == 64 ? ~0ull : ((1ull << entries) - 1);
        auto rot = [&](unsigned long long m) -> unsigned long long {       // bit i of the result: entry (head + i) & mask
            m &= emask;
            return head ? ((m >> head) | (m << (entries - head))) & emask : m;
        };
        unsigned long long vm = rot(__ballot(valid));
        const unsigned long long stop = rot(__ballot(valid && e < low));
        if (stop) vm &= (1ull << __builtin_ctzll(stop)) - 1;
        const uint32_t rank = uint32_t(__builtin_popcountll(vm & ((1ull << ord) - 1)));
        const bool has = in_row && ((vm >> ord) & 1) && rank < attempts;
        if (lane == 0) {                                               // the current position goes in as well (:1229-1234)
            const uint32_t p0 = (head_byte - 1u) & mask;
            rowS[0] = uint8_t(p0); rowS[16 + p0] = uint8_t(tag); Z.tab[relS + p0] = curr;
        }
        {
            const uint32_t p0 = (head_byte - 1u) & mask;
            if (relN == relS) { n_head = p0; if (uint32_t(lane) == p0) { n_tg = tag; n_e = curr; } }
            Z.pf_ip = ip + 1; Z.pf_hash = hN; Z.pf_head = n_head; Z.pf_tg = n_tg; Z.pf_e = n_e;
        }
        Z.ntu = curr + 1;
        const uint32_t r_ = best_candidate(s, ip, end, n_total, e, has, rank, curr, ofb, lane);
        ZLC(from_pf ? 0 : 1, 1); ZLT(from_pf ? 3 : 4);
        return r_;
    }
    ZLC(2, 1); ZLC(7, curr - Z.ntu);
    Z.pf_ip = 0xFFFFFFFFu;                                             // the general path writes rows: nothing read ahead survives it
    {   // ZSTD_row_update_internal: catch up to curr (skipping the middle of long gaps)
        uint32_t idx = Z.ntu;
        if (curr - idx > 384) { row_insert_range(L, Z, P, s, idx, idx + 96, lane); idx = curr - 32; }
        row_insert_range(L, Z, P, s, idx, curr, lane);
        Z.ntu = curr;
    }
#ifndef FOURMC_ZLAZY_NOBATCH
    if (batch_ok) {
        const uint32_t r_ = P.strat >= 5 ? row_batch<3>(Z, P, s, ip, end, ofb, lane) : row_batch<2>(Z, P, s, ip, end, ofb, lane);
        ZLT(5);
        return r_;
    }
#endif
    const uint32_t hash = zhash(ld8(s + ip), P.hlog - rowlog + 8, mls), rel = (hash >> 8) << rowlog, tag = hash & 255;
    uint8_t* const tag_row = Z.tags + 2 * size_t(rel);
    const uint32_t head_byte = tag_row[0], head = head_byte & mask;
    // lane i looks at the i-th newest entry (the rotated match mask of the reference)
    const uint32_t pos = (head + uint32_t(lane)) & mask;
    const bool in_row = uint32_t(lane) < entries;
    const uint32_t e = in_row ? Z.tab[rel + pos] : 0u;
    const bool valid = in_row && tag_row[16 + pos] == tag;
    unsigned long long vm = __ballot(valid);
    const unsigned long long stop = __ballot(valid && e < low);
    if (stop) vm &= (1ull << __builtin_ctzll(stop)) - 1;
    const uint32_t rank = uint32_t(__builtin_popcountll(vm & ((1ull << lane) - 1)));
    const bool has = ((vm >> lane) & 1) && rank < attempts;
    if (lane == 0) {                                               // the current position goes in as well (:1229-1234)
        const uint32_t p0 = (head_byte - 1u) & mask;
        tag_row[0] = uint8_t(p0); tag_row[16 + p0] = uint8_t(tag); Z.tab[rel + p0] = curr;
    }
    Z.ntu = curr + 1;
    const uint32_t r_ = best_candidate(s, ip, end, n_total, e, has, rank, curr, ofb, lane);
    ZLT(5);
    return r_;
}

__device__ __forceinline__ uint32_t hc_search(LazyState& Z, const Params& P, const uint8_t* s, uint32_t ip, uint32_t end,
                                              uint32_t n_total, uint32_t& ofb, int lane)
{
    const uint32_t curr = ip + 2, low = lz_low(Z, P, curr), mls = min(max(P.mml, 4u), 6u);
    const uint32_t chain_size = 1u << P.clog, cmask = chain_size - 1, min_chain = curr > chain_size ? curr - chain_size : 0;
    for (uint32_t idx = Z.ntu; idx < curr; idx++) {                // ZSTD_insertAndFindFirstIndex_internal (small inputs only: serial)
        const uint32_t h = zhash(ld8(s + idx - 2), P.hlog, mls);
        const uint32_t old = Z.tab[h];
        if (lane == 0) { Z.chain[idx & cmask] = old; Z.tab[h] = idx; }
    }
    Z.ntu = curr;
    uint32_t mi = Z.tab[zhash(ld8(s + ip), P.hlog, mls)];
    const uint32_t attempts = 1u << P.slog;                        // 8 .. 256: candidates gathered one per lane, 64 at a time
    uint32_t best_ml = 3, best_ofb = 0, total = 0; bool done = false;
    while (!done && mi >= low && total < attempts) {
        uint32_t cand = 0, n = 0;
        while (mi >= low && total < attempts && n < 64) {
            if (uint32_t(lane) == n) cand = mi;
            n++; total++;
            if (mi <= min_chain) { done = true; break; }
            mi = Z.chain[mi & cmask];
        }
        uint32_t o2 = 0;
        const uint32_t ml2 = best_candidate(s, ip, end, n_total, cand, uint32_t(lane) < n, uint32_t(lane), curr, o2, lane);
        if (ml2 > best_ml) { best_ml = ml2; best_ofb = o2; }       // a later candidate wins only when strictly longer (:708-716)
    }
    if (best_ml > 3) ofb = best_ofb;
    return best_ml;
}

// Binary-tree match finder of ZSTD_btlazy2 (compress/zstd_lazy.c:20-58 ZSTD_updateDUBT, :64-150 ZSTD_insertDUBT1, :231-379
// ZSTD_DUBT_findBestMatch, :383-392 ZSTD_BtFindBestMatch; noDict): zstd level 12 for blocks of 16 KiB + 1 .. 256 KiB, i.e. only
// the short last block of a file.  A tree descent is one dependent step after the other, so this is the reference's code
// run by the whole wave in lockstep (lane 0 writes; byte comparisons use all lanes).  Z.chain is the tree: two links per index.
__device__ __forceinline__ void bt_insert1(LazyState& Z, const Params& P, const uint8_t* s, uint32_t curr, uint32_t end,
                                           uint32_t nb_compares, uint32_t bt_low, int lane)
{
    uint32_t* const bt = Z.chain;
    const uint32_t bt_mask = (1u << (P.clog - 1)) - 1, max_dist = 1u << P.wlog;
    const uint32_t window_low = curr - Z.low_limit > max_dist ? curr - max_dist : Z.low_limit;
    const uint32_t ip = curr - 2;
    uint32_t common_smaller = 0, common_larger = 0;
    uint32_t smaller = 2 * (curr & bt_mask), larger = smaller + 1;          // tree slots to fill; 0xFFFFFFFF: the dummy
    uint32_t mi = bt[smaller];
    for (; nb_compares && mi > window_low; --nb_compares) {
        const uint32_t next = 2 * (mi & bt_mask), match = mi - 2;
        uint32_t ml = min(common_smaller, common_larger);
        ml += count_fwd(s, ip + ml, match + ml, end, lane);
        if (ip + ml == end) break;                                           // equal: no way to know if smaller or larger
        if (s[match + ml] < s[ip + ml]) {
            if (smaller != 0xFFFFFFFFu) bt[smaller] = mi;
            common_smaller = ml;
            if (mi <= bt_low) { smaller = 0xFFFFFFFFu; break; }
            smaller = next + 1; mi = bt[next + 1];
        } else {
            if (larger != 0xFFFFFFFFu) bt[larger] = mi;
            common_larger = ml;
            if (mi <= bt_low) { larger = 0xFFFFFFFFu; break; }
            larger = next; mi = bt[next];
        }
    }
    if (smaller != 0xFFFFFFFFu) bt[smaller] = 0;
    if (larger != 0xFFFFFFFFu) bt[larger] = 0;
}

__device__ __forceinline__ uint32_t bt_search(LazyState& Z, const Params& P, const uint8_t* s, uint32_t ip, uint32_t end, uint32_t& ofb, int lane)
{
    uint32_t* const bt = Z.chain;
    const uint32_t curr = ip + 2, mls = min(max(P.mml, 4u), 6u), bt_mask = (1u << (P.clog - 1)) - 1;
    if (curr < Z.ntu) return 0;                                              // skipped area
    for (uint32_t idx = Z.ntu; idx < curr; idx++) {                          // ZSTD_updateDUBT: chain the new positions in, unsorted
        const uint32_t hh = zhash(ld8(s + idx - 2), P.hlog, mls);
        const uint32_t old = Z.tab[hh];
        bt[2 * (idx & bt_mask)] = old; bt[2 * (idx & bt_mask) + 1] = 1u; Z.tab[hh] = idx;
    }
    Z.ntu = curr;
    const uint32_t h = zhash(ld8(s + ip), P.hlog, mls);
    uint32_t mi = Z.tab[h];
    const uint32_t window_low = lz_low(Z, P, curr);
    const uint32_t bt_low = bt_mask >= curr ? 0u : curr - bt_mask;
    const uint32_t unsort_limit = max(bt_low, window_low);
    uint32_t nb_compares = 1u << P.slog, nb_candidates = nb_compares, previous = 0;
    while (mi > unsort_limit && bt[2 * (mi & bt_mask) + 1] == 1u && nb_candidates > 1) {   // reach the end of the unsorted candidates
        const uint32_t nxt = bt[2 * (mi & bt_mask)];
        bt[2 * (mi & bt_mask) + 1] = previous;
        previous = mi; mi = nxt;
        nb_candidates--;
    }
    if (mi > unsort_limit && bt[2 * (mi & bt_mask) + 1] == 1u) {             // nullify the last one if still unsorted
        bt[2 * (mi & bt_mask)] = 0; bt[2 * (mi & bt_mask) + 1] = 0;
    }
    mi = previous;
    while (mi) {                                                             // batch sort the stacked candidates
        const uint32_t nxt = bt[2 * (mi & bt_mask) + 1];
        bt_insert1(Z, P, s, mi, end, nb_candidates, unsort_limit, lane);
        mi = nxt; nb_candidates++;
    }
    // find the longest match, inserting curr into the tree
    uint32_t common_smaller = 0, common_larger = 0, best = 0;
    uint32_t smaller = 2 * (curr & bt_mask), larger = smaller + 1, match_end_idx = curr + 8 + 1;
    mi = Z.tab[h];
    Z.tab[h] = curr;
    for (; nb_compares && mi > window_low; --nb_compares) {
        const uint32_t next = 2 * (mi & bt_mask), match = mi - 2;
        uint32_t ml = min(common_smaller, common_larger);
        ml += count_fwd(s, ip + ml, match + ml, end, lane);
        if (ml > best) {
            if (ml > match_end_idx - mi) match_end_idx = mi + ml;
            if (4 * int(ml - best) > int(hibit(curr - mi + 1) - hibit(ofb))) { best = ml; ofb = curr - mi + 3; }
            if (ip + ml == end) break;                                       // equal: drop, to keep the tree consistent
        }
        if (s[match + ml] < s[ip + ml]) {
            if (smaller != 0xFFFFFFFFu) bt[smaller] = mi;
            common_smaller = ml;
            if (mi <= bt_low) { smaller = 0xFFFFFFFFu; break; }
            smaller = next + 1; mi = bt[next + 1];
        } else {
            if (larger != 0xFFFFFFFFu) bt[larger] = mi;
            common_larger = ml;
            if (mi <= bt_low) { larger = 0xFFFFFFFFu; break; }
            larger = next; mi = bt[next];
        }
    }
    if (smaller != 0xFFFFFFFFu) bt[smaller] = 0;
    if (larger != 0xFFFFFFFFu) bt[larger] = 0;
    Z.ntu = match_end_idx - 8;                                               // skip repetitive patterns
    return best;
}

template <bool kTree>
__device__ __forceinline__ uint32_t lazy_block(ZLds& L, SeqStore& S, LazyState& Z, const Params& P, uint32_t rep[3],
                                               const uint8_t* s, uint32_t start, uint32_t end, uint32_t n_total, int lane)
{
    const bool use_row = !kTree && P.wlog > 14;
    const uint32_t depth = P.strat >= 5 ? 2u : P.strat == 3 ? 0u : 1u;      // greedy / lazy / lazy2 (btlazy2: 2)
    Z.q_n = 0;                                                     // (searches computed ahead were measured against the previous block's end)
    const int64_t ilimit = int64_t(end) - 8 - (use_row ? 8 : 0);
    const uint32_t prefix_idx = Z.dict_limit, prefix = prefix_idx - 2;
    uint32_t ip = start, anchor = start;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    auto search = [&](uint32_t at, uint32_t& ofb) -> uint32_t {
        if constexpr (kTree) return bt_search(Z, P, s, at, end, ofb, lane);
        else return use_row ? row_search(L, Z, P, s, at, end, n_total, ofb, lane) : hc_search(Z, P, s, at, end, n_total, ofb, lane);
    };
    ip += (ip == prefix) ? 1 : 0;
    {
        const uint32_t c = ip + 2, md = 1u << P.wlog;
        const uint32_t low = c - Z.dict_limit > md ? c - md : Z.dict_limit;
        const uint32_t max_rep = c - low;
        if (rep2 > max_rep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > max_rep) { saved1 = rep1; rep1 = 0; }
    }
    while (int64_t(ip) < ilimit) {
        uint32_t ml = 0, at = ip + 1, ofb = 1;
        if ((rep1 > 0) & (ld4(s + ip + 1 - rep1) == ld4(s + ip + 1)))
            ml = count_fwd(s, ip + 1 + 4, ip + 1 + 4 - rep1, end, lane) + 4;
        if (!(depth == 0 && ml)) {                                 // greedy takes the repeat match without a search (:1549)
            uint32_t found = 999999999;
            const uint32_t ml2 = search(ip, found);
            if (ml2 > ml) { ml = ml2; at = ip; ofb = found; }
        }
        if (ml < 4) { ip += ((ip - anchor) >> 8) + 1; continue; }
        while (depth && int64_t(ip) < ilimit) {                     // depth >= 1
            ip++;
            if ((rep1 > 0) & (ld4(s + ip) == ld4(s + ip - rep1))) {
                const uint32_t mr = count_fwd(s, ip + 4, ip + 4 - rep1, end, lane) + 4;
                const int g2 = int(mr * 3), g1 = int(ml * 3 - uint32_t(hibit(ofb)) + 1);
                if (mr >= 4 && g2 > g1) { ml = mr; ofb = 1; at = ip; }
            }
            {
                uint32_t cand = 999999999;
                const uint32_t ml2 = search(ip, cand);
                const int g2 = int(ml2 * 4 - uint32_t(hibit(cand))), g1 = int(ml * 4 - uint32_t(hibit(ofb)) + 4);
                if (ml2 >= 4 && g2 > g1) { ml = ml2; ofb = cand; at = ip; continue; }
            }
            if (depth == 2 && int64_t(ip) < ilimit) {
                ip++;
                if ((rep1 > 0) & (ld4(s + ip) == ld4(s + ip - rep1))) {
                    const uint32_t mr = count_fwd(s, ip + 4, ip + 4 - rep1, end, lane) + 4;
                    const int g2 = int(mr * 4), g1 = int(ml * 4 - uint32_t(hibit(ofb)) + 1);
                    if (mr >= 4 && g2 > g1) { ml = mr; ofb = 1; at = ip; }
                }
                {
                    uint32_t cand = 999999999;
                    const uint32_t ml2 = search(ip, cand);
                    const int g2 = int(ml2 * 4 - uint32_t(hibit(cand))), g1 = int(ml * 4 - uint32_t(hibit(ofb)) + 7);
                    if (ml2 >= 4 && g2 > g1) { ml = ml2; ofb = cand; at = ip; continue; }
                }
            }
            break;
        }
        if (ofb > 3) {
            const uint32_t off = ofb - 3;
            for (;;) {                                             // catch up, 64 bytes per step
                const uint32_t room = min(at - anchor, at - off - prefix);
                const bool same = uint32_t(lane) < room && s[at - 1 - lane] == s[at - off - 1 - lane];
                const unsigned long long bad = ~__ballot(same);
                const uint32_t k = bad ? uint32_t(__builtin_ctzll(bad)) : 64u;
                at -= k; ml += k;
                if (k < 64) break;
            }
            rep2 = rep1; rep1 = off;
        }
        store_seq(S, s, anchor, at - anchor, ofb, ml, lane);
        ZLC(8, 1);
        anchor = ip = at + ml;
        while ((int64_t(ip) <= ilimit) & (rep2 > 0) && ld4(s + ip) == ld4(s + ip - rep2)) {
            ZLC(9, 1);
            const uint32_t t = rep2;
            ml = count_fwd(s, ip + 4, ip + 4 - rep2, end, lane) + 4;
            rep2 = rep1; rep1 = t;
            store_seq(S, s, anchor, 0, 1, ml, lane);
            ip += ml; anchor = ip;
        }
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return end - anchor;
}

// ------------------------------------------------------------------------------------------------ optimal parser
// ZSTD_btopt (compress/zstd_opt.c, optLevel 0; restated in oracle/zstd_enc_port.c): zstd level 12 for inputs of 16 KiB and less
// (clevels.h:118), i.e. a file's short last block.  One block, so only the first-block statistics exist (:191-226).  The
// forward pass is a chain of dependent price updates: the wave runs it in lockstep (every lane computes and stores the
// same values, byte comparisons use all lanes).  Everything lives in the block's workspace: hash table, binary tree, 3-byte hash table, price nodes
// (7 words each), match list, symbol statistics.  Prices are in 1/256 bit.
constexpr uint32_t kOptNum = 4096, kOptBit = 256;
constexpr int      kOptMaxPrice = 1 << 30;
struct OptState {
    uint32_t *lit_freq, *ll_freq, *ml_freq, *of_freq;             // 256 / 36 / 53 / 32
    uint32_t lit_sum, ll_sum, ml_sum, of_sum, lit_base, ll_base, ml_base, of_base;
    bool predef;
    uint32_t* node;                                               // [kOptNum + 2][7]: price, off, mlen, litlen, rep[3]
    uint32_t* match;                                              // [kOptNum + 2][2]: off, len
    uint32_t* hash3; uint32_t hlog3, next3;
};
// Every lane stores the same value to the same word: a lane-0-only store followed by a load in the other lanes is a
// data race in the language's eyes (the compiler may forward the old value to them), a store by all lanes is not.
#define OPT_W0(lhs, rhs) do { (lhs) = (rhs); } while (0)
__device__ __forceinline__ uint32_t opt_weight(uint32_t stat) { return uint32_t(hibit(stat + 1)) * kOptBit; }      // ZSTD_bitWeight :40-43
__device__ __forceinline__ void opt_base_prices(OptState& o)                                                       // :71-78
{ o.lit_base = opt_weight(o.lit_sum); o.ll_base = opt_weight(o.ll_sum); o.ml_base = opt_weight(o.ml_sum); o.of_base = opt_weight(o.of_sum); }
__device__ __forceinline__ uint32_t opt_literal_price(const OptState& o, uint32_t b)                               // ZSTD_rawLiteralsCost :245-269, one literal
{
    if (o.predef) return 6 * kOptBit;
    return o.lit_base - min(opt_weight(o.lit_freq[b]), o.lit_base - kOptBit);
}
__device__ __forceinline__ uint32_t opt_ll_price(const OptState& o, uint32_t ll)                                   // ZSTD_litLengthPrice :273-292
{
    if (o.predef) return opt_weight(ll);
    const uint32_t c = ll_code(ll);
    return ll_bits(c) * kOptBit + o.ll_base - opt_weight(o.ll_freq[c]);
}
__device__ __forceinline__ uint32_t opt_match_price(const OptState& o, uint32_t off_base, uint32_t mlen)           // ZSTD_getMatchPrice :300-328
{
    const uint32_t ofc = uint32_t(hibit(off_base)), mlb = mlen - 3;
    if (o.predef) return opt_weight(mlb) + (16 + ofc) * kOptBit;
    uint32_t price = ofc * kOptBit + (o.of_base - opt_weight(o.of_freq[ofc]));
    if (ofc >= 20) price += (ofc - 19) * 2 * kOptBit;
    const uint32_t c = ml_code(mlb);
    price += ml_bits(c) * kOptBit + (o.ml_base - opt_weight(o.ml_freq[c]));
    return price + kOptBit / 5;
}
__device__ __forceinline__ void opt_new_rep(uint32_t out[3], uint32_t r0, uint32_t r1, uint32_t r2, uint32_t off_base, uint32_t ll0)   // ZSTD_newRep
{
    if (off_base > 3) { r2 = r1; r1 = r0; r0 = off_base - 3; }
    else {
        const uint32_t code = off_base - 1 + ll0;
        if (code > 0) { const uint32_t v = code == 3 ? r0 - 1 : code == 1 ? r1 : r2; if (code >= 2) r2 = r1; r1 = r0; r0 = v; }
    }
    out[0] = r0; out[1] = r1; out[2] = r2;
}
__device__ __forceinline__ uint32_t opt_hash3(uint32_t v, uint32_t hlog) { return ((v << 8) * 506832829u) >> (32 - hlog); }            // ZSTD_hash3Ptr

// ZSTD_insertBt1 (:414-530): index curr enters the tree; returns how many positions to move on
__device__ __forceinline__ uint32_t opt_tree_insert(LazyState& Z, const Params& P, const uint8_t* s, uint32_t curr, uint32_t end, uint32_t target, int lane)
{
    uint32_t* const bt = Z.chain;
    const uint32_t bt_mask = (1u << (P.clog - 1)) - 1, bt_low = bt_mask >= curr ? 0u : curr - bt_mask;
    const uint32_t window_low = lz_low(Z, P, target), ip = curr - 2;
    const uint32_t h = zhash(ld4(s + ip), P.hlog, 4);
    uint32_t mi = Z.tab[h], smaller = 2 * (curr & bt_mask), larger = smaller + 1, end_idx = curr + 8 + 1, left = 1u << P.slog;
    uint32_t common_s = 0, common_l = 0, best = 8;
    OPT_W0(Z.tab[h], curr);
    for (; left && mi >= window_low; --left) {
        const uint32_t next = 2 * (mi & bt_mask), match = mi - 2;
        uint32_t ml = min(common_s, common_l);
        ml += count_fwd(s, ip + ml, match + ml, end, lane);
        if (ml > best) { best = ml; if (ml > end_idx - mi) end_idx = mi + ml; }
        if (ip + ml == end) break;
        if (s[match + ml] < s[ip + ml]) {
            if (smaller != 0xFFFFFFFFu) OPT_W0(bt[smaller], mi);
            common_s = ml;
            if (mi <= bt_low) { smaller = 0xFFFFFFFFu; break; }
            smaller = next + 1; mi = bt[next + 1];
        } else {
            if (larger != 0xFFFFFFFFu) OPT_W0(bt[larger], mi);
            common_l = ml;
            if (mi <= bt_low) { larger = 0xFFFFFFFFu; break; }
            larger = next; mi = bt[next];
        }
    }
    if (smaller != 0xFFFFFFFFu) OPT_W0(bt[smaller], 0u);
    if (larger != 0xFFFFFFFFu) OPT_W0(bt[larger], 0u);
    const uint32_t positions = best > 384 ? min(192u, best - 384) : 0u;
    return max(positions, end_idx - (curr + 8));
}

// ZSTD_btGetAllMatches (:798-816) = ZSTD_updateTree_internal (:533-552) + ZSTD_insertBtAndGetAllMatches (:559-786), noDict, mls 3:
// every match at ip longer than the ones before it, shortest first (o.match); ip enters the tree
__device__ __forceinline__ uint32_t opt_matches(LazyState& Z, const Params& P, OptState& o, const uint8_t* s, uint32_t ip, uint32_t end,
                                uint32_t r0, uint32_t r1, uint32_t r2, uint32_t ll0, int lane)
{
    const uint32_t curr = ip + 2, sufficient = min(P.tlen, kOptNum - 1);
    uint32_t* const bt = Z.chain;
    uint32_t* const out = o.match;
    const uint32_t minmatch = P.mml == 3 ? 3u : 4u;
    uint32_t n = 0, best = minmatch - 1;                                         // lengthToBeat - 1
    if (curr < Z.ntu) return 0;                                                  // skipped area
    for (uint32_t idx = Z.ntu; idx < curr; ) idx += opt_tree_insert(Z, P, s, idx, end, curr, lane);
    Z.ntu = curr;
    const uint32_t bt_mask = (1u << (P.clog - 1)) - 1, bt_low = bt_mask >= curr ? 0u : curr - bt_mask;
    const uint32_t window_low = lz_low(Z, P, curr), match_low = window_low ? window_low : 1u;
    const uint32_t word = ld4(s + ip), h = zhash(word, P.hlog, 4);
    uint32_t mi = Z.tab[h], smaller = 2 * (curr & bt_mask), larger = smaller + 1, end_idx = curr + 8 + 1, left = 1u << P.slog;
    uint32_t common_s = 0, common_l = 0;
    for (uint32_t code = ll0; code < 3 + ll0; code++) {                          // repeat offsets
        const uint32_t off = code == 3 ? r0 - 1 : code == 0 ? r0 : code == 1 ? r1 : r2;
        uint32_t len = 0;
        if (off - 1 < curr - Z.dict_limit) {                                     // 1 <= off <= distance to the prefix start
            const uint32_t other = ld4(s + ip - off);
            if (curr - off >= window_low && (minmatch == 3 ? (word << 8) == (other << 8) : word == other))      // ZSTD_readMINMATCH :369-380
                len = count_fwd(s, ip + minmatch, ip + minmatch - off, end, lane) + minmatch;
        }
        if (len > best) {
            best = len;
            OPT_W0(out[2 * n], code - ll0 + 1); OPT_W0(out[2 * n + 1], len); n++;
            if (len > sufficient || ip + len == end) return n;
        }
    }
    if (minmatch == 3 && best < 3) {                                             // 3-byte matches through their own hash table (:385-404, :659-688)
        for (uint32_t idx = o.next3; idx < curr; idx++) o.hash3[opt_hash3(ld4(s + idx - 2), o.hlog3)] = idx;   // in index order
        o.next3 = curr;
        const uint32_t i3 = o.hash3[opt_hash3(word, o.hlog3)];
        if (i3 >= match_low && curr - i3 < (1u << 18)) {
            const uint32_t len = count_fwd(s, ip, i3 - 2, end, lane);
            if (len >= 3) {
                best = len;
                OPT_W0(out[0], curr - i3 + 3); OPT_W0(out[1], len); n = 1;
                if (len > sufficient || ip + len == end) { Z.ntu = curr + 1; return 1; }
            }
        }
    }
    OPT_W0(Z.tab[h], curr);
    for (; left && mi >= match_low; --left) {
        const uint32_t next = 2 * (mi & bt_mask), match = mi - 2;
        uint32_t ml = min(common_s, common_l);
        ml += count_fwd(s, ip + ml, match + ml, end, lane);
        if (ml > best) {
            if (ml > end_idx - mi) end_idx = mi + ml;
            best = ml;
            OPT_W0(out[2 * n], curr - mi + 3); OPT_W0(out[2 * n + 1], ml); n++;
            if (ml > kOptNum || ip + ml == end) break;                           // equal to the end: no order, keep the tree consistent
        }
        if (s[match + ml] < s[ip + ml]) {
            if (smaller != 0xFFFFFFFFu) OPT_W0(bt[smaller], mi);
            common_s = ml;
            if (mi <= bt_low) { smaller = 0xFFFFFFFFu; break; }
            smaller = next + 1; mi = bt[next + 1];
        } else {
            if (larger != 0xFFFFFFFFu) OPT_W0(bt[larger], mi);
            common_l = ml;
            if (mi <= bt_low) { larger = 0xFFFFFFFFu; break; }
            larger = next; mi = bt[next];
        }
    }
    if (smaller != 0xFFFFFFFFu) OPT_W0(bt[smaller], 0u);
    if (larger != 0xFFFFFFFFu) OPT_W0(bt[larger], 0u);
    Z.ntu = end_idx - 8;                                                         // skip repetitive patterns
    return n;
}

// ZSTD_compressBlock_opt_generic (:1039-1325), optLevel 0, no dictionary, no long-distance matches
__device__ __forceinline__ uint32_t opt_block(ZLds& L, SeqStore& S, LazyState& Z, const Params& P, uint32_t rep[3], uint32_t* area,
                                                        const uint8_t* s, uint32_t start, uint32_t end, int lane)
{
    OptState o;
    o.hlog3 = min(P.wlog, 17u);                                                  // ZSTD_reset_matchState: hashLog3 = MIN(ZSTD_HASHLOG3_MAX, windowLog)
    o.hash3 = area; area += size_t(1) << o.hlog3;
    o.node = area; area += (kOptNum + 2) * 7;
    o.match = area;
    o.lit_freq = L.count; o.ll_freq = L.qstack; o.ml_freq = L.qstack + 64; o.of_freq = L.qstack + 128;      // LDS: free until the entropy stage
    uint32_t* const node = o.node;
#define ND(i, f) node[(i) * 7 + (f)]                                             /* f: 0 price, 1 off, 2 mlen, 3 litlen, 4..6 rep */
    {   // ZSTD_rescaleFreqs :123-240, first block, no dictionary
        const uint32_t n = end - start;
        o.predef = n <= 1024;
        for (uint32_t i = lane; i < (1u << o.hlog3); i += 64) o.hash3[i] = 0;
        { uint32_t largest, max_sym; hist_bytes(L, s + start, n, largest, max_sym, lane); }
        uint32_t part = 0;
        for (int i = lane; i < 256; i += 64) { const uint32_t v = 1 + (o.lit_freq[i] >> 8); o.lit_freq[i] = v; part += v; }
        o.lit_sum = rl(scan_add(part), 63);
        for (int i = lane; i < 36; i += 64) o.ll_freq[i] = i == 0 ? 4u : i == 1 ? 2u : 1u;
        for (int i = lane; i < 53; i += 64) o.ml_freq[i] = 1;
        for (int i = lane; i < 32; i += 64) o.of_freq[i] = i < 11 ? uint32_t((0x23444321126ull >> (4 * i)) & 15) : 1u;   // 6,2,1,1,2,3,4,4,4,3,2
        o.ll_sum = 4 + 2 + 34; o.ml_sum = 53; o.of_sum = 32 + 21;
        opt_base_prices(o);
    }
    const int64_t ilimit = int64_t(end) - 8;
    const uint32_t sufficient = min(P.tlen, kOptNum - 1), minmatch = P.mml == 3 ? 3u : 4u;
    uint32_t ip = start, anchor = start;
    o.next3 = Z.ntu;
    ip += (ip + 2 == Z.dict_limit) ? 1u : 0u;
    while (int64_t(ip) < ilimit) {
        uint32_t cur, last_pos = 0;
        uint32_t last_off, last_mlen, last_litlen;
        bool jump = false;
        {   // the matches at ip open a series
            const uint32_t litlen = ip - anchor, ll0 = litlen ? 0u : 1u;
            const uint32_t nb = opt_matches(Z, P, o, s, ip, end, rep[0], rep[1], rep[2], ll0, lane);
            if (!nb) { ip++; continue; }
            const uint32_t price0 = opt_ll_price(o, litlen);
            OPT_W0(ND(0, 0), price0); OPT_W0(ND(0, 2), 0u); OPT_W0(ND(0, 3), litlen);
            OPT_W0(ND(0, 4), rep[0]); OPT_W0(ND(0, 5), rep[1]); OPT_W0(ND(0, 6), rep[2]);
            const uint32_t max_len = o.match[2 * (nb - 1) + 1], max_off = o.match[2 * (nb - 1)];
            if (max_len > sufficient) {                                          // long match: taken at once
                last_litlen = litlen; last_mlen = max_len; last_off = max_off; cur = 0; jump = true;
            } else {
                const uint32_t lits_price = price0 + opt_ll_price(o, 0);
                uint32_t pos = 1;
                for (; pos < minmatch; pos++) OPT_W0(ND(pos, 0), uint32_t(kOptMaxPrice));
                for (uint32_t k = 0; k < nb; k++) {
                    const uint32_t off = o.match[2 * k], len = o.match[2 * k + 1];
                    for (; pos <= len; pos++) {
                        const uint32_t pr = lits_price + opt_match_price(o, off, pos);
                        OPT_W0(ND(pos, 0), pr); OPT_W0(ND(pos, 1), off); OPT_W0(ND(pos, 2), pos); OPT_W0(ND(pos, 3), litlen);
                    }
                }
                last_pos = pos - 1;
            }
        }
        if (!jump) {
            for (cur = 1; cur <= last_pos; cur++) {
                const uint32_t inr = ip + cur;
                {   // one more literal, if that is not dearer
                    const uint32_t pm = ND(cur - 1, 2), litlen = pm == 0 ? ND(cur - 1, 3) + 1 : 1u;
                    const int price = int(ND(cur - 1, 0)) + int(opt_literal_price(o, s[inr - 1])) + int(opt_ll_price(o, litlen)) - int(opt_ll_price(o, litlen - 1));
                    if (price <= int(ND(cur, 0))) { OPT_W0(ND(cur, 0), uint32_t(price)); OPT_W0(ND(cur, 1), 0u); OPT_W0(ND(cur, 2), 0u); OPT_W0(ND(cur, 3), litlen); }
                }
                const uint32_t c_mlen = ND(cur, 2), c_litlen = ND(cur, 3), c_off = ND(cur, 1);
                uint32_t cr[3];
                if (c_mlen != 0) { const uint32_t pv = cur - c_mlen; opt_new_rep(cr, ND(pv, 4), ND(pv, 5), ND(pv, 6), c_off, c_litlen == 0 ? 1u : 0u); }
                else { cr[0] = ND(cur - 1, 4); cr[1] = ND(cur - 1, 5); cr[2] = ND(cur - 1, 6); }
                OPT_W0(ND(cur, 4), cr[0]); OPT_W0(ND(cur, 5), cr[1]); OPT_W0(ND(cur, 6), cr[2]);
                if (int64_t(inr) > ilimit) continue;                             // the last match starts at least 8 bytes before the end
                if (cur == last_pos) break;
                const int c_price = int(ND(cur, 0));
                if (int(ND(cur + 1, 0)) <= c_price + int(kOptBit / 2)) continue; // unpromising position
                const uint32_t ll0 = c_mlen != 0 ? 1u : 0u, litlen = c_mlen == 0 ? c_litlen : 0u;
                const uint32_t base = uint32_t(c_price) + opt_ll_price(o, 0);
                const uint32_t nb = opt_matches(Z, P, o, s, inr, end, cr[0], cr[1], cr[2], ll0, lane);
                if (!nb) continue;
                const uint32_t max_len = o.match[2 * (nb - 1) + 1];
                if (max_len > sufficient || cur + max_len >= kOptNum) {
                    last_mlen = max_len; last_off = o.match[2 * (nb - 1)]; last_litlen = litlen;
                    cur -= c_mlen == 0 ? c_litlen : 0u;                          // may wrap: then it is the first sequence
                    if (cur > kOptNum) cur = 0;
                    jump = true;
                    break;
                }
                for (uint32_t k = 0; k < nb; k++) {
                    const uint32_t off = o.match[2 * k], len = o.match[2 * k + 1];
                    const uint32_t first = k ? o.match[2 * (k - 1) + 1] + 1 : minmatch;
                    for (uint32_t mlen = len; mlen >= first; mlen--) {           // downwards
                        const uint32_t pos = cur + mlen;
                        const int price = int(base) + int(opt_match_price(o, off, mlen));
                        if (pos > last_pos || price < int(ND(pos, 0))) {
                            while (last_pos < pos) { last_pos++; OPT_W0(ND(last_pos, 0), uint32_t(kOptMaxPrice)); }
                            OPT_W0(ND(pos, 0), uint32_t(price)); OPT_W0(ND(pos, 1), off); OPT_W0(ND(pos, 2), mlen); OPT_W0(ND(pos, 3), litlen);
                        } else break;                                            // optLevel 0: early abort
                    }
                }
            }
            if (!jump) {
                last_off = ND(last_pos, 1); last_mlen = ND(last_pos, 2); last_litlen = ND(last_pos, 3);
                cur = last_pos > last_litlen + last_mlen ? last_pos - (last_litlen + last_mlen) : 0u;
            }
        }
        // shortest path: the next series' repeat offsets, then the chosen arrivals walked back and emitted front to back
        {
            const uint32_t a = ND(cur, 4), b = ND(cur, 5), c = ND(cur, 6);
            if (last_mlen != 0) opt_new_rep(rep, a, b, c, last_off, last_litlen == 0 ? 1u : 0u);
            else { rep[0] = a; rep[1] = b; rep[2] = c; }
        }
        const uint32_t store_end = cur + 1;
        uint32_t store_start = store_end, seq_pos = cur;
        OPT_W0(ND(store_end, 1), last_off); OPT_W0(ND(store_end, 2), last_mlen); OPT_W0(ND(store_end, 3), last_litlen);
        while (seq_pos > 0) {
            const uint32_t f1 = ND(seq_pos, 1), f2 = ND(seq_pos, 2), f3 = ND(seq_pos, 3);
            store_start--;
            OPT_W0(ND(store_start, 1), f1); OPT_W0(ND(store_start, 2), f2); OPT_W0(ND(store_start, 3), f3);
            const uint32_t back = f3 + f2;
            seq_pos = seq_pos > back ? seq_pos - back : 0u;
        }
        for (uint32_t k = store_start; k <= store_end; k++) {
            const uint32_t off = ND(k, 1), mlen = ND(k, 2), llen = ND(k, 3);
            if (mlen == 0) { ip = anchor + llen; continue; }                     // trailing literals: the next series starts behind them
            {   // ZSTD_updateStats :332-363
                for (uint32_t u = lane; u < llen; u += 64) atomicAdd(&o.lit_freq[s[anchor + u]], 2u);
                o.lit_sum += 2 * llen;
                const uint32_t lc = ll_code(llen), oc = uint32_t(hibit(off)), mc = ml_code(mlen - 3);
                const uint32_t v0 = o.ll_freq[lc], v1 = o.of_freq[oc], v2 = o.ml_freq[mc];
                OPT_W0(o.ll_freq[lc], v0 + 1); OPT_W0(o.of_freq[oc], v1 + 1); OPT_W0(o.ml_freq[mc], v2 + 1);
                o.ll_sum++; o.of_sum++; o.ml_sum++;
            }
            store_seq(S, s, anchor, llen, off, mlen, lane);
            anchor += llen + mlen;
            ip = anchor;
        }
        opt_base_prices(o);
    }
#undef ND
    return end - anchor;
}

// ------------------------------------------------------------------------------------------------ frame
__device__ __forceinline__ Params level_params(uint32_t n, int level)
{
    Params p;
    uint32_t wlog, hlog, clog;
    {
        const int lv = level < 1 ? 1 : level > kMaxLevel ? kMaxLevel : level;     // (the launchers refuse every other level)
        const LevelRow r = kLevelRowsDev[lv - 1][(n <= 256 * 1024) + (n <= 128 * 1024) + (n <= 16 * 1024)];
        wlog = r.wlog; clog = r.clog; hlog = r.hlog; p.slog = r.slog; p.mml = r.mml; p.tlen = r.tlen; p.strat = r.strat;
    }
    const uint32_t src_log = n < 64 ? 6u : uint32_t(hibit(n - 1)) + 1;
    if (wlog > src_log) wlog = src_log;
    if (hlog > wlog + 1) hlog = wlog + 1;
    if (clog - (p.strat >= 6 ? 1u : 0u) > wlog) clog = wlog + (p.strat >= 6 ? 1u : 0u);   // ZSTD_cycleLog: a binary tree has half as many nodes
    if (wlog < 10) wlog = 10;
    p.wlog = wlog; p.hlog = hlog; p.clog = clog;
    return p;
}

__device__ __forceinline__ bool is_rle(const uint8_t* s, uint32_t n, int lane)
{
    const uint8_t v = s[0];
    bool diff = false;
    for (uint32_t i = lane; i < n; i += 64) diff |= s[i] != v;
    return __ballot(diff) == 0;
}

// ZSTD_compress(dst, cap, src, n, 1); returns the frame size or a negative ZSTD error number
// kTree: the instance for the binary-tree strategies of level 12 (btlazy2 up to 256 KiB, btopt up to 16 KiB: a file's short last
// block), compiled into a kernel of its own so that their code does not weigh on the register allocation of the fast /
// dfast / lazy paths every full block takes (with the tree finder inlined next to it, dfast ran 7 % slower)
// kFast: 1 = the level-1 kernel's instance (strategy "fast" only), 2 = the level-3 kernel's ("dfast" only), 0 = the lazy levels: the
// other match finders are not compiled into an instance, so its
// registers are allocated for the dense window alone)
template <bool kTree, int kFast>
__device__ __forceinline__ int zstd_encode_frame(ZLds& L, const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint8_t* work, int level, bool serial, int lane,
                                                 uint32_t* lds_tab = nullptr)
{
    Params P = level_params(n, level);
    if (kFast) P.strat = uint32_t(kFast);                         // (zstd_encode_one sent this block here because that is its strategy)
    // (lds_tab: the measurement build -DZ1_LDS_TABLE hands the level-1 kernel 64 KiB of LDS for the hash table of blocks whose
    // hashLog is 14 - one block per CU instead of eight; profiles/r06_encoders.md has what that costs)
    uint32_t* const tab = (lds_tab && P.hlog <= 14) ? lds_tab : reinterpret_cast<uint32_t*>(work + kStoreBytes);
    uint32_t* const tab_s = reinterpret_cast<uint32_t*>(work + kStoreBytes + (size_t(4) << max(P.hlog, 17u)));      // dfast: short-hash table
    SeqStore S;
    S.ll = reinterpret_cast<uint32_t*>(work); S.ml = S.ll + kSeqCap; S.off = S.ml + kSeqCap;
    S.llc = work + kOffCodes; S.ofc = S.llc + kCodeBytes; S.mlc = S.ofc + kCodeBytes;
    S.lit = work + kOffLit + kLitPad;
#ifdef Z1_PROF
    for (int i = 0; i < 19; i++) S.prof[i] = 0;
    S.prof[19] = __builtin_readcyclecounter();
#endif
    FseCt* const prevfse = reinterpret_cast<FseCt*>(work + kOffPrev);
    uint8_t* const tmp = work + kOffTmp;
    LazyState Z;
    Z.tab = tab; Z.ntu = 2; Z.low_limit = 2; Z.dict_limit = 2;
    Z.pf_ip = 0xFFFFFFFFu; Z.pf_hash = Z.pf_head = Z.pf_tg = Z.pf_e = 0;
    Z.q_pos = 0; Z.q_n = 0;
    for (int i = 0; i < 2; i++) Z.q_ml[i] = Z.q_ofb[i] = Z.q_rel[i] = Z.q_tag[i] = Z.q_slot[i] = 0;
#ifdef ZL_PROF
    for (int i = 0; i < 20; i++) Z.pc[i] = 0;
#endif
    Z.tags = reinterpret_cast<uint8_t*>(tab) + (size_t(4) << P.hlog);                                   // rows: tag table behind the entries
    Z.chain = tab + (size_t(1) << P.hlog);                                                              // hash chains: chain table there
    uint32_t o = 0;
    if (cap < 18) return kErrTooSmall;
    {
        const uint32_t wsize = 1u << P.wlog;
        const bool single = wsize >= n;
        const uint32_t fcs = (n >= 256) + (n >= 65536 + 256);
        uint64_t lo = 0xFD2FB528ull | (uint64_t((uint32_t(single) << 5) + (fcs << 6)) << 32), hi = 0;
        uint32_t k = 5;
        if (!single) { lo |= uint64_t((P.wlog - 10) << 3) << 40; k = 6; }
        const uint32_t fb = fcs == 0 ? (single ? 1u : 0u) : fcs == 1 ? 2u : 4u;
        const uint64_t fv = fcs == 1 ? n - 256 : n;
        lo |= fv << (8 * k); if (k == 6) hi = fv >> 16;
        if (uint32_t(lane) < k + fb) dst[lane] = uint8_t(lane < 8 ? lo >> (8 * lane) : hi >> (8 * (lane - 8)));
        o = k + fb;
    }
    if (!n) {
        if (cap - o < 4) return kErrTooSmall;
        if (lane == 0) { dst[o] = 1; dst[o + 1] = 0; dst[o + 2] = 0; }
        return int(o) + 3;
    }
    {   // hash table = 0, same-slot scoreboard = empty, literal tables = none
        uint4* t4 = reinterpret_cast<uint4*>(tab);
        const uint32_t n16 = (4u << P.hlog) / 16;
        for (uint32_t i = lane; i < n16; i += 64) t4[i] = make_uint4(0, 0, 0, 0);
        if (P.strat == 2) {
            uint4* s4 = reinterpret_cast<uint4*>(tab_s);
            const uint32_t m16 = (4u << P.clog) / 16;
            for (uint32_t i = lane; i < m16; i += 64) s4[i] = make_uint4(0, 0, 0, 0);
        }
        if (P.strat >= 3) {                                    // tag table (2 bytes per entry) or chain table (4 bytes per entry)
            uint4* s4 = reinterpret_cast<uint4*>(Z.tags);
            const uint32_t m16 = (P.strat < 6 && P.wlog > 14) ? (2u << P.hlog) / 16 : (4u << P.clog) / 16;
            for (uint32_t i = lane; i < m16; i += 64) s4[i] = make_uint4(0, 0, 0, 0);
        }
        for (int i = lane; i < 1024; i += 64) L.score[i] = 0xFFFFFFFFu;
        for (int i = lane; i < 256; i += 64) { L.huf[0][i] = 0; L.huf[1][i] = 0; }
    }
    uint32_t block = 1u << P.wlog; if (block > n) block = n; if (block > kSub) block = kSub;
    uint64_t t_mf = 0, t_lit = 0, t_seq = 0, t_out = 0, t0 = __builtin_readcyclecounter(), t1;   // phase cycle counters (profiling aid)
#define ZPH(acc) do { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } while (0)
    Entropy pe, ne;
    pe.huf_repeat = kRepNone; pe.rep[0] = 1; pe.rep[1] = 4; pe.rep[2] = 8;
    pe.fse_repeat[0] = pe.fse_repeat[1] = pe.fse_repeat[2] = kRepNone;
    ne = pe;
    int cur = 0;
    bool first = true;
    uint32_t pos = 0;
    while (pos < n) {
        const uint32_t len = min(n - pos, block);
        const uint32_t last = (len >= n - pos) ? 1u : 0u;
        int c = 0;
        if (cap - o < 3 + 2 + 1) return kErrTooSmall;
        const uint32_t bcap = cap - o - 3;
        uint8_t* const out = dst + o + 3;
        {   // ZSTD_window_enforceMaxDist(window, block START) and the nextToUpdate floor (zstd_compress.c:4017-4021)
            const uint32_t start_idx = pos + 2, md = 1u << P.wlog;
            if (start_idx > md) { Z.low_limit = max(Z.low_limit, start_idx - md); Z.dict_limit = max(Z.dict_limit, Z.low_limit); }
            Z.ntu = max(Z.ntu, Z.low_limit);
        }
        bool tables_built = false;
        if (len >= 7) {
            S.nseq = 0; S.nlit = 0;
            ne.rep[0] = pe.rep[0]; ne.rep[1] = pe.rep[1]; ne.rep[2] = pe.rep[2];
            ZPH(t_out);
            if (pos + 2 > Z.ntu + 384) { const uint32_t gap = pos + 2 - Z.ntu - 384; Z.ntu = pos + 2 - min(gap, 192u); }   // ZSTD_buildSeqStore :2890-2896
            uint32_t tail;
            if constexpr (kFast == 1) tail = fast_block(L, S, tab, P, ne.rep, src, pos, pos + len, n, serial, lane);
            else if constexpr (kFast == 2) tail = dfast_block(L, S, tab, tab_s, P, ne.rep, src, pos, pos + len, serial, lane);
            else if constexpr (kTree) tail = P.strat == 7 ? opt_block(L, S, Z, P, ne.rep, Z.chain + (size_t(1) << P.clog), src, pos, pos + len, lane)
                                                     : lazy_block<true>(L, S, Z, P, ne.rep, src, pos, pos + len, n, lane);
            else if (P.strat >= 3) tail = lazy_block<false>(L, S, Z, P, ne.rep, src, pos, pos + len, n, lane);
            else return kErrGeneric;                               // fast and dfast are the other kernels' (zstd_encode_fast_kernel, zstd_encode_dfast_kernel)
            gather_literals(S, src, pos, lane);
            copy_bytes(S.lit + S.nlit, src + pos + len - tail, tail, lane);
            S.nlit += tail;
            ZPH(t_mf);
            const bool suspect = S.nseq == 0 || S.nlit / S.nseq >= 20;
            const int lsz = compress_literals(L, cur, pe, ne, out, bcap, S.lit, S.nlit, suspect, P.strat, lane);
            c = lsz;
            ZPH(t_lit);
            if (lsz >= 0) {
                const int ssz = encode_sequences(L, out + lsz, bcap - uint32_t(lsz), S, P.strat, pe, ne, prevfse, tmp, tables_built, lane);
                c = ssz <= 0 ? ssz : lsz + ssz;
            }
            ZPH(t_seq);
            if (c == kErrTooSmall && len <= bcap) c = 0;
            if (c < 0) return c;
            if (c > 0 && uint32_t(c) >= len - ((len >> 6) + 2)) c = 0;
            if (!first && c < 25 && is_rle(src + pos, len, lane)) { c = 1; if (lane == 0) out[0] = src[pos]; }
            if (c > 1) {                                           // ZSTD_blockState_confirmRepcodesAndEntropyTables
                cur ^= 1; pe = ne;
                if (tables_built && P.strat >= 4) {                // keep this block's FSE tables for the next block's repeat mode
                    const uint32_t* from = reinterpret_cast<const uint32_t*>(&L.ct[0]);
                    uint32_t* to = reinterpret_cast<uint32_t*>(prevfse);
                    for (uint32_t i = lane; i < 3 * sizeof(FseCt) / 4; i += 64) to[i] = from[i];
                }
            }
        }
        if (c == 0) {
            const uint32_t h = last + (len << 3);
            if (len + 3 > cap - o) return kErrTooSmall;
            if (lane < 3) dst[o + lane] = uint8_t(h >> (8 * lane));
            copy_bytes(dst + o + 3, src + pos, len, lane);
            o += 3 + len;
        } else {
            const uint32_t h = c == 1 ? last + (1u << 1) + (len << 3) : last + (2u << 1) + (uint32_t(c) << 3);
            if (lane < 3) dst[o + lane] = uint8_t(h >> (8 * lane));
            o += 3 + uint32_t(c);
        }
        pos += len; first = false;
    }
    ZPH(t_out);
#ifdef ZL_PROF
    if (lane == 0) printf("ZLPROF block %u n %u out %u: mf %llu lit %llu seq %llu out %llu Mclk | searches pf %llu nopf %llu general %llu | Mclk pf %llu nopf %llu general %llu | catch-up ins %llu seqs %llu rep-after %llu | batch Mclk: rows-issued %llu has+cand-issue %llu eq-wait %llu ext %llu (n %llu) keys %llu\n",
                          blockIdx.x, n, o, t_mf >> 20, t_lit >> 20, t_seq >> 20, t_out >> 20, Z.pc[0], Z.pc[1], Z.pc[2], Z.pc[3] >> 20, Z.pc[4] >> 20, Z.pc[5] >> 20, Z.pc[7], Z.pc[8], Z.pc[9], Z.pc[10] >> 20, Z.pc[11] >> 20, Z.pc[12] >> 20, Z.pc[13] >> 20, Z.pc[14], Z.pc[15] >> 20);
#endif
    if (lane == 0) { uint64_t* c = reinterpret_cast<uint64_t*>(S.lit + kSub + 64); c[0] = t_mf; c[1] = t_lit; c[2] = t_seq; c[3] = t_out;
#ifdef Z1_PROF
        for (int i = 0; i < 17; i++) c[4 + i] = S.prof[i];
#endif
    }
    return int(o);
}

// container_mode 0: ZSTD_compress(dst + dst_off, dst_cap, src + src_off, src_len, 1) -> size or -(error number)
// container_mode 1: native/4mc.c:467-489 (capacity n-1; an error stores the block raw)

template <bool kTree, int kFast>
__device__ __forceinline__ void zstd_encode_one(ZLds& L, const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                                                uint8_t* work_base, int container_mode, int level, int serial, uint32_t* lds_tab = nullptr)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const int lane = threadIdx.x;
    const fourmc_block blk = blocks[b];
    // (the descriptor arrives through a vector load: pin what is derived from it to scalar registers, or every length, limit and
    // loop counter of the block is computed on the vector unit)
    const uint32_t n = U(blk.src_len);
    {   // which kernel takes the block: its strategy decides (fast / dfast have kernels of their own, the binary-tree strategies too)
        const uint32_t st = level_params(n, level).strat;
        const int family = st == 1 ? 1 : st == 2 ? 2 : st >= 6 ? 3 : 0, mine = kFast ? kFast : kTree ? 3 : 0;
        if (family != mine) return;                             // another kernel's block
    }
    const uint8_t* src = src_base + U64(blk.src_off);
    uint8_t* dst = dst_base + U64(blk.dst_off);
    const uint32_t cap = container_mode ? (n ? n - 1 : 0) : U(blk.dst_cap);
    int r = zstd_encode_frame<kTree, kFast>(L, src, n, dst, cap, work_base + size_t(b) * (kStoreBytes + table_bytes(level)), level, serial != 0, lane, lds_tab);
    if (container_mode && r <= 0) { copy_bytes(dst, src, n, lane); r = int(n); }
    if (lane == 0) blocks[b].result = r;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void zstd_encode_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                        uint8_t* work_base, int container_mode, int level, int serial)
{
    __shared__ ZLds L;
    zstd_encode_one<false, 0>(L, src_base, dst_base, blocks, nblocks, work_base, container_mode, level, serial);
}

// level 1 (4mz "fast", the configuration the bench quotes)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void zstd_encode_fast_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                             uint8_t* work_base, int container_mode, int level, int serial)
{
    __shared__ ZLds L;
#ifdef Z1_LDS_TABLE
    __shared__ __attribute__((aligned(16))) uint32_t ldstab[1u << 14];
    zstd_encode_one<false, 1>(L, src_base, dst_base, blocks, nblocks, work_base, container_mode, level, serial, ldstab);
#else
    zstd_encode_one<false, 1>(L, src_base, dst_base, blocks, nblocks, work_base, container_mode, level, serial);
#endif
}

// level 3 (4mz "medium")
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void zstd_encode_dfast_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                              uint8_t* work_base, int container_mode, int level, int serial)
{
    __shared__ ZLds L;
    zstd_encode_one<false, 2>(L, src_base, dst_base, blocks, nblocks, work_base, container_mode, level, serial);
}

// level 12, blocks of 256 KiB and less (a file's short last block): binary-tree finder, optimal parser below 16 KiB
__global__ __launch_bounds__(64)
void zstd_encode_tree_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                            uint8_t* work_base, int container_mode, int level, int serial)
{
    __shared__ ZLds L;
    zstd_encode_one<true, 0>(L, src_base, dst_base, blocks, nblocks, work_base, container_mode, level, serial);
}

} // namespace

extern "C" int fourmc_zstd_enc_level_ok(int level) { return level >= 1 && level <= kMaxLevel; }
extern "C" size_t fourmc_zstd_enc_work_bytes(uint32_t n, int level) { return size_t(n) * (kStoreBytes + table_bytes(level)); }

extern "C" hipError_t fourmc_launch_zstd_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                                void* d_work, int container_mode, int level, int serial, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (level < 1 || level > kMaxLevel) return hipErrorInvalidValue;
    // every block is taken by the kernel of its strategy (zstd_encode_one); a level's four rows name at most these kernels
    bool fam[4] = {false, false, false, false};
    for (int c = 0; c < 4; c++) { const uint32_t st = kLevelRows[level - 1][c].strat; fam[st == 1 ? 1 : st == 2 ? 2 : st >= 6 ? 3 : 0] = true; }
    const uint8_t* s8 = static_cast<const uint8_t*>(d_src); uint8_t* d8 = static_cast<uint8_t*>(d_dst); uint8_t* w8 = static_cast<uint8_t*>(d_work);
    if (fam[1]) hipLaunchKernelGGL(zstd_encode_fast_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, w8, container_mode, level, serial);
    if (fam[2]) hipLaunchKernelGGL(zstd_encode_dfast_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, w8, container_mode, level, serial);
    if (fam[0]) hipLaunchKernelGGL(zstd_encode_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, w8, container_mode, level, serial);
    if (fam[3]) hipLaunchKernelGGL(zstd_encode_tree_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, w8, container_mode, level, serial);
    return hipGetLastError();
}
