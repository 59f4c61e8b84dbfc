// 4mc_amd/csrc/lz4_decode.hip — K1: batched LZ4 block decode on gfx950 (wave64).
//
// Replaces the per-block call LZ4_decompress_safe(in, out, csize, usize) of the reference
// (native/4mc.c:661, native/jniDecompressor.c:88 -> native/lz4/lz4.c:2345-2350 -> :1936-2339).
//
// "exact" kernel: one wavefront walks one block's sequences in order and reproduces the
// reference's accept/reject set and negative return codes (including the wider acceptance of its
// x86-64 fast loop, lz4.c:1995-2110), so results are bit-identical on valid AND corrupt input.
//
// Data movement per block (HBM-bound byte work, no MFMA):
//   * compressed stream: 16 B/lane coalesced loads, one 1 KiB granule ahead of use, staged in a
//     4 KiB LDS ring; the parser sees it through a 64-byte per-lane lookahead register (`la`),
//     so token / length / offset bytes are wave-uniform v_readlane reads, not memory round trips;
//   * literals: stored straight from the lookahead register (lane j owns stream byte la_pos+j);
//   * matches: 64 bytes per step, read back from the block's own output (L2-resident, <=64 KiB
//     behind the write cursor); overlapping matches (offset < length) are expanded from the
//     period so that every step is a full-width copy instead of a byte-serial chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"
#include "lz4seg.h"
#include "lz4tile.h"
#include <string.h>
#include <atomic>

namespace {

constexpr int kRing  = 4096;   // LDS bytes of compressed-stream ring per wave
constexpr int kChunk = 1024;   // refill granule: 64 lanes x 16 B
constexpr int kAhead = 2048;   // keep this much of the stream staged beyond the read cursor

struct Stream {
    const uint8_t* abase;   // 16-byte aligned address at or below the block's first byte
    int      delta;         // first byte - abase            (0..15)
    int      qend;          // delta + csize  (end of the stream in aligned coordinates)
    int      fill_hi;       // ring holds aligned positions [.., fill_hi); multiple of kChunk
    uint8_t* ring;          // LDS
    uint4    pend;          // granule [fill_hi, fill_hi + kChunk) already requested from HBM
    uint32_t la;            // lookahead: lane j holds stream byte la_pos + j
    int      la_pos;
    int      lane;

    __device__ __forceinline__ uint4 fetch(int q) const {
        // aligned 16 B granules are safe to read whenever they contain at least one stream byte
        const int g = q + 16 * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g < qend) v = *reinterpret_cast<const uint4*>(abase + g);
        return v;
    }
    __device__ __forceinline__ void advance() {
        *reinterpret_cast<uint4*>(ring + ((fill_hi + 16 * lane) & (kRing - 1))) = pend;
        fill_hi += kChunk;
        pend = fetch(fill_hi);
    }
    __device__ __forceinline__ void init(const uint8_t* src, int csize, uint8_t* lds, int ln) {
        // pointer arithmetic (not an integer round trip) keeps the address space known: global_load,
        // not flat_load - a pending FLAT access would force every later s_waitcnt to vmcnt(0)
        delta = int(reinterpret_cast<uintptr_t>(src) & 15);
        abase = src - delta;
        qend = delta + csize;
        ring = lds; lane = ln;
        fill_hi = 0;
        pend = fetch(0);
        la_pos = -(1 << 30);
        la = 0;
    }
    // load the lookahead register so that lane 0 sits on stream position p
    __device__ __forceinline__ void reload(int p) {
        const int q = p + delta;
        if (q >= fill_hi + kChunk) {               // jumped over staged data (long literal run)
            fill_hi = q & ~(kChunk - 1);
            pend = fetch(fill_hi);
        }
        while (fill_hi < q + kAhead && fill_hi < qend) advance();
        la = ring[(q + lane) & (kRing - 1)];
        la_pos = p;
    }
    __device__ __forceinline__ uint32_t get(int p) {      // p is wave-uniform
        if (p < la_pos || p >= la_pos + 64) reload(p);
        return (uint32_t)__builtin_amdgcn_readlane((int)la, p - la_pos);
    }
};

// Length continuation bytes (lz4.c:1903-1928).  On failure returns false with ip = position the
// reference reports.  Consumes whole runs of 255 per step via a ballot over the lookahead.
__device__ __forceinline__ bool more_len(Stream& s, int& ip, int lim, bool check_first, int& len)
{
    if (check_first && ip >= lim) return false;
    for (;;) {
        if (ip < s.la_pos || ip >= s.la_pos + 64) s.reload(ip);
        const int l0 = ip - s.la_pos;
        const unsigned long long not255 = __ballot(s.la != 255u) >> l0;
        const int avail = 64 - l0;
        int n, add; bool done;
        if (not255 == 0) { n = avail; add = 255 * avail; done = false; }
        else {
            const int t = __builtin_ctzll(not255);
            n = t + 1;
            add = 255 * t + __builtin_amdgcn_readlane((int)s.la, l0 + t);
            done = true;
        }
        if (ip + n > lim) { ip = (ip + 1 > lim + 1) ? ip + 1 : lim + 1; return false; }
        ip += n;
        len = len > 0x40000000 - add ? 0x40000000 : len + add;     // saturates far above any buffer: every later bound check fails as the reference's pointer-wrap tests do (lz4.c:1903-1928)
        if (done) return true;
    }
}

__device__ __forceinline__ void copy_literals(Stream& s, const uint8_t* src, uint8_t* dst,
                                              int ip, int op, int n)
{
    int done = 0;
    if (ip >= s.la_pos && ip < s.la_pos + 64) {
        const int n1 = min(n, s.la_pos + 64 - ip);
        const int p = s.la_pos + s.lane;
        if (p >= ip && p < ip + n1) dst[op + (p - ip)] = (uint8_t)s.la;
        done = n1;
    }
    // long runs: straight HBM -> HBM
    if (done < n) wave_copy(dst + op + done, src + ip + done, n - done, s.lane);
}

// One wave decodes one block.  Mirrors the control flow restated in oracle/lz4_port.c.
// (ip0, op0) != (0, 0): resume behind sequences another kernel has executed (lz4_seg.hip) - only ever at a token the reference's
// fast loop would still be in (at least kMargin stream bytes and kOMargin output bytes from the ends, lz4seg.h)
__device__ __forceinline__ int lz4_decode_block(const uint8_t* src, int csize, uint8_t* dst, int cap,
                                uint8_t* lds, int lane, int ip0 = 0, int op0 = 0)
{
    if (cap < 0) return -1;
    if (cap == 0) return (csize == 1 && src[0] == 0) ? 0 : -1;          // lz4.c:1977-1981
    if (csize == 0) return -1;

    Stream s; s.init(src, csize, lds, lane);
    const int iend = csize, oend = cap;
    int ip = ip0, op = op0;
    bool fast = (oend - op) >= 64;                                      // lz4.c:1990

    for (;;) {
        if (ip < s.la_pos || ip + 24 > s.la_pos + 64) s.reload(ip);
        const uint32_t token = s.get(ip); ip++;
        int lit = int(token >> 4);
        int mlen = int(token & 15);
        bool shortcut = false;

        if (lit == 15) { if (!more_len(s, ip, iend - 15, true, lit)) return -ip - 1; }

        bool check_end;   // does the literal run have to pass the end-of-block test?
        if (fast) {
            check_end = (token >> 4) == 15 ? (op + lit > oend - 32 || ip + lit > iend - 32)
                                           : (ip > iend - 17);
            if (check_end) fast = false;
        } else {
            shortcut = (token >> 4) != 15 && ip < iend - 16 && op <= oend - 32;   // :2128
            check_end = !shortcut;
        }
        if (check_end && (op + lit > oend - 12 || ip + lit > iend - 8)) {
            // terminating literal run (lz4.c:2175-2225)
            if (ip + lit != iend || op + lit > oend) return -ip - 1;
            copy_literals(s, src, dst, ip, op, lit);
            return op + lit;
        }
        copy_literals(s, src, dst, ip, op, lit);
        ip += lit; op += lit;

        const int off = int(s.get(ip)) | (int(s.get(ip + 1)) << 8);
        ip += 2;
        const int from = op - off;

        if (fast) {
            if (mlen == 15) {
                if (!more_len(s, ip, iend - 4, false, mlen)) return -ip - 1;
                mlen += 4;
                if (from < 0) return -ip - 1;
                if (op + mlen >= oend - 64) fast = false;
            } else {
                mlen += 4;
                if (op + mlen >= oend - 64) fast = false;
                else if (from < 0) return -ip - 1;
            }
            if (!fast) {
                if (from < 0) return -ip - 1;
                if (op + mlen > oend - 5) return -ip - 1;
            }
        } else if (shortcut && mlen != 15 && off >= 8 && from >= 0) {
            mlen += 4;                                                  // lz4.c:2143-2152
        } else {
            if (mlen == 15) { if (!more_len(s, ip, iend - 4, false, mlen)) return -ip - 1; }
            mlen += 4;
            if (from < 0) return -ip - 1;                               // lz4.c:2250
            if (op + mlen > oend - 5) return -ip - 1;                   // lz4.c:2317
        }
        if (off == 0) return INT32_MIN;     // undefined in the reference (reads unwritten output)
        copy_match(dst, op, off, mlen, lane);
        op += mlen;
    }
}

// container_mode = 0: raw codec call, result = LZ4_decompress_safe's return value.
// container_mode = 1: one iteration of decodeFourMC's loop (native/4mc.c:603-668) after the
//   checksum pass: blocks flagged FOURMC_BLK_BADSUM are skipped, src_len == dst_cap means a
//   stored block (plain copy, :635-642), a negative codec result becomes FOURMC_BLK_CORRUPT (:662).
__global__ __launch_bounds__(64)
void lz4_decode_exact_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                             fourmc_block* blocks, uint32_t nblocks, int container_mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    int r;
    if (container_mode) {
        if (blk.result == FOURMC_BLK_BADSUM) return;
        if (blk.src_len == blk.dst_cap) {
            wave_copy(dst, src, int(blk.src_len), threadIdx.x);
            r = int(blk.src_len);
        } else {
            r = lz4_decode_block(src, int(blk.src_len), dst, int(blk.dst_cap), ring, threadIdx.x);
            if (r < 0) r = FOURMC_BLK_CORRUPT;
        }
    } else {
        r = lz4_decode_block(src, int(blk.src_len), dst, int(blk.dst_cap), ring, threadIdx.x);
    }
    if (threadIdx.x == 0) blocks[b].result = r;
}


// ================================================================================================
// Fast path ("batch" decoder).  Same block format, same results on every stream it accepts; any
// irregularity (rule violation, offset beyond the produced output, odd end-of-block shape) makes it
// return kRetry and the exact kernel above redoes the block, so error codes stay the reference's.
//
// Instead of one sequence per step it takes a 64-byte window of the compressed stream and
//   1. treats EVERY byte as a candidate token in parallel (lane j: literal count, next-token slot),
//   2. walks the chain of real tokens with v_readlane only (no memory in the serial part),
//   3. prefix-sums the output sizes of all sequences found (typically 6-12),
//   4. produces the batch's output 64 bytes per step in OUTPUT order: each lane finds the sequence
//      that owns its byte (LDS owner map + max-scan), literals are pulled from the window register
//      with a lane permute, match bytes whose source precedes the step are loaded from the
//      block's own output, sources inside the step are resolved by pointer jumping over lanes.
// Long literal runs / long matches (length nibble 15) and the block tail take the one-sequence
// path, which is already 64 bytes wide per step.
constexpr int kRetry = -1000000003;      // internal: "let the exact kernel decide"

// Wave64 inclusive scans on the VALU cross-lane network (DPP row shifts + row broadcasts, no LDS):
// 4 row_shr steps scan each row of 16 lanes, row_bcast15 / row_bcast31 carry the row totals.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }

__device__ __forceinline__ uint32_t scan_add(uint32_t v, int)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v);            // row_bcast15 -> rows 1,3
    v += dpp0<0x143, 0xc>(v);            // row_bcast31 -> rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t scan_max(uint32_t v, int)      // values are >= 0, identity 0
{
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v));
    v = max(v, dpp0<0x143, 0xc>(v));
    return v;
}

#ifdef FOURMC_RESEARCH
#include "../../tools/research/lz4_trio.inc"
#endif

// second pass: blocks the fast path handed back (result == kRetry) are decoded by the exact walker
__global__ __launch_bounds__(64)
void lz4_decode_retry_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                             fourmc_block* blocks, uint32_t nblocks, int container_mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (blk.result != kRetry) return;
    int r = lz4_decode_block(src_base + blk.src_off, int(blk.src_len), dst_base + blk.dst_off, int(blk.dst_cap),
                             ring, threadIdx.x);
    if (container_mode && r < 0) r = FOURMC_BLK_CORRUPT;
    if (threadIdx.x == 0) blocks[b].result = r;
}

// the same for the segment-parallel path (lz4_seg.hip): blocks it handed back entirely (kRetry), and blocks it executed up to the
// token where the reference's end-of-block rules begin (kResume: token and output position in the block's workspace slot)
__global__ __launch_bounds__(64)
void lz4_decode_resume_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                              fourmc_block* blocks, uint32_t nblocks, int container_mode, const uint32_t* ws, int redo,
                              uint32_t ws_stride, uint32_t res_at)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    int ip0 = 0, op0 = 0;
    if (blk.result == lz4seg::kResumeCode) {
        const uint32_t* meta = ws + size_t(b) * ws_stride;            // the block's slot: {token position, output position} at res_at
        ip0 = __builtin_amdgcn_readfirstlane(int(meta[res_at]));
        op0 = __builtin_amdgcn_readfirstlane(int(meta[res_at + 1]));
    } else if (blk.result != kRetry || !redo) return;
    int r = lz4_decode_block(src_base + blk.src_off, int(blk.src_len), dst_base + blk.dst_off, int(blk.dst_cap),
                             ring, threadIdx.x, ip0, op0);
    if (container_mode && r < 0) r = FOURMC_BLK_CORRUPT;
    if (threadIdx.x == 0) blocks[b].result = r;
}

} // namespace

// Block-parallel path (lz4_parse.hip + lz4_exec.hip) for every block, then the exact walker for whatever they handed back
// (rule violations, blocks beyond the parallel path's size limits), so results and error codes stay the reference's.
extern "C" size_t fourmc_lz4_decode_tok_offset(void) { return lz4par::kTokOff; }
// Device workspace of the block-parallel pair (parse records): 10.8 MB per block, 21.7 GiB for a full batch.  Only launches
// that run that pair need it; every other decode path leases nothing (tens of GiB of dead HBM per stream).
extern "C" size_t fourmc_lz4_parse_work_bytes(uint32_t n)
{
    const uint32_t m = n < lz4par::kMaxBatch ? n : lz4par::kMaxBatch;
    return size_t(m ? m : 1) * lz4par::kSlotBytes;
}
extern "C" int fourmc_gpu_get_lz4_decode_path(void);
// Which kernels a launch of n blocks runs, in pieces of how many blocks, with how much workspace: resolved ONCE per call (the
// lease and the launch see the same answer even if another thread changes the selection in between).  shrink: how often the
// workspace could not be had - the pieces halve; below 64 blocks an automatic choice falls back to the exact walker, which needs
// no workspace (until round 5 the fallback was the walk + window copier, which left the product in round 6).
// "auto" by launch size (measured on the container corpus, tools/decode_sizes.sh: 128 .. 512 blocks 13.5 - 15 ms through the tile
// path against 30 ms through either other path, 1024: 24 / 31 ms, 1536: 34 / 34 ms, 2048: 43 / 35 ms, 16 384: 299 / 172 ms): the tile
// path - one workgroup per block, two per CU, the shortest chain per block - up to kAutoTileMax blocks, the segment-parallel path
// - one wave per block, 24 per CU - for launches that fill the chip several times over.  FOURMC_TILE_MAX overrides.
static uint32_t auto_tile_max()
{
    static const uint32_t v = [] { const char* e = getenv("FOURMC_TILE_MAX"); const long x = e ? atol(e) : -1; return x >= 0 ? uint32_t(x) : 1536u; }();
    return v;
}
extern "C" fourmc_lz4_plan fourmc_lz4_decode_plan(uint32_t n, uint32_t shrink)
{
    fourmc_lz4_plan pl; pl.path = fourmc_gpu_get_lz4_decode_path(); pl.batch = n ? n : 1; pl.work_bytes = 0; pl.ok = 1;
    const bool automatic = pl.path == 6;
    if (pl.path == 6) pl.path = n <= auto_tile_max() ? 13 : 11;
    if (pl.path >= 11 && pl.path <= 16) {
        const bool tile = pl.path == 13 || pl.path == 14;
        uint32_t b = tile ? fourmc_lz4_tile_batch() : fourmc_lz4_seg_batch();
        if (b < 64) b = 64;                                       // (FOURMC_TILE_BATCH / FOURMC_SEG_BATCH below 64: an explicit choice would fail before any allocation)
        for (uint32_t k = 0; k < shrink && b >= 64; k++) b /= 2;
        if (b < 64) {
            if (automatic) { pl.path = 2; return pl; }
            pl.ok = 0; return pl;
        }
        pl.batch = n < b ? (n ? n : 1) : b;
        pl.work_bytes = tile ? fourmc_lz4_tile_work_bytes(pl.batch) : fourmc_lz4_seg_work_bytes(pl.batch);
        return pl;
    }
#ifdef FOURMC_RESEARCH
    if (pl.path == 1 || pl.path == 3) pl.work_bytes = fourmc_lz4_parse_work_bytes(n);
#endif
    return pl;
}

// Which path serves LZ4 decode launches.  All produce identical results (anything irregular goes to the exact walker either way):
//   6  "auto"            the tile path up to FOURMC_TILE_MAX blocks per launch, the segment-parallel path above
//   13 "tile"            one workgroup per block, the LZ4 window in LDS (lz4_tile.hip)            14: without the exact walker behind it (test aid)
//   11 "seg"             walk kernel (one lane per stream segment) + batch executor (lz4_seg.hip) 12: test aid
//   2  "exact"           the exact walker alone (one wave per block; also what an automatic choice ends at without any workspace)
// The research side build (make research: libhadoop-4mc-research.so, -DFOURMC_RESEARCH; sources under tools/research/) adds the
// measured alternatives that no launch of the product selects: 0 wave trio, 1 / 3 block-parallel parse + executor, 4 / 5 row
// pipeline, 7 / 8 lane per sequence, 9 / 10 walk + window copier (K1wx: the default of rounds 3 - 4), 15 / 16 group executor (round 6).
// FOURMC_DECODE (auto | tile | seg | exact, and the research names) or fourmc_gpu_set_lz4_decode_path() select.
static int g_decode_path = -1;
static bool path_known(int path)
{
#ifdef FOURMC_RESEARCH
    return path >= 0 && path <= 16;
#else
    return path == 2 || path == 6 || (path >= 11 && path <= 14);
#endif
}
extern "C" void fourmc_gpu_set_lz4_decode_path(int path) { g_decode_path = path_known(path) ? path : 6; }
extern "C" int fourmc_gpu_get_lz4_decode_path(void)
{
    if (g_decode_path < 0) {
        const char* mode = getenv("FOURMC_DECODE");
        int p = 6;
        if (mode && !strcmp(mode, "rows")) p = 4;
        if (mode && !strcmp(mode, "trio")) p = 0;
        if (mode && !strcmp(mode, "rowsonly")) p = 5;
        if (mode && !strcmp(mode, "lanes")) p = 7;
        if (mode && !strcmp(mode, "lanesonly")) p = 8;
        if (mode && !strcmp(mode, "wx")) p = 9;
        if (mode && !strcmp(mode, "wxonly")) p = 10;
        if (mode && !strcmp(mode, "exact")) p = 2;
        if (mode && !strcmp(mode, "seg")) p = 11;
        if (mode && !strcmp(mode, "segonly")) p = 12;
        if (mode && !strcmp(mode, "tile")) p = 13;
        if (mode && !strcmp(mode, "tileonly")) p = 14;
        if (mode && !strcmp(mode, "ring")) p = 15;
        if (mode && !strcmp(mode, "ringonly")) p = 16;
        if (mode && !strcmp(mode, "par")) p = 1;
        if (mode && !strcmp(mode, "paronly")) p = 3;
        g_decode_path = path_known(p) ? p : 6;
    }
    return g_decode_path;
}

extern "C" hipError_t fourmc_launch_lz4_decode(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                               uint32_t n, int container_mode, const fourmc_lz4_plan* plan, void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint8_t* s8 = static_cast<const uint8_t*>(d_src);
    uint8_t* d8 = static_cast<uint8_t*>(d_dst);
    const int path = plan->path;
    // (6 "auto" was resolved by fourmc_lz4_decode_plan: the tile path - lz4_tile.hip, one workgroup per block with the LZ4 window in
    // LDS - up to 1536 blocks, the segment-parallel path above; the exact walker when neither can have its workspace)
#ifdef FOURMC_RESEARCH
    if (path >= 11 && path <= 16) {
#else
    if (path >= 11 && path <= 14) {
#endif
        // walk + executor (tile: lz4_tile.hip, segment-parallel: lz4_seg.hip), then the exact walker for the last bytes of every
        // block and for whatever was handed back; 12 / 14: test aid, blocks handed back stay kRetry
        const bool tile = path == 13 || path == 14, ring = path >= 15;
        const uint32_t step = plan->batch ? plan->batch : n;
        if (plan->work_bytes == 0 || d_work == nullptr) return hipErrorInvalidValue;
        for (uint32_t b0 = 0; b0 < n; b0 += step) {
            const uint32_t m = n - b0 < step ? n - b0 : step;
#ifdef FOURMC_RESEARCH
            hipError_t e = tile ? fourmc_launch_lz4_tile(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream)
                         : ring ? fourmc_launch_lz4_ring(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream)
                                : fourmc_launch_lz4_seg(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream);
#else
            (void)ring;
            hipError_t e = tile ? fourmc_launch_lz4_tile(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream)
                                : fourmc_launch_lz4_seg(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream);
#endif
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(lz4_decode_resume_kernel, dim3(m), dim3(64), 0, stream, s8, d8, d_blocks + b0, m, container_mode,
                               static_cast<const uint32_t*>(d_work), (path == 11 || path == 13 || path == 15) ? 1 : 0,
                               tile ? uint32_t(lz4tile::kWsWords) : uint32_t(lz4seg::kWsWords), tile ? lz4tile::kMetaResIp : lz4seg::kMetaResIp);
        }
        return hipGetLastError();
    }
    if (path == 2) {
        hipLaunchKernelGGL(lz4_decode_exact_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
#ifdef FOURMC_RESEARCH
    if (path == 9 || path == 10) {
        hipError_t e = fourmc_launch_lz4_wx(d_src, d_dst, d_blocks, n, container_mode, stream, nullptr, 0);
        if (e != hipSuccess || path == 10) return e;      // 10: test aid, shows what the path alone did
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    if (path == 4 || path == 5) {
        hipError_t e = fourmc_launch_lz4_rows(d_src, d_dst, d_blocks, n, container_mode, stream);
        if (e != hipSuccess || path == 5) return e;       // 5: test aid, shows what the row pipeline alone did
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    if (path == 7 || path == 8) {
        hipError_t e = fourmc_launch_lz4_lanes(d_src, d_dst, d_blocks, n, container_mode, stream);
        if (e != hipSuccess || path == 8) return e;       // 8: test aid, shows what the lane-per-sequence path alone did
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    if (path == 0) {
        hipLaunchKernelGGL(lz4_decode_fast_kernel, dim3(n), dim3(64 * (kCopiers + 1)), 0, stream, s8, d8, d_blocks, n, container_mode, (const uint32_t*)nullptr, 0u);
        hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
        return hipGetLastError();
    }
    for (uint32_t b0 = 0; b0 < n; b0 += lz4par::kMaxBatch) {
        const uint32_t m = n - b0 < lz4par::kMaxBatch ? n - b0 : lz4par::kMaxBatch;
        hipError_t e = fourmc_launch_lz4_parse(d_src, d_dst, d_blocks + b0, m, container_mode, d_work, stream);
        if (e != hipSuccess) return e;
        e = fourmc_launch_lz4_exec(d_src, d_dst, d_blocks + b0, m, d_work, stream);
        if (e != hipSuccess) return e;
    }
    if (path == 3) return hipGetLastError();              // test aid: show what the parallel path alone did
    hipLaunchKernelGGL(lz4_decode_retry_kernel, dim3(n), dim3(64), 0, stream, s8, d8, d_blocks, n, container_mode);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;                          // (path_known() keeps every other value out)
#endif
}
