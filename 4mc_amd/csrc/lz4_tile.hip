// 4mc_amd/csrc/lz4_tile.hip - K1t: tile LZ4 block decode on gfx950 (wave64), the 64 KiB LZ4 window resident in LDS.
//
// Replaces LZ4_decompress_safe(in, out, csize, usize) per block (native/4mc.c:661, native/jniDecompressor.c:88 ->
// native/lz4/lz4.c:2345-2350 -> :1936-2339) for every block the exact walker (lz4_decode.hip) does not have to see.
//
//   WALK kernel, one wave per block, one LANE per stream segment (as lz4_seg.hip: a chain started at an arbitrary byte falls onto
//     the true token chain after a few hops), but what it leaves is ONE BIT PER STREAM BYTE - "a token starts here" - in words the
//     lane owns (segment lengths are multiples of 32).  A chain that enters a segment somewhere else than assumed rewrites the
//     segment's words from its start until it falls onto a bit of the chain that is already there.  No record lists: 1/8 byte of
//     workspace per stream byte instead of 8 bytes per sequence written and read back.
//   EXEC kernel, one workgroup of 512 threads per block, two per CU.  The last 64 KiB of output - everything an LZ4 offset can
//     reach - live in an LDS ring, so a match never goes to memory: HBM sees the stream once (coalesced 16-byte loads), the bitmap
//     once, and the output once (aligned 16-byte stores).  Per CHUNK of 2 KiB of stream: the tokens are compacted out of the
//     bitmap, decoded one per thread from the staged stream and placed by a prefix sum.  The chunk's output is produced in TILES of
//     <= 4096 bytes, ONE THREAD PER OUTPUT BYTE: every sequence marks where its literal part and its match part begin, a max-scan
//     gives every byte its (sequence, part); literal bytes come from the staged stream, match bytes whose source lies in front of
//     the tile from the ring (every read of old ring contents happens before the tile's first write: the ring is exactly 64 KiB),
//     match bytes whose source lies INSIDE the tile keep a 16-bit pointer to it and are resolved by chasing pointers to a byte that
//     is final - the threads need no order among themselves for that, and a resolved byte is final for everyone behind it (what a
//     serial decoder does as a chain of dependent copies is here a few dependent LDS reads per byte, all bytes at once).
//     A sequence of any length is simply clipped to the tile: no escape paths.
//   The last 64 stream bytes / 128 output bytes of a block - where the reference's end-of-block rules apply (lz4.c:2120-2330) - and
//     anything irregular go to the exact walker: it RESUMES at the token the fast path stopped at (kResume) or redoes the block
//     (kRetry), so accept / reject set and return codes stay the reference's.
//
// All byte work; no MFMA.  tools/model/tile_decode_model.c is the executable model these kernels were written from.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"
#include "lz4tile.h"
#ifndef FOURMC_TILE_WIN
#define FOURMC_TILE_WIN 64
#endif

namespace {

using namespace lz4tile;

typedef __attribute__((address_space(1))) uint8_t gbyte;
typedef __attribute__((address_space(1))) const uint8_t cgbyte;
typedef __attribute__((address_space(1))) uint32_t gword;
typedef __attribute__((address_space(1))) const uint32_t cgword;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld4u(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32_u*>(p); }
__device__ __forceinline__ u32x2 ld8u(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32x2_u*>(p); }
__device__ __forceinline__ u32x4 ld16u_g(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32x4_u*>(p); }
__device__ __forceinline__ void st16g(gbyte* p, u32x4 v) { *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(p) = v; }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
// inclusive scans over the 64 lanes (values that are 0 where a lane has nothing: 0 is the identity of both)
__device__ __forceinline__ uint32_t scan_add(uint32_t v)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t scan_max(uint32_t v)
{
    v = umax(v, dpp0<0x111, 0xf>(v)); v = umax(v, dpp0<0x112, 0xf>(v)); v = umax(v, dpp0<0x114, 0xf>(v)); v = umax(v, dpp0<0x118, 0xf>(v));
    v = umax(v, dpp0<0x142, 0xa>(v)); v = umax(v, dpp0<0x143, 0xc>(v));
    return v;
}
// inclusive prefix sum / maximum / minimum over the lanes of a row of 16 (three DPP steps): lane 7 holds the first eight lanes' result
template <int CTRL> __device__ __forceinline__ uint32_t dppk(uint32_t old, uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(v), CTRL, 0xf, 0xf, false)); }
__device__ __forceinline__ uint32_t row_add8(uint32_t v) { v += dppk<0x111>(0u, v); v += dppk<0x112>(0u, v); v += dppk<0x114>(0u, v); return v; }
__device__ __forceinline__ uint32_t row_max8(uint32_t v) { v = umax(v, dppk<0x111>(0u, v)); v = umax(v, dppk<0x112>(0u, v)); v = umax(v, dppk<0x114>(0u, v)); return v; }
__device__ __forceinline__ uint32_t row_min8(uint32_t v) { v = min(v, dppk<0x111>(0xFFFFFFFFu, v)); v = min(v, dppk<0x112>(0xFFFFFFFFu, v)); v = min(v, dppk<0x114>(0xFFFFFFFFu, v)); return v; }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(l))); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }

// profiling build (make tprof, -DK1T_PROF): cycle counters per phase, left in the spare words of the block's meta area
#ifdef K1T_PROF
struct Prof {
    unsigned long long t[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void add(int i, unsigned long long& since) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t[i] += n - since; since = n; }
    __device__ __forceinline__ void count(int i, unsigned long long n = 1) { t[i] += n; }
    __device__ __forceinline__ void dump(gword* meta, uint32_t at, int n, bool who) const { if (who) for (int i = 0; i < n; i++) { meta[at + 2 * i] = uint32_t(t[i]); meta[at + 2 * i + 1] = uint32_t(t[i] >> 32); } }
};
#else
struct Prof {
    __device__ __forceinline__ unsigned long long now() const { return 0; }
    __device__ __forceinline__ void add(int, unsigned long long&) {}
    __device__ __forceinline__ void count(int, unsigned long long = 1) {}
    __device__ __forceinline__ void dump(gword*, uint32_t, int, bool) const {}
};
#endif

__device__ __forceinline__ bool eligible(const fourmc_block& blk)
{ return blk.src_len >= kMinSrc && blk.src_len <= kMaxSrc && blk.dst_cap >= kMinCap && blk.dst_cap <= lz4par::kDstMax; }

// ================================================================================================ WALK kernel
// Length extension bytes from q on - 255, 255, ..., b (b < 255) - eight per load: adds them to `len`, leaves q behind the last.
// false: they reach `limit`, or (as lz4_seg.hip's walk: lengths beyond `cap` are the exact walker's) len passes cap on a 255.
__device__ __forceinline__ bool ext_run(cgbyte* s, uint32_t limit, uint32_t& q, uint32_t& len, uint32_t cap)
{
    for (;;) {
        if (q + 8u <= limit) {
            const u32x2 v = ld8u(s + q);
            const uint32_t nlo = ~v.x, nhi = ~v.y;
            if ((nlo | nhi) == 0u) { len += 8u * 255u; q += 8u; if (len > cap) return false; continue; }
            const uint32_t k = nlo ? uint32_t(__builtin_ctz(nlo)) >> 3 : 4u + (uint32_t(__builtin_ctz(nhi)) >> 3);
            const uint32_t b = k < 4u ? (v.x >> (8u * k)) & 255u : (v.y >> (8u * (k - 4u))) & 255u;
            len += 255u * k;
            if (k && len > cap) return false;
            len += b; q += k + 1u;
            return true;
        }
        if (q >= limit) return false;
        const uint32_t b = s[q++]; len += b;
        if (b != 255u) return true;
        if (len > cap) return false;
    }
}

struct Hop { uint32_t next; bool stop; };
// One token at p (per lane), bytes from memory.  stop: the token or its bytes reach beyond limit = csize - kMargin; the chain halts
// AT p and the exact walker takes over there.  Every byte read lies below csize.
__device__ __forceinline__ Hop decode_tok(cgbyte* s, uint32_t limit, uint32_t p)
{
    Hop h; h.next = p; h.stop = true;
    if (p >= limit) return h;
    const u32x2 L0 = ld8u(s + p);                               // p + 8 <= csize - 56
    const uint32_t tok = L0.x & 255u, mn = tok & 15u;
    uint32_t ll = tok >> 4, q = p + 1;
    if (ll == 15) {
        uint32_t b = (L0.x >> 8) & 255u; ll += b; q++;
        if (b == 255u) {
            if (ll > (1u << 23) || !ext_run(s, limit, q, ll, 1u << 23)) return h;
        }
    }
    const uint32_t mo = q + ll;
    if (mo + 2 > limit) return h;
    uint32_t q2 = mo + 2;
    if (mn == 15) { uint32_t ml = 0; if (!ext_run(s, limit, q2, ml, 0xFFFFFFFFu)) return h; }
    if (q2 > limit) return h;
    h.next = q2; h.stop = false;
    return h;
}

struct LaneSeg {            // one lane's segment: words [sj >> 5, wend) of the bitmap are the lane's
    gword* bm;
    uint32_t sj, seg_end, wend;
    uint32_t exitp, entry; bool tail;
};
// leave word `curw` behind with `acc`, zero the words up to `w`
__device__ __forceinline__ void bm_advance(gword* bm, uint32_t& curw, uint32_t& acc, uint32_t w)
{
    bm[curw] = acc;
    for (uint32_t x = curw + 1; x < w; x++) bm[x] = 0;
    curw = w; acc = 0;
}
// Phase 1: the chain that starts at the segment's first byte, walked to the segment's end, its tokens' bits into the lane's words.
// The lane reads its stretch of the stream through a window of its own in LDS - a ring of kWinDw dwords, dword k of lane l at word
// k * 64 + l, so that 64 lanes reading "their" dword never meet in a bank.  The windows of ALL lanes still walking are topped up
// together whenever one of them has less than a quarter of the window ahead.
constexpr uint32_t kWinDw = FOURMC_TILE_WIN;    // dwords of stream window per lane (LDS: 256 bytes x kWinDw per wave)
template <uint32_t kWinDw>
__device__ __forceinline__ void walk_first(LaneSeg& g, cgbyte* s, uint32_t csize, uint32_t limit, uint32_t* ring)
{
    constexpr uint32_t M = kWinDw - 1u, W = 4u * kWinDw, LOW = W / 4u;
    uint32_t p = g.sj; bool tail;
    uint32_t curw = g.sj >> 5, acc = 0;
    uint32_t wlo = p & ~15u, whi = wlo;                                // the ring holds stream bytes [wlo, whi), both multiples of 16
    const uint32_t fill_end = csize & ~15u;                            // whole 16-byte pieces only
    auto in_win = [&](uint32_t a) -> bool { return a >= wlo && a + 4u <= whi; };
    auto win4 = [&](uint32_t a) -> uint32_t {
        const uint32_t d = a >> 2, lo = ring[(d & M) * 64u], hi = ring[((d + 1u) & M) * 64u];
        return __builtin_amdgcn_alignbyte(hi, lo, a & 3u);
    };
    // 4 stream bytes at a: from the lane's window; from memory for the lanes whose window does not hold them - on a path of its own,
    // taken when ANY lane needs it (a load under a per-lane condition would put a wait for all memory operations behind every token)
    auto get4 = [&](uint32_t a) -> uint32_t {
        if (__builtin_expect(__ballot(!in_win(a)) != 0, 0)) {
            uint32_t v = 0;
            if (!in_win(a)) v = ld4u(s + a);
            asm volatile("" : "+v"(v));
            return in_win(a) ? win4(a) : v;
        }
        return win4(a);
    };
    for (;;) {
        if (p >= g.seg_end) { tail = false; break; }
        if (p >= limit) { tail = true; break; }
        if (__ballot(p + LOW > whi && whi < fill_end)) {
            if (p >= whi || p < wlo) { wlo = p & ~15u; whi = wlo; }
            uint32_t target = (p & ~15u) + (W - 16u); target = target < fill_end ? target : fill_end;
            for (int round = 0; round < (W > 128u ? 2 : 1); round++) {
                if (!__ballot(whi < target)) break;
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) if (whi + 16u * j < target) v[j] = ld16u_g(s + whi + 16u * j);
#pragma unroll
                for (int j = 0; j < 8; j++) if (whi + 16u * j < target) {
                    const uint32_t d = (whi >> 2) + 4u * j;
                    ring[(d & M) * 64u] = v[j].x; ring[((d + 1u) & M) * 64u] = v[j].y; ring[((d + 2u) & M) * 64u] = v[j].z; ring[((d + 3u) & M) * 64u] = v[j].w;
                }
                const uint32_t got = target - whi; whi += got < 128u ? got : 128u;
            }
            if (whi - wlo > W) wlo = whi - W;
        }
        const uint32_t L0 = get4(p);
        const uint32_t tok = L0 & 255u, mn = tok & 15u;
        uint32_t ll = tok >> 4, q = p + 1; bool stop = false;
        if (ll == 15) {
            uint32_t bq = (L0 >> 8) & 255u; ll += bq; q++;
            if (bq == 255u) stop = ll > (1u << 23) || !ext_run(s, limit, q, ll, 1u << 23);
        }
        const uint32_t mo = q + ll;
        if (stop || mo + 2 > limit) { tail = true; break; }
        uint32_t q2 = mo + 2;
        if (mn == 15) {
            const uint32_t L1 = get4(mo);
            const uint32_t e0 = (L1 >> 16) & 255u; q2++;
            if (e0 == 255u) {
                const uint32_t e1 = L1 >> 24; q2++;
                if (e1 == 255u) { uint32_t ml = 0; stop = !ext_run(s, limit, q2, ml, 0xFFFFFFFFu); }
            }
        }
        if (stop || q2 > limit) { tail = true; break; }
        const uint32_t w = p >> 5;
        if (w != curw) bm_advance(g.bm, curw, acc, w);
        acc |= 1u << (p & 31u);
        p = q2;
    }
    bm_advance(g.bm, curw, acc, g.wend);
    g.exitp = p; g.entry = g.sj; g.tail = tail;
}
// The true chain enters the segment at e: rewrite the lane's words from the segment's start for it, until it falls onto a bit of
// the chain that is there already (every bit there is a token of ONE chain: what a walk from the segment's start or an earlier
// entry left) or leaves the segment.
__device__ __forceinline__ void walk_from_entry(LaneSeg& g, cgbyte* s, uint32_t limit, uint32_t e)
{
    uint32_t q = e, curw = g.sj >> 5, acc = 0, old = g.bm[curw];
    g.entry = e;
    for (;;) {
        if (q >= g.seg_end) { g.exitp = q; g.tail = false; break; }
        const uint32_t w = q >> 5;
        if (w != curw) { const uint32_t o2 = g.bm[w]; bm_advance(g.bm, curw, acc, w); old = o2; }
        if ((old >> (q & 31u)) & 1u) { g.bm[curw] = acc | (old & ~((1u << (q & 31u)) - 1u)); return; }      // merged: the bits from q on stay
        const Hop h = decode_tok(s, limit, q);
        if (h.stop) { g.exitp = q; g.tail = true; break; }
        acc |= 1u << (q & 31u);
        q = h.next;
    }
    bm_advance(g.bm, curw, acc, g.wend);
}

__global__ __launch_bounds__(64)
void lz4_tile_walk_kernel(const uint8_t* __restrict__ src_base, const fourmc_block* blocks, uint32_t nblocks,
                          int container_mode, uint32_t* ws)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    gword* meta = (gword*)(ws + size_t(b) * kWsWords);
    const uint32_t lane = threadIdx.x;
    const bool skip = (container_mode && (blk.result == FOURMC_BLK_BADSUM || blk.src_len == blk.dst_cap)) || !eligible(blk);
    if (skip) { if (lane == 0) meta[kMetaStatus] = 0; return; }
    cgbyte* s = (cgbyte*)(src_base + blk.src_off);
    const uint32_t csize = blk.src_len, limit = csize - kMargin;
    uint32_t nseg = limit / kMinSeg; nseg = nseg < 1 ? 1 : (nseg > uint32_t(kSegs) ? uint32_t(kSegs) : nseg);
    const uint32_t seglen = ((limit + nseg - 1) / nseg + 31u) & ~31u;
    const uint32_t nwords = (csize + 31u) >> 5;
    __shared__ uint32_t win[kWinDw * 64];
    uint32_t* ring = win + lane;

    LaneSeg g;
    g.bm = meta + kMetaWords;
    g.sj = lane * seglen;
    g.seg_end = (lane + 1 == nseg) ? 0xFFFFFFFFu : (lane + 1) * seglen;
    g.wend = (lane + 1 == nseg) ? nwords : ((lane + 1) * seglen) >> 5;
    g.exitp = 0; g.entry = 0xFFFFFFFFu; g.tail = true;
    const bool mine = lane < nseg;
    Prof pf; unsigned long long tp = pf.now();
    // phase 1: every lane its own chain
    if (mine) walk_first<kWinDw>(g, s, csize, limit, ring);
    pf.add(0, tp);
    // phase 2: the chain of the segment in front left at `pe`: assume it is the true one, thread it into this segment
    {
        const int from = int(lane ? lane - 1 : 0) * 4;           // lane j reads lane j-1
        const uint32_t pe = uint32_t(__builtin_amdgcn_ds_bpermute(from, int(g.exitp)));
        const uint32_t pt = uint32_t(__builtin_amdgcn_ds_bpermute(from, int(g.tail ? 1u : 0u)));
        uint32_t sj = pe / seglen; sj = sj > nseg - 1 ? nseg - 1 : sj;
        if (mine && lane >= 1 && !pt && sj == lane && pe != g.sj) walk_from_entry(g, s, limit, pe);
    }
    pf.add(1, tp);
    // phase 3: follow the true chain through the segments; redo what was assumed wrong (one lane at a time: rare); segments the
    // chain jumps over have no tokens
    uint32_t cur = 0, tail_ip = 0;
    for (;;) {
        const uint32_t ex = rdl(g.exitp, cur);
        if (rdl(g.tail ? 1u : 0u, cur)) { tail_ip = ex; break; }
        uint32_t j = ex / seglen; j = j > nseg - 1 ? nseg - 1 : j;
        if (lane > cur && lane < j) {
            for (uint32_t x = g.sj >> 5; x < g.wend; x++) g.bm[x] = 0;
        }
        if (rdl(g.entry, j) != ex) {
            if (lane == j) walk_from_entry(g, s, limit, ex);
            pf.count(3);
        }
        cur = j;
    }
    pf.add(2, tp); pf.dump(meta, kMetaProf, 4, lane == 0);
    if (lane == 0) { meta[kMetaTailIp] = tail_ip; meta[kMetaStatus] = 1; }
}

// ================================================================================================ EXEC kernel
// A thread owns the 8 bytes of one 8-byte-aligned group of the ring per tile (tile coordinate u = 8 t + k; the tile's first byte
// sits at u = mis = its ring address & 7): its marks are scanned, its sources fetched, its pointers chased and its bytes written by
// the same thread - the scan's result stays in registers, the group's pointers travel as one 16-byte LDS access and its bytes as
// one 8-byte access.
//   code[u]   first the MARKS (0: none; e: source entry e begins at u), then the POINTERS of the tile's bytes: twice the tile
//             coordinate (= the byte offset in code[]) of the byte it copies - its own for a byte that is final
//   ent[e]    {w0, w1}: the byte at tile coordinate u comes from ring / stage address ((u + w0 + gb) & 0xFFFF) | w1 (w1 = 0x10000: the
//             staged stream, which lies 64 KiB behind the ring's first byte); it is a byte of the tile itself when
//             ulo <= u + w0 < 4096 (w0 = -offset for matches, 0x20000 | ... for everything else)
// What bounds this kernel is the vector instruction stream (wave64 on a SIMD16: one vector instruction per 4 clk and SIMD; 2.5 G
// wave-instructions per ms on the chip), so the per-byte passes are written for few of them: ~7 per byte in the source pass, one
// per byte and round in the chase (the LDS read IS the step: the value read is the next address).
constexpr uint32_t kGroup    = kTile / kThreads;                           // 8
constexpr uint32_t kOffCode  = 0;
constexpr uint32_t kOffEnt   = kOffCode + 2u * kTile;
constexpr uint32_t kScWords  = 44;
constexpr uint32_t kOffSc    = kOffEnt + 8u * kEntries;
constexpr uint32_t kOffRing  = (kOffSc + 4u * kScWords + 15u) & ~15u;
constexpr uint32_t kOffStage = kOffRing + kRing;
constexpr uint32_t kLdsBytes = kOffStage + kStage;
static_assert(kLdsBytes <= 81920, "two workgroups per CU");
constexpr uint32_t S_NTOK = 0, S_NEXT = 1, S_CUTPOS = 2, S_BYDL = 3, S_SUM = 4, S_NF = 12, S_FB = 20, S_TOT = 28, S_BYE = 36, S_SLOW = 37;

// Workgroup barrier for LDS traffic only: every LDS operation of the wave is done, then the barrier.  (__syncthreads() also waits for
// the wave's outstanding GLOBAL loads and stores - the next chunk's bytes, loaded a chunk ahead, and the tile's flush would be waited
// for at the next of the ~8 barriers per tile.)
#define WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// ... and for what the waves of the workgroup hand each other through GLOBAL memory (the fused walk's bitmap words: written by the
// segment's thread, read by its neighbour's and by the one thread of phase 3): the stores have left for the CU's write-through L1 /
// the L2 before the barrier lets anyone read (the plain barrier relied on same-CU ordering it did not enforce)
#define WG_BARRIER_G() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return uint32_t(uintptr_t((const __attribute__((address_space(3))) void*)p)); }
// p[k] = the 16-bit word at LDS address p[k] (eight independent reads, one wait)
__device__ __forceinline__ void hop8(uint32_t (&p)[8])
{
    uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
    asm volatile("s_nop 1\n\tds_read_u16 %0, %8\n\tds_read_u16 %1, %9\n\tds_read_u16 %2, %10\n\tds_read_u16 %3, %11\n\t"
                 "ds_read_u16 %4, %12\n\tds_read_u16 %5, %13\n\tds_read_u16 %6, %14\n\tds_read_u16 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory");
    p[0] = q0; p[1] = q1; p[2] = q2; p[3] = q3; p[4] = q4; p[5] = q5; p[6] = q6; p[7] = q7;
}
// FUSED: the walk runs in front, by the same workgroup - one stream segment per THREAD (512 chains in lockstep instead of 64: a
// block's walk takes a tenth of the one-wave kernel's time, which is what a launch of a few hundred blocks waits for), the per-lane
// stream windows in the LDS that becomes the ring afterwards, the hand-over words in what becomes code[]; the chain through the
// segments is followed by one thread over those words.
constexpr uint32_t kFusedWin = 32;               // dwords of stream window per lane: 8 KiB per wave, 64 KiB for the workgroup = the ring's bytes
constexpr uint32_t S_TAIL = 42, S_ANY = 43;
template <bool FUSED>
__global__ __launch_bounds__(kThreads)
void lz4_tile_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                          int container_mode, uint32_t* ws)
{
#ifdef FOURMC_TILE_PAD      // (occupancy experiment: more LDS than needed = one workgroup per CU)
    __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes + FOURMC_TILE_PAD];
#else
    __shared__ __attribute__((aligned(16))) uint8_t smem[kLdsBytes];
#endif
    uint16_t* const code = reinterpret_cast<uint16_t*>(smem + kOffCode);     // marks, then pointers (while a chunk's tokens are decoded: their positions)
    uint32_t* const ent  = reinterpret_cast<uint32_t*>(smem + kOffEnt);
    uint32_t* const sc   = reinterpret_cast<uint32_t*>(smem + kOffSc);
    uint8_t*  const ring = smem + kOffRing;                                  // [0, 64 KiB): the output window; [64 KiB, + kStage): the chunk's stream bytes
    uint8_t*  const stage = smem + kOffStage;
    uint16_t* const toks = code + 128;                                       // (behind the chunk's bitmap words)
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = rfl(tid >> 6);
    if (container_mode && blk.src_len == blk.dst_cap) {                 // stored block (native/4mc.c:635-642)
        const uint8_t* sp = src_base + blk.src_off; uint8_t* dp = dst_base + blk.dst_off;
        const uint32_t n = blk.src_len;
        const uint32_t head = min(n, uint32_t((16u - uint32_t(uintptr_t(dp) & 15u)) & 15u));
        if (tid < head) dp[tid] = sp[tid];
        uint32_t k = head + 16u * tid;
        for (; k + 16u <= n; k += 16u * kThreads) *reinterpret_cast<uint4*>(dp + k) = ld16u(sp + k);
        const uint32_t body = head + ((n - head) & ~15u);
        if (body + tid < n) dp[body + tid] = sp[body + tid];
        if (tid == 0) blocks[b].result = int(blk.src_len);
        return;
    }
    gword* meta = (gword*)(ws + size_t(b) * kWsWords);
    if (!eligible(blk) || (!FUSED && rfl(meta[kMetaStatus]) != 1u)) { if (tid == 0) blocks[b].result = lz4par::kRetryCode; return; }
    cgbyte* s = (cgbyte*)(src_base + blk.src_off);
    gbyte* dst = (gbyte*)(dst_base + blk.dst_off);
    cgword* bm = (cgword*)(meta + kMetaWords);
    const uint32_t csize = blk.src_len, cap = blk.dst_cap, olimit = cap - kOMargin;
    const uint32_t nwords = (csize + 31u) >> 5;
    Prof pf; unsigned long long tp = pf.now();
    uint32_t tail_ip;
    if (FUSED) {
        const uint32_t limit = csize - kMargin;
        uint32_t nseg = limit / kMinSeg; nseg = nseg < 1 ? 1 : (nseg > uint32_t(kThreads) ? uint32_t(kThreads) : nseg);
        const uint32_t seglen = ((limit + nseg - 1) / nseg + 31u) & ~31u;
        uint32_t* const X = reinterpret_cast<uint32_t*>(code);          // [t] where the chain of segment t left, [512 + t] whether it ended in the tail,
        constexpr uint32_t XT = kThreads, XE = 2 * kThreads, XD = 3 * kThreads;     // [1024 + t] where it was entered, [1536 + t] no token of the true chain in it
        auto seg_of = [&](uint32_t j) -> LaneSeg {
            LaneSeg g; g.bm = meta + kMetaWords; g.sj = j * seglen;
            g.seg_end = (j + 1 == nseg) ? 0xFFFFFFFFu : (j + 1) * seglen;
            g.wend = (j + 1 == nseg) ? nwords : ((j + 1) * seglen) >> 5;
            g.exitp = 0; g.entry = 0xFFFFFFFFu; g.tail = true;
            return g;
        };
        LaneSeg g = seg_of(tid);
        const bool mine = tid < nseg;
        // phase 1: every thread its own chain
        if (mine) walk_first<kFusedWin>(g, s, csize, limit, reinterpret_cast<uint32_t*>(ring) + wv * (kFusedWin * 64u) + lane);
        pf.add(8, tp);
        X[tid] = g.exitp; X[XT + tid] = g.tail ? 1u : 0u; X[XD + tid] = 0u;
        if (tid == 0) sc[S_ANY] = 0u;
        WG_BARRIER_G();
        pf.add(9, tp);
        // phase 2: the chain of the segment in front left at `pe`: assume it is the true one, thread it into this segment
        {
            const uint32_t pj = tid ? tid - 1u : 0u;
            const uint32_t pe = X[pj], pt = X[XT + pj];
            WG_BARRIER_G();                                                // (everyone has read what phase 1 left)
            uint32_t sj = pe / seglen; sj = sj > nseg - 1 ? nseg - 1 : sj;
            if (mine && tid >= 1 && !pt && sj == tid && pe != g.sj) walk_from_entry(g, s, limit, pe);
            X[tid] = g.exitp; X[XT + tid] = g.tail ? 1u : 0u; X[XE + tid] = g.entry;
        }
        WG_BARRIER_G();
        // ... and again where the chain in front did not fall back onto its own before its segment ended (it then leaves somewhere
        // else than assumed): all such segments at once, a few rounds; what is still open after them is the one thread's below
        for (uint32_t round = 1; round <= 6u; round++) {
            const uint32_t pj = tid ? tid - 1u : 0u;
            const uint32_t pe = X[pj], pt = X[XT + pj];
            uint32_t sj = pe / seglen; sj = sj > nseg - 1 ? nseg - 1 : sj;
            const bool need = mine && tid >= 1 && !pt && sj == tid && pe != g.entry;
            if (__ballot(need) && lane == 0) sc[S_ANY] = round;
            WG_BARRIER_G();                                                // (and everyone has read what the round before left)
            if (rfl(sc[S_ANY]) != round) break;
            if (need) { walk_from_entry(g, s, limit, pe); X[tid] = g.exitp; X[XT + tid] = g.tail ? 1u : 0u; X[XE + tid] = g.entry; }
            WG_BARRIER_G();
        }
        pf.add(10, tp);
        // phase 3: one thread follows the true chain through the segments and redoes what was assumed wrong (rare)
        if (tid == 0) {
            uint32_t cur = 0, tip = 0;
            for (;;) {
                const uint32_t ex = X[cur];
                if (X[XT + cur]) { tip = ex; break; }
                uint32_t j = ex / seglen; j = j > nseg - 1 ? nseg - 1 : j;
                for (uint32_t d = cur + 1; d < j; d++) X[XD + d] = 1u;
                if (X[XE + j] != ex) {
                    LaneSeg h = seg_of(j); h.exitp = X[j]; h.tail = X[XT + j] != 0u;
                    walk_from_entry(h, s, limit, ex);
                    X[j] = h.exitp; X[XT + j] = h.tail ? 1u : 0u;
                }
                cur = j;
            }
            sc[S_TAIL] = tip;
        }
        WG_BARRIER_G();
        pf.add(11, tp);
        if (mine && X[XD + tid]) for (uint32_t x = g.sj >> 5; x < g.wend; x++) g.bm[x] = 0;       // segments the chain jumps over have no tokens
        tail_ip = rfl(sc[S_TAIL]);
        __threadfence();                                                 // the bitmap is read back by other threads of the workgroup
        WG_BARRIER();
        pf.add(12, tp);
    } else tail_ip = rfl(meta[kMetaTailIp]);
    const uint32_t A = uint32_t(uintptr_t(dst)) & 0xFFFFu;             // ring index of output position P: (A + P) & 0xFFFF - congruent to P's address mod 16
    uint32_t ip = 0, opos = 0, flushed = 0, res_ip = tail_ip;
    uint32_t nmax = 256;                                                 // sequences the next chunk decodes: about what fills one tile
    bool cut = false, failed = false;
    const uint32_t u0 = kGroup * tid;
    const uint32_t Lcode = lds_addr(code);
    uint32_t self[kGroup];                                               // LDS addresses of the thread's own pointers
#pragma unroll
    for (uint32_t k = 0; k < kGroup; k++) self[k] = Lcode + 2u * (u0 + k);
    *reinterpret_cast<u32x4*>(code + u0) = u32x4{0, 0, 0, 0};
    if (tid == 0) { ent[2] = 0x20000u; ent[3] = 0; ent[2 * (kEntries - 1)] = 0x20000u; ent[2 * (kEntries - 1) + 1] = 0; }     // the bytes around a tile keep what the ring holds

    // the next chunk's stream bytes and bitmap word are loaded a chunk ahead, into registers
    u32x4 nst = u32x4{0, 0, 0, 0}; uint32_t nbw = 0;
    auto prefetch = [&](uint32_t at) {
        // (no "else zero": a value merged with a load's result is waited for where it is merged.  A piece that would reach beyond the
        // stream is read from its last 16 bytes instead: nothing a token of the bitmap needs lies there.)
        const uint32_t sb = at & ~15u, a = sb + 16u * tid;
        if (tid < kStage / 16u) nst = ld16u_g(s + (a + 16u <= csize ? a : csize - 16u));
        if (tid < kChunk / 32u) { const uint32_t wi = (at >> 5) + tid; nbw = bm[wi < nwords ? wi : nwords - 1u]; }      // (words beyond the stream: masked by the chunk's end)
    };
    auto arrived = [&]() { asm volatile("" : "+v"(nst.x), "+v"(nst.y), "+v"(nst.z), "+v"(nst.w), "+v"(nbw)); };
    if (tail_ip > 0) prefetch(0);
    arrived();                                                           // (waited for once, here: inside the loop the wait sits in front of a tile's flush)
    WG_BARRIER();

    while (ip < tail_ip && !cut) {
        // ---- the chunk's stream bytes into the stage; token positions, compacted: thread t looks at the bits of stream bytes
        // [3 t, 3 t + 3) of the chunk - a token takes at least three bytes, so at most one begins there - and its rank among the
        // chunk's tokens is a ballot, a count of the lanes below and the counts of the waves in front
        const uint32_t sbase = ip & ~15u;
        {
            const uint32_t cb = (ip >> 5) << 5;
            uint32_t cend = cb + kChunk; cend = cend < tail_ip ? cend : tail_ip;
            if (tid < kStage / 16u) *reinterpret_cast<u32x4*>(stage + 16u * tid) = nst;
            uint32_t* const bw = reinterpret_cast<uint32_t*>(code);      // (49 words of what is zero between tiles; zeroed again below)
            if (tid <= kChunk / 32u) bw[tid] = tid < kChunk / 32u ? nbw : 0u;
            WG_BARRIER();
            const uint32_t bo = 3u * tid;
            const uint32_t bits3 = __builtin_amdgcn_alignbit(bw[(bo >> 5) + 1u], bw[bo >> 5], bo & 31u) & 7u;
            const uint32_t ppos = cb + bo + uint32_t(__builtin_ctz(bits3 | 8u));
            const bool has = bits3 != 0u && ppos >= ip && ppos < cend;
            const unsigned long long hm = __ballot(has);
            const uint32_t rk_w = __builtin_amdgcn_mbcnt_hi(uint32_t(hm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(hm), 0u));
            if (lane == 0) sc[S_SUM + wv] = uint32_t(__builtin_popcountll(hm));
            WG_BARRIER();
            const uint32_t c8 = row_add8(sc[S_SUM + (lane & 7u)]);
            const uint32_t rank = (wv ? rdl(c8, wv - 1u) : 0u) + rk_w;
            if (has && rank < nmax) toks[rank] = uint16_t(ppos - sbase);
            if (tid == 0) sc[S_NTOK] = rdl(c8, 7);
        }
        WG_BARRIER();
        const uint32_t ntok = rfl(sc[S_NTOK]);
        uint32_t n = ntok < nmax ? ntok : nmax;
        if (n == 0) { failed = true; break; }                           // (the bitmap has a token at ip: cannot happen)
        pf.add(0, tp);
        // ---- one sequence per thread, fields from the stage (from memory beyond it)
        uint32_t ll = 0, ml = 0, off = 0, lsrc = 0, sz = 0, tpos = 0;
        bool beyond = false;                                             // literals that reach beyond the stage
        const bool seqwave = 64u * wv < n;
        if (tid < n) {
            // the common path reads the stage only; a sequence whose fields reach beyond it (the chunk's last one at most: no token of
            // the chunk begins behind its fields) is decoded from memory and hands its numbers over THROUGH LDS - a register written
            // by a global load and merged here would put a wait for every outstanding global operation (the flush before) on everyone
            const uint32_t p = toks[tid];
            const uint32_t t = stage[p], t1 = stage[p + 1], mn = t & 15u;
            // straight line for what nearly every token is: at most one extension byte per length, fields inside the stage; one
            // branch for everything else (runs of 255, fields beyond the stage)
            const bool x1 = (t >> 4) == 15u, x2 = mn == 15u;
            uint32_t q = p + 1u + (x1 ? 1u : 0u);
            ll = (t >> 4) + (x1 ? t1 : 0u);
            uint32_t mo = q + ll;
            const uint32_t moc = mo + 3u < kStage ? mo : 0u;
            const uint32_t o0 = stage[moc], o1 = stage[moc + 1], e0 = stage[moc + 2], e1 = stage[moc + 3];
            off = o0 | (o1 << 8);
            ml = mn + 4u + (x2 ? e0 : 0u);
            uint32_t q2 = mo + 2u + (x2 ? 1u : 0u);
            if ((x1 && t1 == 255u) || (x2 && e0 == 255u) || mo + 3u >= kStage) {
                bool slow = mo + 3u >= kStage;
                if (x1 && t1 == 255u) {                                  // (the first extension byte was not the last: ll, mo and what was read at mo are void)
                    uint32_t bq = 255u;
                    while (bq == 255u) { if (q >= kStage) { slow = true; break; } bq = stage[q++]; ll += bq; }
                    mo = q + ll; slow = slow || mo + 3u >= kStage;
                    if (!slow) {
                        off = uint32_t(stage[mo]) | (uint32_t(stage[mo + 1]) << 8);
                        const uint32_t f0 = stage[mo + 2], f1 = stage[mo + 3];
                        ml = mn + 4u + (x2 ? f0 : 0u); q2 = mo + 2u + (x2 ? 1u : 0u);
                        if (x2 && f0 == 255u) { uint32_t b2 = f1; ml += b2; q2++; while (b2 == 255u) { if (q2 >= kStage) { slow = true; break; } b2 = stage[q2++]; ml += b2; } }
                    }
                } else if (!slow) {                                      // x2 && e0 == 255
                    uint32_t b2 = e1; ml += b2; q2++;
                    while (b2 == 255u) { if (q2 >= kStage) { slow = true; break; } b2 = stage[q2++]; ml += b2; }
                }
                if (slow) {
                    cgbyte* g = s + sbase;
                    uint32_t L = t >> 4, Q = p + 1;
                    if (L == 15u) { uint32_t qa = sbase + Q; (void)ext_run(s, csize - kMargin, qa, L, 0xFFFFFFFFu); Q = qa - sbase; }
                    const uint32_t MO = Q + L;
                    const uint32_t OF = uint32_t(g[MO]) | (uint32_t(g[MO + 1]) << 8);
                    uint32_t Q2 = MO + 2, M = mn + 4u;
                    if (mn == 15u) { uint32_t qa = sbase + Q2; (void)ext_run(s, csize - kMargin, qa, M, 0xFFFFFFFFu); Q2 = qa - sbase; }
                    sc[S_SLOW] = L; sc[S_SLOW + 1] = M; sc[S_SLOW + 2] = OF; sc[S_SLOW + 3] = Q; sc[S_SLOW + 4] = Q2;
                }
                if (slow) { ll = sc[S_SLOW]; ml = sc[S_SLOW + 1]; off = sc[S_SLOW + 2]; q = sc[S_SLOW + 3]; q2 = sc[S_SLOW + 4]; mo = q + ll; }
            }
            lsrc = q;                                                    // (relative to the stage)
            beyond = ll != 0u && mo > kStage;
            tpos = sbase + p;
            const uint32_t z = ll + ml;                                  // (both below 2^30)
            sz = z > kSzClamp ? kSzClamp : z;
            if (tid == n - 1) sc[S_NEXT] = sbase + q2;                  // where the next chunk begins if it takes all: the token behind the chunk's last
        }
        // ---- placed by a prefix sum over the workgroup
        uint32_t winc = 0;
        if (seqwave) { winc = scan_add(sz); if (lane == 63) sc[S_SUM + wv] = winc; }
        else if (lane == 63) sc[S_SUM + wv] = 0;
        WG_BARRIER();
        if (tid < (128u + kSeqs + kGroup - 1) / kGroup) *reinterpret_cast<u32x4*>(code + u0) = u32x4{0, 0, 0, 0};      // (the bitmap words and the token positions were here)
        uint32_t incl = 0, outl = 0, mst = 0;
        if (seqwave) {
            // (the eight waves' sums: one load, a prefix sum on the DPP network, one lane read - a load per term is a wait per term,
            // a lane read and a scalar add per term is the scalar unit's time)
            const uint32_t w8 = row_add8(sc[S_SUM + (lane & 7u)]);
            const uint32_t wbase = wv ? rdl(w8, wv - 1u) : 0u;
            incl = wbase + winc; outl = incl - sz; mst = outl + ll;      // output start of the literals / of the match, chunk-relative
        }
        // how many sequences the chunk takes: those whose output ends below the output-side margin and - but for the first - inside
        // one tile; an offset beyond the output among them hands the block back
        {
            const bool fits_o = tid < n && opos + incl <= olimit && sz < kSzClamp;
            const bool fits_t = tid < n && (incl <= kTile - 8u || tid == 0u);
            const bool bad = tid < n && (off == 0u || off > opos + mst);
            const unsigned long long fo = __ballot(fits_o), ft = __ballot(fits_t), bmk = __ballot(bad), byb = __ballot(beyond);
            const uint32_t fb = bmk ? 64u * wv + uint32_t(__builtin_ctzll(bmk)) : 0xFFFFu;
            uint32_t tot = 0;
            if (seqwave) tot = scan_max(fits_o && fits_t ? incl : 0u);
            if (lane == 63) {
                sc[S_NF + wv] = uint32_t(__builtin_popcountll(fo)) | (uint32_t(__builtin_popcountll(ft)) << 8) | (byb ? 0x10000u : 0u);      // (both are prefixes)
                sc[S_FB + wv] = fb; sc[S_TOT + wv] = tot;
            }
            if (tid < n) { ent[2 * (3u + 2u * tid)] = 0u - off; ent[2 * (3u + 2u * tid) + 1] = 0u; }
            if (beyond) { sc[S_BYDL] = lsrc - outl; sc[S_BYE] = 2u + 2u * tid; }      // its literals' stage index at chunk output 0, its entry
        }
        WG_BARRIER();
        uint32_t nfo, nft, nbad, total; bool has_beyond;
        {
            const uint32_t l8 = lane & 7u, x8 = sc[S_NF + l8];
            const uint32_t xs = rdl(row_add8(((x8 & 255u) | ((x8 & 0xFF00u) << 8)) + ((x8 >> 16) << 28)), 7);      // counts side by side (each <= 512), "beyond" flags on top
            nfo = xs & 0xFFFFu; nft = (xs >> 16) & 0xFFFu; has_beyond = (xs >> 28) != 0u;
            nbad = rdl(row_min8(sc[S_FB + l8]), 7); total = rdl(row_max8(sc[S_TOT + l8]), 7);
        }
        const uint32_t nfit = nfo < nft ? nfo : nft;
        if (nbad < nfit) { failed = true; break; }
        uint32_t next_ip = rfl(sc[S_NEXT]);
        if (nfit < n) {
            if (nfo <= nft) cut = true;                                  // the output-side tail begins at sequence nfit
            if (tid == nfit) sc[S_CUTPOS] = tpos;                       // (read behind the barriers of the tile loop)
        }
        const uint32_t ntaken = nfit;
        const uint32_t by_dl = has_beyond ? rfl(sc[S_BYDL]) : 0u, by_e = has_beyond ? rfl(sc[S_BYE]) : 0u;
        pf.add(1, tp);
        // ---- the chunk's output, a tile at a time (one tile, unless the first sequence alone is longer)
        bool first_tile = true;
        for (uint32_t R0 = 0; R0 < total;) {
            const uint32_t gb0 = A + opos, mis = gb0 & 7u, gb = (gb0 - mis) & 0xFFFFu;       // ring address of tile coordinate 0
            uint32_t T = total - R0 < kTile - 8u ? total - R0 : kTile - 8u;
            if (gb + mis + T > kRing) T = kRing - gb - mis;                     // (a tile does not wrap around the ring's end)
            const uint32_t ulo = mis, uhi = mis + T;                             // the tile's bytes: tile coordinates [ulo, uhi)
            WG_BARRIER();                                                     // (code[] is zero, the tile before has left sc[])
            if (first_tile) {
                // which token the next chunk begins with is known now: its bytes are loaded while the tile is produced
                if (ntaken < n) next_ip = rfl(sc[S_CUTPOS]);
                if (!cut && next_ip < tail_ip) prefetch(next_ip);
                first_tile = false;
            }
            // marks: where the sequence's literal part and its match part begin inside the tile
            if (R0 == 0u && T == total) {                                // (the chunk is one tile - nearly always: nothing is clipped)
                if (tid < ntaken) {
                    if (ll != 0u) {
                        code[outl + mis] = uint16_t(2u + 2u * tid);
                        ent[2 * (2u + 2u * tid)] = 0x20000u | ((lsrc - mis - outl - gb) & 0xFFFFu);      // the literal at tile coordinate u is stage byte lsrc + (u - mis) - outl
                        ent[2 * (2u + 2u * tid) + 1] = 0x10000u;
                    }
                    code[mst + mis] = uint16_t(3u + 2u * tid);
                }
            } else if (tid < ntaken) {
                const uint32_t me = outl + sz;
                if (ll != 0u && outl < R0 + T && mst > R0) {
                    code[(outl > R0 ? outl : R0) - R0 + mis] = uint16_t(2u + 2u * tid);
                    // the literal at tile coordinate u is stage byte lsrc + (R0 + u - mis) - outl
                    ent[2 * (2u + 2u * tid)] = 0x20000u | ((lsrc + R0 - mis - outl - gb) & 0xFFFFu);
                    ent[2 * (2u + 2u * tid) + 1] = 0x10000u;
                }
                if (mst < R0 + T && me > R0) code[(mst > R0 ? mst : R0) - R0 + mis] = uint16_t(3u + 2u * tid);
            }
            if (tid == kThreads - 1) { if (mis) code[0] = 1; code[uhi] = uint16_t(kEntries - 1u); }
            WG_BARRIER();
            // max-scan: every byte gets the mark in front of it
            uint32_t c[kGroup];
            {
                const u32x4 cv = *reinterpret_cast<const u32x4*>(code + u0);
                c[0] = cv.x & 0xFFFFu; c[1] = cv.x >> 16; c[2] = cv.y & 0xFFFFu; c[3] = cv.y >> 16;
                c[4] = cv.z & 0xFFFFu; c[5] = cv.z >> 16; c[6] = cv.w & 0xFFFFu; c[7] = cv.w >> 16;
#pragma unroll
                for (int k = 1; k < 8; k++) c[k] = umax(c[k], c[k - 1]);
                const uint32_t wi = scan_max(c[7]);
                if (lane == 63) sc[S_SUM + wv] = wi;
                uint32_t ex = dpp0<0x138, 0xf>(wi);                      // wave_shr:1 - the lanes in front (0 for lane 0)
                WG_BARRIER();
                const uint32_t m8 = row_max8(sc[S_SUM + (lane & 7u)]);
                ex = umax(ex, wv ? rdl(m8, wv - 1u) : 0u);
#pragma unroll
                for (int k = 0; k < 8; k++) c[k] = umax(c[k], ex);
            }
            pf.add(2, tp);
            // source pass: where every byte comes from.  Final: literals, matches from in front of the tile, the bytes around the tile
            // (value from LDS: stage or ring; every read of what the ring held happens before the tile's first ring write, behind the
            // next barrier).  Otherwise: a pointer to the byte's source in the tile.
            const bool active = u0 < uhi;
            uint32_t pt[kGroup], bv[kGroup];
#pragma unroll
            for (uint32_t k = 0; k < kGroup; k++) { pt[k] = self[k]; bv[k] = 0; }
            if (active) {
                u32x2 e[kGroup];
#pragma unroll
                for (uint32_t k = 0; k < kGroup; k++) e[k] = *reinterpret_cast<const u32x2*>(ent + 2u * c[k]);
                const uint32_t ub = u0 - ulo, xb = u0 + gb, base2 = Lcode + 2u * ulo;
                uint32_t aa[kGroup];
#pragma unroll
                for (uint32_t k = 0; k < kGroup; k++) {
                    const uint32_t tt = ub + k + e[k].x;                 // (source's tile coordinate) - ulo
                    aa[k] = and_or(xb + k + e[k].x, 0xFFFFu, e[k].y);
                    pt[k] = tt < kTile ? 2u * tt + base2 : self[k];
                }
#pragma unroll
                for (uint32_t k = 0; k < kGroup; k++) bv[k] = ring[aa[k]];
            }
            const uint32_t gaddr = (gb + u0) & 0xFFFFu;
            WG_BARRIER();
            if (active) {
                *reinterpret_cast<u32x2*>(ring + gaddr) = u32x2{bv[0] | (bv[1] << 8) | (bv[2] << 16) | (bv[3] << 24), bv[4] | (bv[5] << 8) | (bv[6] << 16) | (bv[7] << 24)};
                *reinterpret_cast<u32x4*>(code + u0) = u32x4{(pt[0] - Lcode) | ((pt[1] - Lcode) << 16), (pt[2] - Lcode) | ((pt[3] - Lcode) << 16),
                                                             (pt[4] - Lcode) | ((pt[5] - Lcode) << 16), (pt[6] - Lcode) | ((pt[7] - Lcode) << 16)};
                if (has_beyond) {                                        // literals beyond the stage (rare; long runs): from memory into the ring, load and
#pragma unroll                                                           // store inside the branch - nothing a global load wrote is live behind it
                    for (uint32_t k = 0; k < kGroup; k++) {
                        const uint32_t si = by_dl + R0 + (u0 + k) - mis;
                        if (c[k] == by_e && si >= kStage) ring[gaddr + k] = s[sbase + si];
                    }
                }
            }
            WG_BARRIER();
            pf.add(3, tp);
            // chase: a pointer is replaced by the pointer it points to until nothing moves (a final byte points to itself); what a
            // thread has found so far goes back into code[] every round, so that everyone who passes through these bytes jumps ahead
            {
                bool open = active && ((pt[0] ^ self[0]) | (pt[1] ^ self[1]) | (pt[2] ^ self[2]) | (pt[3] ^ self[3]) |
                                       (pt[4] ^ self[4]) | (pt[5] ^ self[5]) | (pt[6] ^ self[6]) | (pt[7] ^ self[7])) != 0u;
                const bool mine = open;
                while (__ballot(open)) {
                    uint32_t q[kGroup];
#pragma unroll
                    for (uint32_t k = 0; k < kGroup; k++) q[k] = pt[k];
                    hop8(q);
                    uint32_t moved = 0;
#pragma unroll
                    for (uint32_t k = 0; k < kGroup; k++) { q[k] += Lcode; moved |= q[k] ^ pt[k]; pt[k] = q[k]; }
                    open = open && moved != 0u;
                    if (open) *reinterpret_cast<u32x4*>(code + u0) = u32x4{(pt[0] - Lcode) | ((pt[1] - Lcode) << 16), (pt[2] - Lcode) | ((pt[3] - Lcode) << 16),
                                                                           (pt[4] - Lcode) | ((pt[5] - Lcode) << 16), (pt[6] - Lcode) | ((pt[7] - Lcode) << 16)};
                }
                if (mine) {
                    // the bytes the pointers ended at are final: in the ring since the barrier (tile coordinate = pointer / 2; no wrap inside a tile)
#pragma unroll
                    for (uint32_t k = 0; k < kGroup; k++) bv[k] = ring[gb + ((pt[k] - Lcode) >> 1)];
                    *reinterpret_cast<u32x2*>(ring + gaddr) = u32x2{bv[0] | (bv[1] << 8) | (bv[2] << 16) | (bv[3] << 24), bv[4] | (bv[5] << 8) | (bv[6] << 16) | (bv[7] << 24)};
                }
            }
            WG_BARRIER();
            pf.add(4, tp);
            // flush: whole 16-byte pieces by ADDRESS; the piece the tile ends in waits for the next tile.  (The next chunk's bytes
            // have arrived by now: waited for HERE, in front of this tile's stores, not behind them.)
            arrived();
            const uint32_t E = opos + T;
            {
                const uint32_t mis16 = (A + flushed) & 15u;
                if (mis16) {                                             // the block's first bytes up to an aligned address
                    uint32_t h = flushed + 16u - mis16; h = h < E ? h : E;
                    if (tid < h - flushed) dst[flushed + tid] = ring[(A + flushed + tid) & 0xFFFFu];
                    flushed = h;
                }
                if (((A + flushed) & 15u) == 0u) {
                    const uint32_t np = (E - flushed) >> 4;
                    if (tid < np) st16g(dst + flushed + 16u * tid, *reinterpret_cast<const u32x4*>(ring + ((A + flushed + 16u * tid) & 0xFFFFu)));
                    flushed += 16u * np;
                }
            }
            *reinterpret_cast<u32x4*>(code + u0) = u32x4{0, 0, 0, 0};
            opos = E; R0 += T;
            pf.add(5, tp); pf.count(6);
        }
        if (first_tile && ntaken < n) { WG_BARRIER(); next_ip = rfl(sc[S_CUTPOS]); }      // (no output at all: the cut was at the chunk's first sequence)
        // the next chunk decodes about as many sequences as fill one tile: more when this one was taken whole and left room, fewer when
        // the tile was full before the chunk's sequences were
        if (ntaken == n && total < kTile - 8u - (kTile >> 3)) nmax = nmax + (nmax >> 2) + 8u;
        else if (ntaken < n) nmax = ntaken + (ntaken >> 4) + 2u;
        nmax = nmax < 8u ? 8u : (nmax > kSeqs ? kSeqs : nmax);
        ip = next_ip;
        WG_BARRIER();                                                 // (sc, code, ent, stage are rewritten by the next chunk)
    }
    if (cut && !failed) res_ip = ip;
    if (!failed) {
        // what the last tile left in the piece it ended in
        if (tid < opos - flushed) dst[flushed + tid] = ring[(A + flushed + tid) & 0xFFFFu];
    }
    pf.dump(meta, kMetaProf + 8, 13, tid == 0);
    if (tid == 0) {
        if (failed) blocks[b].result = lz4par::kRetryCode;
        else { meta[kMetaResIp] = res_ip; meta[kMetaResOp] = opos; blocks[b].result = kResumeCode; }
    }
}

} // namespace

extern "C" size_t fourmc_lz4_tile_work_bytes(uint32_t n) { return size_t(n) * lz4tile::kWsWords * 4u; }
// Blocks per launch pair: 0.53 MB of workspace per block (the token bitmap of the largest stream), so the 16 384 blocks of a
// 64 GiB launch are ONE piece of 8.6 GB; on a smaller device the pieces stay below 20 % of its memory.  FOURMC_TILE_BATCH overrides.
extern "C" uint32_t fourmc_lz4_tile_batch(void)
{
    static const uint32_t batch = [] {
        uint32_t b = 16384;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) {
            const size_t fit = tot / 5 / (size_t(lz4tile::kWsWords) * 4u);
            if (fit < b) b = fit < 64 ? 64u : uint32_t(fit);
        }
        if (const char* e = getenv("FOURMC_TILE_BATCH")) { const long x = atol(e); if (x > 0) b = uint32_t(x); }
        return b;
    }();
    return batch;
}

// FOURMC_TILE_WALK=separate: the walk as a kernel of its own, one wave per block (what a launch of many thousand blocks would
// rather have: 64 chains per block are enough to fill the chip then); default: fused into the executor's workgroup
static bool tile_walk_fused()
{
    static const bool v = [] { const char* e = getenv("FOURMC_TILE_WALK"); return !(e && !strcmp(e, "separate")); }();
    return v;
}
extern "C" hipError_t fourmc_launch_lz4_tile(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                             int container_mode, void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (tile_walk_fused()) {
        hipLaunchKernelGGL(lz4_tile_exec_kernel<true>, dim3(n), dim3(lz4tile::kThreads), 0, stream, static_cast<const uint8_t*>(d_src),
                           static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, static_cast<uint32_t*>(d_work));
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lz4_tile_walk_kernel, dim3(n), dim3(64), 0, stream, static_cast<const uint8_t*>(d_src), d_blocks, n,
                       container_mode, static_cast<uint32_t*>(d_work));
    hipLaunchKernelGGL(lz4_tile_exec_kernel<false>, dim3(n), dim3(lz4tile::kThreads), 0, stream, static_cast<const uint8_t*>(d_src),
                       static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, static_cast<uint32_t*>(d_work));
    return hipGetLastError();
}
