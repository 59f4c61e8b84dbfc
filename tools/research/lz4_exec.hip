// 4mc_amd/csrc/lz4_exec.hip - K1x: executes the sequence records of lz4_parse.hip; the LZ77 copy loop of the reference
// (native/lz4/lz4.c:2060-2110, :2300-2325) over a 16 KiB window of the output held in LDS.
//
// One workgroup of three waves per block, output produced in WINDOWS of 1 KiB (lz4par.h):
//   * the LITERAL wave walks the windows a few ahead of the chain wave.  A lane is one sequence (clipped to the window):
//     it decodes its token from the wave's staged copy of the stream (prefetched a window ahead), copies its literals
//     into the ring, copies its match straight from the block's flushed output in HBM when the source has left the part
//     of the ring that is guaranteed to be intact, and queues every other match (<= 64 per slot) for the chain wave.
//     Nothing it does depends on recent output.
//   * the CHAIN wave executes the queued matches strictly in order.  It reads and writes the LDS only; the LDS executes
//     one wave's accesses in order, so a copy sees every earlier copy without any flag or wait.  Inside a slot the entries
//     whose source lies below the first entry still missing run together; dependency chains (the usual shape of
//     structured data) degenerate to one or two entries per step, which the whole wave copies byte-parallel.
//   * the FLUSH wave writes completed windows to HBM with aligned 16-byte stores and publishes F_vis.
// All irregular, byte-granular accesses stay in the LDS (byte-unaligned wider LDS accesses cost 64 clk per instruction
// on gfx950, byte accesses 2-4: profiles/r02_ubench_lds_unaligned.txt); loads are always issued in batches before the
// stores that depend on them.  HBM sees the coalesced stream reads, the far match gathers and the 16-byte flush stores.
// Ring reuse: window w overwrites window w - kRW, which has been flushed and which no reader may touch any more because
// readers never reach further back than kRW - kAhead - 1 windows through the ring.  Every wait is bounded; a wave that
// waits too long aborts the block to the exact kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"

using namespace lz4par;

namespace {

constexpr int kRW    = 16;                    // windows in the ring
constexpr int kRing  = kRW * kWin;
constexpr uint32_t kRM = kRing - 1;
constexpr int kAhead = 3;                     // windows the literal wave may be ahead of the chain wave
constexpr int kLag   = 6;                     // flush stores in flight before the oldest one is waited for
constexpr int kGuard = 64;                    // readable bytes behind the ring / the staged stream (batched reads overshoot)
constexpr int kShort = 32;                    // pieces up to this length are copied by their own lane
constexpr int kPF    = 4;                     // dwords prefetched per piece by the literal wave (longer pieces: whole-wave copy on demand)
constexpr int kNS    = 4;                     // slots between the literal wave and the chain wave
constexpr uint32_t kSpinLimit = 1u << 21;
constexpr int kXT    = 64 * 3;


// sync words: relaxed workgroup-scope atomics (a volatile access makes the backend wait for every single load / store)
// (the value is wave-uniform: readfirstlane moves it to an SGPR, so that loops on it are scalar branches, not exec-mask loops)
__device__ __forceinline__ uint32_t ldv(const uint32_t* p) { return uint32_t(__builtin_amdgcn_readfirstlane(int(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)))); }
__device__ __forceinline__ void stv(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t xdpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t xscan_add(uint32_t v)
{
    v += xdpp0<0x111, 0xf>(v); v += xdpp0<0x112, 0xf>(v); v += xdpp0<0x114, 0xf>(v); v += xdpp0<0x118, 0xf>(v);
    v += xdpp0<0x142, 0xa>(v); v += xdpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t xscan_max(uint32_t v)      // values >= 0
{
    v = max(v, xdpp0<0x111, 0xf>(v)); v = max(v, xdpp0<0x112, 0xf>(v)); v = max(v, xdpp0<0x114, 0xf>(v)); v = max(v, xdpp0<0x118, 0xf>(v));
    v = max(v, xdpp0<0x142, 0xa>(v)); v = max(v, xdpp0<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); }

typedef uint8_t* ring_t;       // LDS; the hot copies address it through 32-bit LDS addresses and inline asm
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return uint32_t(uintptr_t((const __attribute__((address_space(3))) void*)p)); }

// profiling build (make prof, tools/k1x_prof.py): per wave, cycles spent per section; PT(i) charges the time since the
// previous mark to counter i
#ifdef K1X_PROF
struct Prof { unsigned long long t[8]; unsigned long long last; };
#define PROF_DECL Prof prof_; for (int i_ = 0; i_ < 8; i_++) prof_.t[i_] = 0; prof_.last = __builtin_readcyclecounter();
#define PT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); prof_.t[i] += n_ - prof_.last; prof_.last = n_; } while (0)
#define PADD(i, v) do { prof_.t[i] += (v); } while (0)
#define PROF_OUT(B, wave, lane) do { if ((lane) == 0) { unsigned long long* d_ = (B).dbg + 8 * (wave); for (int i_ = 0; i_ < 8; i_++) d_[i_] = prof_.t[i_]; } } while (0)
#else
#define PROF_DECL
#define PT(i) do {} while (0)
#define PADD(i, v) do {} while (0)
#define PROF_OUT(B, wave, lane) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------- copy primitives
// Byte loads / stores of the LDS as inline asm: plain C byte accesses get merged into byte-unaligned dword accesses (64 clk
// each on gfx950), volatile ones are waited for one by one.  d16 / d16_hi halves: two bytes per VGPR, no packing ALU.
template <int OFF> __device__ __forceinline__ void st_lo(uint32_t a, uint32_t r) { asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(a), "v"(r), "n"(OFF) : "memory"); }
template <int OFF> __device__ __forceinline__ void st_hi(uint32_t a, uint32_t r) { asm volatile("ds_write_b8_d16_hi %0, %1 offset:%2" :: "v"(a), "v"(r), "n"(OFF) : "memory"); }
#include "ldscopy.inc"

// m <= 32 bytes per lane from LDS address s to LDS address d, not overlapping, s readable up to 34 bytes past its start
// whatever m is (also for lanes with m = 0).  All loads of all lanes are issued before the first store.
// G (wave-uniform) = m / 4 of the longest piece, or -1 to find out here.
__device__ __forceinline__ void copy_upto32_lds(uint32_t d, uint32_t s, uint32_t m, int G = -1)
{
    if (G < 0) {
        if (__ballot(m != 0) == 0) return;
        G = 0;
#pragma unroll
        for (int g = 1; g <= 8; g++) if (__ballot(m >= uint32_t(4 * g)) != 0) G = g;
    }
    switch (G) {
        case 0: cp32_g0(d, s, m); break;
        case 1: cp32_g1(d, s, m); break;
        case 2: cp32_g2(d, s, m); break;
        case 3: cp32_g3(d, s, m); break;
        case 4: cp32_g4(d, s, m); break;
        case 5: cp32_g5(d, s, m); break;
        case 6: cp32_g6(d, s, m); break;
        case 7: cp32_g7(d, s, m); break;
        default: cp32_g8(d, s, m); break;
    }
}

// the same from HBM (g valid for every lane): unaligned dword loads, byte stores
__device__ __forceinline__ void copy_upto32_hbm(uint32_t d, const uint8_t* g, uint32_t m)
{
    if (__ballot(m != 0) == 0) return;
    uint32_t w[8], tl[3];
    const uint32_t t0 = m & ~3u;
    struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
#pragma unroll
    for (int q = 0; q < 8; q++) { w[q] = 0; if (__ballot(m >= uint32_t(4 * q + 4)) != 0) w[q] = reinterpret_cast<const U4*>(g + 4 * q)->v; }
#pragma unroll
    for (int i = 0; i < 3; i++) tl[i] = g[t0 + i];
#define FOURMC_ST4(q) if (m >= uint32_t(4 * q + 4)) { st_lo<4 * q>(d, w[q]); st_lo<4 * q + 1>(d, w[q] >> 8); st_hi<4 * q + 2>(d, w[q]); st_hi<4 * q + 3>(d, w[q] >> 8); }
    FOURMC_ST4(0) FOURMC_ST4(1) FOURMC_ST4(2) FOURMC_ST4(3) FOURMC_ST4(4) FOURMC_ST4(5) FOURMC_ST4(6) FOURMC_ST4(7)
#undef FOURMC_ST4
    const uint32_t tm = m & 3u, dt = d + t0;
    if (tm > 0) st_lo<0>(dt, tl[0]);
    if (tm > 1) st_lo<1>(dt, tl[1]);
    if (tm > 2) st_lo<2>(dt, tl[2]);
}

// one lane, n <= kShort bytes inside the ring with LZ4 (byte-serial) semantics dst[k] = dst[k - off]: copied in steps
// whose distance doubles (off, 2 off, ..: always a multiple of the period), so that every step is a plain copy
__device__ __forceinline__ void lane_copy_ring(ring_t ring, uint32_t dst, uint32_t off, uint32_t n, int G = -1)
{
    const uint32_t ra = lds_addr(ring);
    uint32_t done = 0, span = off;
    while (__ballot(done < n)) {
        const uint32_t m = done < n ? min(span, n - done) : 0u;
        const uint32_t sa = (dst + done - span) & kRM, da = (dst + done) & kRM;
        const bool wraps = m && sa + m > uint32_t(kRing);
        copy_upto32_lds(ra + da, ra + sa, wraps ? 0u : m, G);
        if (__ballot(wraps)) { if (wraps) for (uint32_t k = 0; k < m; k++) { const uint8_t a = ring[(sa + k) & kRM]; cbar(); ring[da + k] = a; cbar(); } }
        done += m; span <<= 1;
    }
}

// whole wave, one piece of any length inside the ring (wave-uniform arguments)
__device__ __forceinline__ void wave_copy_ring(ring_t ring, uint32_t dst, uint32_t off, uint32_t n, int lane)
{
    if (off >= 256 || off >= n) {
        for (uint32_t k0 = 0; k0 < n; k0 += 256) {
            uint8_t a[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { const uint32_t k = k0 + 64 * i + lane; a[i] = ring[(dst - off + k) & kRM]; }
            cbar();
#pragma unroll
            for (int i = 0; i < 4; i++) { const uint32_t k = k0 + 64 * i + lane; if (k < n) ring[(dst + k) & kRM] = a[i]; }
            cbar();
        }
        return;
    }
    if (off >= 64) {
        for (uint32_t k = lane; k < n; k += 64) { const uint8_t a = ring[(dst - off + k) & kRM]; cbar(); ring[(dst + k) & kRM] = a; cbar(); }
        return;
    }
    // overlapping: the output is periodic; D = the smallest multiple of off that is >= 64 keeps every later step a plain copy
    const uint32_t D = off * ((63u + off) / off);
    for (uint32_t k = lane; k < min(n, D); k += 64) { const uint8_t a = ring[(dst - off + (k % off)) & kRM]; cbar(); ring[(dst + k) & kRM] = a; }
    cbar();
    for (uint32_t k0 = D; k0 < n; k0 += 64) {
        const uint32_t k = k0 + lane;
        if (k < n) { const uint8_t a = ring[(dst + k - D) & kRM]; cbar(); ring[(dst + k) & kRM] = a; }
        cbar();
    }
}

// whole wave: n bytes from HBM (wave-uniform arguments) into the ring at dst (inside one window)
__device__ __forceinline__ void wave_copy_hbm(ring_t ring, uint32_t dst, const uint8_t* g, uint32_t n, int lane)
{
    for (uint32_t k0 = 0; k0 < n; k0 += 256) {
        uint8_t a[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t k = k0 + 64 * i + lane; a[i] = k < n ? g[k] : uint8_t(0); }
        cbar();
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t k = k0 + 64 * i + lane; if (k < n) ring[(dst + k) & kRM] = a[i]; }
    }
}

// ---------------------------------------------------------------------------------------------- the three waves
struct Blk {
    const uint8_t* src; uint8_t* dst; const uint4* wdesc; const uint2* rec; unsigned long long* dbg;
    uint32_t iend, nseq, total, nwin, a0;
};

struct Slot {                                 // the ring matches of (at most) 64 consecutive sequences, in order
    uint32_t n, epos, last, maxlen;
    uint32_t dst[64];
    uint32_t ol[64];                          // offset | length << 16
};
struct XSync {
    uint32_t E_win;                           // windows whose every byte is in the ring
    uint32_t F_vis;                           // windows flushed to HBM and visible
    uint32_t ready;                           // slots published by the literal wave
    uint32_t consumed;                        // slots the chain wave is done with
    uint32_t abort;
};

__device__ __forceinline__ bool spin_fail(XSync* sy, uint32_t& spins)
{
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit) { stv(&sy->abort, 1); return true; }
    return ldv(&sy->abort) != 0;
}

// ------------------------------------------------------------------------------------------------ literal wave
// Walks the windows in order, up to kAhead windows ahead of the chain wave.  A batch is 64 consecutive sequences of a
// window (records decoded by the parser, clipped to the window).  Software pipeline: while batch k is stored into the
// ring, batch k + 1 is already unpacked and its HBM loads - literal bytes from the stream, bytes of matches whose source has
// left the ring's guaranteed part from the flushed output - are in flight; the wave never waits for a load it has just
// issued.  Every other match goes into the batch's slot for the chain wave.
struct Batch {                                  // per lane unless marked uniform
    uint32_t w, W1, lastb, valid;               // uniform: window, its end, last batch of the window
    uint32_t ls, ln;                            // literal piece: destination (shifted position), length
    uint32_t ds, off, mn, s0;                   // match piece: destination, offset, length, source
    uint32_t cp;                                // stream position of the literal piece
    uint32_t far;                               // the match is copied from HBM here
    uint32_t epos;                              // uniform: completed position once the batch is in the ring
    uint32_t L[kPF], F[kPF];                    // first 4 * kPF bytes of the literal piece / of the far match source
};

struct __attribute__((packed, aligned(1))) U4u { uint32_t v; };
// bytes [0, m) of the dwords w -> LDS address d (m <= 4 * kPF)
__device__ __forceinline__ void store_upto32(uint32_t d, const uint32_t (&w)[kPF], uint32_t m)
{
    if (__ballot(m != 0) == 0) return;
#define FOURMC_ST4(q) if (__ballot(m > uint32_t(4 * q)) != 0) { \
        if (m >= uint32_t(4 * q + 4)) { st_lo<4 * q>(d, w[q]); st_lo<4 * q + 1>(d, w[q] >> 8); st_hi<4 * q + 2>(d, w[q]); st_hi<4 * q + 3>(d, w[q] >> 8); } \
        else if (m > uint32_t(4 * q)) { st_lo<4 * q>(d, w[q]); if (m > uint32_t(4 * q + 1)) st_lo<4 * q + 1>(d, w[q] >> 8); if (m > uint32_t(4 * q + 2)) st_hi<4 * q + 2>(d, w[q]); } }
    FOURMC_ST4(0) FOURMC_ST4(1) FOURMC_ST4(2) FOURMC_ST4(3)
    static_assert(kPF == 4, "store_upto32 is written for four dwords");
#undef FOURMC_ST4
}

__device__ __forceinline__ void literal_wave(const Blk& B, ring_t ring, Slot* slots, XSync* sy, int lane)
{
    const uint32_t endp = B.total + B.a0;
    const uint32_t ra = lds_addr(ring);
    uint32_t produced = 0;
    PROF_DECL
    // window-level prefetch: descriptors two windows ahead, the records of a window's first three batches one window ahead
    auto load_desc = [&](uint32_t w) -> uint4 {                              // lanes 0..3: A(w), B(w), A(w+1), B(w+1)
        const uint32_t i = min(2 * w + uint32_t(lane & 3), 2 * B.nwin + 1);
        return B.wdesc[i];
    };
    struct Recs { uint2 r[3]; };
    auto load_recs = [&](const uint4& dl, Recs& d) {
        const uint32_t first = rl(dl.x, 0);
#pragma unroll
        for (int q = 0; q < 3; q++) { const uint32_t i = first + 64u * q + lane; d.r[q] = i < B.nseq ? B.rec[i] : make_uint2(0, 0); }
    };
    // iteration state
    uint32_t w = 0, j0 = 0, obase = 0;
    uint4 dcur = load_desc(0), dnext = load_desc(1), dnn = dnext;
    Recs cur; load_recs(dcur, cur);
    Recs nxt = cur;
    bool win_fresh = true;                      // the first batch of window w is next

    // unpack batch (w, j0), clip it to the window, issue its HBM loads
    auto prepare = [&](Batch& b) {
        if (win_fresh) {
            win_fresh = false;
            if (w + 1 < B.nwin) { load_recs(dnext, nxt); dnn = load_desc(w + 2); }
            obase = rl(dcur.y, 0);
        }
        const uint32_t first = rl(dcur.x, 0), last = min(rl(dcur.x, 2), B.nseq - 1);
        const uint32_t W0 = w << kWinLog, W1 = min(W0 + uint32_t(kWin), endp);
        const int lbw = int(w) + kAhead - kRW + 1;
        const uint32_t lowb = lbw > 0 ? uint32_t(lbw) << kWinLog : 0u;     // the ring is guaranteed from here on
        const uint32_t sidx = first + j0 + lane;
        const bool act = sidx <= last;
        uint2 r = j0 == 0 ? cur.r[0] : j0 == 64 ? cur.r[1] : j0 == 128 ? cur.r[2] : (act ? B.rec[sidx] : make_uint2(0, 0));
        uint32_t litpos = r.x & 0x7fffffu, ll = (r.x >> 23) | ((r.y >> 27) << 9), ml = (r.y >> 16) & 2047u, off = r.y & 0xffffu;
        if (ll == kRecLLSat || ml == kRecMLSat) {                           // at least a window long: the last of this window, exact in the next descriptor
            litpos = rl(dcur.w, 2); ll = rl(dcur.x, 3); ml = rl(dcur.y, 3); off = rl(dcur.z, 3);
        }
        if (j0 == 0 && lane == 0) { litpos = rl(dcur.w, 0); ll = rl(dcur.x, 1); ml = rl(dcur.y, 1); off = rl(dcur.z, 1); }   // the window's first sequence
        if (!act) { ll = 0; ml = 0; }
        const uint32_t len = ll + ml;
        const uint32_t incl = xscan_add(len);
        const uint32_t sp = obase + (incl - len) + B.a0;                     // shifted output position of the sequence
        obase += rl(incl, 63);
        b.w = w; b.W1 = W1; b.valid = 1;
        b.lastb = first + j0 + 64 > last ? 1u : 0u;
        b.epos = b.lastb ? W1 : min(W1, obase + B.a0);
        // literal piece
        const uint32_t ls = max(sp, W0), le = min(sp + ll, W1);
        b.ls = ls; b.ln = le > ls ? le - ls : 0u;
        b.cp = litpos + (ls - sp);
        // match piece
        const uint32_t mstart = sp + ll;
        const uint32_t ds = max(mstart, W0), de = min(mstart + ml, W1);
        b.ds = ds; b.off = off; b.mn = (ml && de > ds) ? de - ds : 0u; b.s0 = ds - off;
        b.far = (b.mn && b.s0 < lowb) ? 1u : 0u;                             // never overlapping: off > kWin >= mn
        // loads: up to 32 literal bytes (clamped inside the stream; what lies behind the piece is never stored)
        {
            const uint32_t G = b.ln <= 4u * kPF ? (b.ln + 3) >> 2 : 0u;
            const uint8_t* g = B.src + b.cp;
#pragma unroll
            for (int q = 0; q < kPF; q++) { b.L[q] = 0; if (__ballot(G > uint32_t(q)) != 0) { if (G > uint32_t(q)) b.L[q] = reinterpret_cast<const U4u*>(b.cp + 4 * q + 4 <= B.iend ? g + 4 * q : B.src + B.iend - 4)->v; } }
        }
        if (__ballot(b.far)) {
            uint32_t need = b.far ? ((b.s0 + b.mn - 1) >> kWinLog) + 1 : 0;
            need = rl(xscan_max(need), 63);
            for (uint32_t spins = 0; ldv(&sy->F_vis) < need; ) if (spin_fail(sy, spins)) { b.valid = 0; return; }
            cbar();
            const uint32_t G = (b.far && b.mn <= 4u * kPF) ? (b.mn + 3) >> 2 : 0u;
            const uint8_t* g = B.dst + (b.far ? b.s0 - B.a0 : 0u);
#pragma unroll
            for (int q = 0; q < kPF; q++) { b.F[q] = 0; if (__ballot(G > uint32_t(q)) != 0) { if (G > uint32_t(q)) b.F[q] = reinterpret_cast<const U4u*>(g + 4 * q)->v; } }
        }
        // advance
        j0 += 64;
        if (first + j0 > last) { w++; j0 = 0; dcur = dnext; dnext = dnn; cur = nxt; win_fresh = true; }
    };

    // first touch of what prepare() loaded: the wait for those loads sits here, after the OTHER batch's stores
    // (a piece that ends in the last 3 bytes of the stream has its tail loaded a few bytes early: shift it back)
    auto finalize = [&](Batch& b) {
        const uint32_t G = b.ln <= 4u * kPF ? (b.ln + 3) >> 2 : 0u;
#pragma unroll
        for (int q = 0; q < kPF; q++) if (G > uint32_t(q) && b.cp + 4 * q + 4 > B.iend) b.L[q] >>= 8 * (b.cp + 4 * q + 4 - B.iend);
        asm volatile("" : "+v"(b.L[0]), "+v"(b.F[0]));
    };

    // store batch b into the ring and publish its slot
    auto commit = [&](Batch& b) -> bool {
        PT(1);
        for (uint32_t spins = 0; ldv(&sy->E_win) + kAhead < b.w || ldv(&sy->F_vis) + kRW <= b.w; ) if (spin_fail(sy, spins)) return false;
        for (uint32_t spins = 0; produced >= ldv(&sy->consumed) + kNS; ) if (spin_fail(sy, spins)) return false;
        cbar();
        PT(0);
        // literals
        {
            const bool sh = b.ln <= 4u * kPF;
            store_upto32(ra + (b.ls & kRM), b.L, sh ? b.ln : 0u);
            unsigned long long lg = __ballot(!sh);
            while (lg) {
                const int l = __builtin_ctzll(lg); lg &= lg - 1;
                wave_copy_hbm(ring, rl(b.ls, l), B.src + rl(b.cp, l), rl(b.ln, l), lane);
            }
        }
        PT(3);
        // matches from HBM
        if (__ballot(b.far)) {
            const bool sh = b.far && b.mn <= 4u * kPF;
            store_upto32(ra + (b.ds & kRM), b.F, sh ? b.mn : 0u);
            unsigned long long lg = __ballot(b.far && !sh);
            while (lg) {
                const int l = __builtin_ctzll(lg); lg &= lg - 1;
                wave_copy_hbm(ring, rl(b.ds, l), B.dst + rl(b.s0, l) - B.a0, rl(b.mn, l), lane);
            }
        }
        PT(4);
        // the rest goes to the chain wave
        Slot* s = slots + (produced % kNS);
        const bool near = b.mn && !b.far;
        const unsigned long long nb = __ballot(near);
        if (near) {
            const uint32_t idx = __builtin_amdgcn_mbcnt_hi(uint32_t(nb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(nb), 0));
            s->dst[idx] = b.ds; s->ol[idx] = b.off | (b.mn << 16);
        }
        const uint32_t mx = rl(xscan_max(near ? b.mn : 0u), 63);
        if (lane == 0) { s->n = uint32_t(__builtin_popcountll(nb)); s->epos = b.epos; s->last = b.lastb; s->maxlen = mx; }
        lds_fence();                                  // every byte of the batch is in the LDS before the slot is published
        produced++;
        if (lane == 0) stv(&sy->ready, produced);
        PT(5);
        return true;
    };

    Batch a, b2;                                 // ping-pong: a register copy of a batch would wait for its loads
    prepare(a); finalize(a);
    for (;;) {
        bool more = w < B.nwin;
        if (more) prepare(b2);
        if (!a.valid || !commit(a)) return;
        if (!more) break;
        finalize(b2);
        more = w < B.nwin;
        if (more) prepare(a);
        if (!b2.valid || !commit(b2)) return;
        if (!more) break;
        finalize(a);
    }
    PROF_OUT(B, 0, lane);
}

// ------------------------------------------------------------------------------------------------ chain wave
// Executes the slots strictly in order.  Everything below a slot's first entry that is not one of its entries is final;
// inside a slot an entry may run once no earlier entry that is still missing can overlap its source (entries are sorted by
// destination and disjoint, so "nothing missing below the first missing entry's destination" is the test).
__device__ __forceinline__ uint32_t chain_slot(ring_t ring, const Slot* s, uint32_t n, uint32_t maxlen, int lane)
{
    const uint32_t ra = lds_addr(ring);
    const bool act = uint32_t(lane) < n;
    const uint32_t dst = act ? s->dst[lane] : 0xffffffffu;
    const uint32_t ol = act ? s->ol[lane] : 0;
    const uint32_t off = ol & 0xffff, len = ol >> 16;
    const uint32_t hi = min(dst - off + len, dst);              // end of the part of the source that others produce
    const float flane = float(lane) + 0.5f;
    unsigned long long undone = __ballot(act);
    // 1) everything whose source ends below the slot's first entry is independent of the slot: one batched copy
    if (maxlen <= uint32_t(kShort)) {
        const uint32_t D0 = rl(dst, 0);
        const bool ready = act && (hi <= D0 || lane == 0);
        const unsigned long long rb = __ballot(ready);
        if (__builtin_popcountll(rb) > 2) {
            lane_copy_ring(ring, dst, off, ready ? len : 0u, int(maxlen >> 2));
            undone &= ~rb;
        }
    }
    // 2) what is left depends on entries of this slot: strictly in order, the whole wave copies one entry byte-parallel
    //    (byte k of an entry = source byte k mod off, and all of [dst - off, dst) is final when its turn comes)
    uint32_t steps = 0;
    while (undone) {
        const int l = __builtin_ctzll(undone); undone &= undone - 1;
        const uint32_t d0 = rl(dst, l), oo = rl(off, l), nn = rl(len, l);
        steps++;
        if (nn > 64) { wave_copy_ring(ring, d0, oo, nn, lane); continue; }
        uint32_t km = uint32_t(lane);
        if (oo < nn) {                                          // overlapping: lane mod off (exact: lane, off < 64)
            const uint32_t qd = uint32_t(flane * __builtin_amdgcn_rcpf(float(oo)));
            km = uint32_t(lane) - qd * oo;
        }
        const uint32_t sa = ra + ((d0 - oo + km) & kRM), da = ra + ((d0 + uint32_t(lane)) & kRM);
        uint32_t v;
        asm volatile("s_nop 1\n\tds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(sa) : "memory");
        if (uint32_t(lane) < nn) st_lo<0>(da, v);
    }
    return steps;
}

__device__ __forceinline__ void chain_wave(const Blk& B, ring_t ring, Slot* slots, XSync* sy, int lane)
{
    uint32_t cons = 0;
    __builtin_amdgcn_s_setprio(3);
    PROF_DECL
    for (uint32_t w = 0; w < B.nwin; w++) {
        w = rfl(w);
        for (;;) {
            cons = rfl(cons);
            for (uint32_t spins = 0; ldv(&sy->ready) <= cons; ) if (spin_fail(sy, spins)) return;
            cbar();
            PT(0);
            const Slot* s = slots + (cons % kNS);
            const uint32_t n = ldv(&s->n), last = ldv(&s->last), maxlen = ldv(&s->maxlen);
            if (n) { const uint32_t steps = chain_slot(ring, s, n, maxlen, lane); (void)steps; PADD(3, steps); PADD(4, n); }
            PADD(5, 1);
            lds_fence();
            PT(1);
            cons++;
            if (lane == 0) { stv(&sy->consumed, cons); if (last) stv(&sy->E_win, w + 1); }
            PT(2);
            if (last) break;
        }
    }
    PROF_OUT(B, 1, lane);
}

// ------------------------------------------------------------------------------------------------ flush wave
__device__ __forceinline__ void flush_wave(const Blk& B, ring_t ring, XSync* sy, int lane)
{
    const uint32_t endp = B.total + B.a0;
    PROF_DECL
    for (uint32_t f = 0; f < B.nwin; f++) {
        PT(1);
        for (uint32_t spins = 0; ldv(&sy->E_win) <= f; ) if (spin_fail(sy, spins)) return;
        cbar();
        PT(0);
        const uint32_t p0 = (f << kWinLog) + 16u * lane;
        const uint4 v = *reinterpret_cast<const uint4*>(const_cast<const uint8_t*>(ring) + (p0 & kRM));   // behind the barrier above
        uint8_t* g = B.dst + p0 - B.a0;        // 16-byte aligned by construction of a0
        if (p0 >= B.a0 && p0 + 16 <= endp) *reinterpret_cast<uint4*>(g) = v;
        else {
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t wv = k < 4 ? v.x : k < 8 ? v.y : k < 12 ? v.z : v.w;
                if (p0 + k >= B.a0 && p0 + k < endp) g[k] = uint8_t(wv >> (8 * (k & 3)));
            }
        }
        if (f == 0 || f + 1 == B.nwin) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) stv(&sy->F_vis, f + 1); }
        else { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); if (lane == 0 && f >= uint32_t(kLag)) stv(&sy->F_vis, f + 1 - kLag); }
    }
    PT(1);
    PROF_OUT(B, 2, lane);
}

} // namespace

__global__ __launch_bounds__(kXT, 6)
void lz4_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                     const uint8_t* work)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing + kGuard];
    __shared__ Slot slots[kNS];
    __shared__ XSync sy;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint8_t* slot = work + size_t(b) * kSlotBytes;
    const ParHdr* hdr = reinterpret_cast<const ParHdr*>(slot);
    if (hdr->status != kParsed) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    Blk B;
    B.src = src_base + blk.src_off; B.dst = dst_base + blk.dst_off;
    B.wdesc = reinterpret_cast<const uint4*>(slot + kWdescOff);
    B.rec = reinterpret_cast<const uint2*>(slot + kTokOff);
    B.dbg = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(slot) + kDbgOff);
    // wave-uniform by construction; loaded per lane, so tell the compiler (scalar loop control instead of exec-mask loops)
    auto uni = [](uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); };
    B.iend = uni(blk.src_len); B.nseq = uni(hdr->nseq); B.total = uni(hdr->total); B.nwin = uni(hdr->nwin); B.a0 = uni(hdr->a0);
    if (threadIdx.x == 0) { sy.E_win = 0; sy.F_vis = 0; sy.ready = 0; sy.consumed = 0; sy.abort = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (wave == 0) literal_wave(B, ring, slots, &sy, lane);
    else if (wave == 1) chain_wave(B, ring, slots, &sy, lane);
    else {
        flush_wave(B, ring, &sy, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) blocks[b].result = ldv(&sy.abort) ? kRetryCode : int(B.total);
    }
}

extern "C" hipError_t fourmc_launch_lz4_exec(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                              const void* d_work, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_exec_kernel, dim3(n), dim3(kXT), 0, stream, static_cast<const uint8_t*>(d_src),
                       static_cast<uint8_t*>(d_dst), d_blocks, n, static_cast<const uint8_t*>(d_work));
    return hipGetLastError();
}
