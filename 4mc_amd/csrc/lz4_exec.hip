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
constexpr int kCB    = 2048;                  // bytes of the stream staged per window
constexpr int kGuard = 64;                    // readable bytes behind the ring / the staged stream (batched reads overshoot)
constexpr int kShort = 32;                    // pieces up to this length are copied by their own lane
constexpr int kNS    = 4;                     // slots between the literal wave and the chain wave
constexpr uint32_t kSpinLimit = 1u << 21;
constexpr int kXT    = 64 * 3;


// sync words: relaxed workgroup-scope atomics (a volatile access makes the backend wait for every single load / store)
__device__ __forceinline__ uint32_t ldv(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
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
__device__ __forceinline__ void copy_upto32_lds(uint32_t d, uint32_t s, uint32_t m)
{
    if (__ballot(m != 0) == 0) return;
    int G = 0;
#pragma unroll
    for (int g = 1; g <= 8; g++) if (__ballot(m >= uint32_t(4 * g)) != 0) G = g;
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
__device__ __forceinline__ void lane_copy_ring(ring_t ring, uint32_t dst, uint32_t off, uint32_t n)
{
    const uint32_t ra = lds_addr(ring);
    uint32_t done = 0, span = off;
    while (__ballot(done < n)) {
        const uint32_t m = done < n ? min(span, n - done) : 0u;
        const uint32_t sa = (dst + done - span) & kRM, da = (dst + done) & kRM;
        const bool wraps = m && sa + m > uint32_t(kRing);
        copy_upto32_lds(ra + da, ra + sa, wraps ? 0u : m);
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
    const uint8_t* src; uint8_t* dst; const uint4* wdesc; const uint32_t* tok; unsigned long long* dbg;
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

typedef const __attribute__((address_space(3))) uint8_t* lds_bytes;   // keeps the two sides of the choice below apart (no flat loads)
// stream byte at position p: from the wave's staged copy [cs, cs + kCB) or from HBM
__device__ __forceinline__ uint32_t sbyte(const Blk& B, const uint8_t* cbuf, uint32_t cs, uint32_t p)
{
    const uint32_t i = p - cs;
    if (i < uint32_t(kCB)) return ((lds_bytes)cbuf)[i];
    return B.src[p];
}

__device__ __forceinline__ bool spin_fail(XSync* sy, uint32_t& spins)
{
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpinLimit) { stv(&sy->abort, 1); return true; }
    return ldv(&sy->abort) != 0;
}

// ------------------------------------------------------------------------------------------------ literal wave
// Walks the windows in order, up to kAhead windows ahead of the chain wave: token decode, literals into the ring, matches
// whose source has left the ring's guaranteed part straight from the flushed output, and for every batch of 64 sequences
// one slot with the matches that read the ring.
__device__ __forceinline__ void literal_wave(const Blk& B, ring_t ring, uint8_t* cbuf, Slot* slots, XSync* sy, int lane)
{
    const uint32_t endp = B.total + B.a0;
    uint32_t produced = 0;
    PROF_DECL
    // prefetch registers: descriptors two windows ahead, stream bytes and token positions one window ahead
    auto load_desc = [&](uint32_t w) -> uint4 {                              // lanes 0..3: A(w), B(w), A(w+1), B(w+1)
        const uint32_t i = min(2 * w + uint32_t(lane & 3), 2 * B.nwin + 1);
        return B.wdesc[i];
    };
    struct Data { uint4 c[kCB / 1024]; uint32_t t[3]; };
    auto load_data = [&](const uint4& dl, Data& d) {
        const uint32_t first = rl(dl.x, 0), cs = rl(dl.z, 0);
#pragma unroll
        for (int q = 0; q < kCB / 1024; q++) {
            const uint32_t g = cs + 1024u * q + 16u * lane;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (g + 16 <= B.iend) v = ld16u(B.src + g);
            else if (g < B.iend) {
                for (uint32_t i = 0; i < B.iend - g; i++) {
                    const uint32_t by = uint32_t(B.src[g + i]) << (8 * (i & 3));
                    if (i < 4) v.x |= by; else if (i < 8) v.y |= by; else if (i < 12) v.z |= by; else v.w |= by;
                }
            }
            d.c[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 3; q++) { const uint32_t i = first + 64u * q + lane; d.t[q] = i < B.nseq ? B.tok[i] : 0; }
    };

    uint4 dcur = load_desc(0);
    uint4 dnext = load_desc(1);
    Data cur; load_data(dcur, cur);
    Data nxt = cur;
    for (uint32_t w = 0; w < B.nwin; w++) {
        const bool have_next = w + 1 < B.nwin;
        uint4 dnn = dnext;
        if (have_next) { load_data(dnext, nxt); dnn = load_desc(w + 2); }
        const uint32_t first = rl(dcur.x, 0), opos0 = rl(dcur.y, 0), cs = rl(dcur.z, 0), lit0 = rl(dcur.w, 0);
        const uint32_t ll0 = rl(dcur.x, 1), ml0 = rl(dcur.y, 1), off0 = rl(dcur.z, 1);
        const uint32_t last = min(rl(dcur.x, 2), B.nseq - 1);
        const uint32_t W0 = w << kWinLog, W1 = min(W0 + uint32_t(kWin), endp);
        const int lbw = int(w) + kAhead - kRW + 1;
        const uint32_t lowb = lbw > 0 ? uint32_t(lbw) << kWinLog : 0u;     // the ring is guaranteed from here on
        PT(1);
        // may this window be produced yet?  (not too far ahead of the chain wave; the ring slot's previous window flushed)
        for (uint32_t spins = 0; ldv(&sy->E_win) + kAhead < w || ldv(&sy->F_vis) + kRW <= w; ) if (spin_fail(sy, spins)) return;
        cbar();
        PT(0);
#pragma unroll
        for (int q = 0; q < kCB / 1024; q++) *reinterpret_cast<uint4*>(cbuf + 1024 * q + 16 * lane) = cur.c[q];
        uint32_t obase = opos0;
        for (uint32_t j0 = 0; first + j0 <= last; j0 += 64) {
            const uint32_t sidx = first + j0 + lane;
            const bool act = sidx <= last;
            const uint32_t tp = j0 == 0 ? cur.t[0] : j0 == 64 ? cur.t[1] : j0 == 128 ? cur.t[2] : (act ? B.tok[sidx] : 0u);
            // ---- decode (every rule was checked by the parser; the window's first sequence comes decoded)
            uint32_t ll = 0, ml = 0, off = 0, litpos = 0;
            if (act) {
                if (j0 == 0 && lane == 0) { ll = ll0; ml = ml0; off = off0; litpos = lit0; }
                else {
                    const uint32_t tk = sbyte(B, cbuf, cs, tp);
                    uint32_t p = tp + 1;
                    ll = tk >> 4;
                    if (ll == 15) { uint32_t bb; do { bb = sbyte(B, cbuf, cs, p); p++; ll += bb; } while (bb == 255); }
                    litpos = p;
                    if (sidx != B.nseq - 1) {
                        p += ll;
                        off = sbyte(B, cbuf, cs, p) | (sbyte(B, cbuf, cs, p + 1) << 8);
                        p += 2;
                        ml = tk & 15;
                        if (ml == 15) { uint32_t bb; do { bb = sbyte(B, cbuf, cs, p); p++; ml += bb; } while (bb == 255); }
                        ml += 4;
                    }
                }
            }
            const uint32_t len = ll + ml;
            const uint32_t incl = xscan_add(len);
            const uint32_t sp = obase + (incl - len) + B.a0;                 // shifted output position of the sequence
            obase += rl(incl, 63);
            PT(1);
            // ---- a free slot for this batch
            for (uint32_t spins = 0; produced >= ldv(&sy->consumed) + kNS; ) if (spin_fail(sy, spins)) return;
            cbar();
            PT(2);
            // ---- literals
            {
                const uint32_t ls = max(sp, W0), le = min(sp + ll, W1);
                const uint32_t n = (act && le > ls) ? le - ls : 0;
                const uint32_t cp = litpos + (ls - sp);
                const bool staged = cp - cs + n <= uint32_t(kCB);           // cp >= cs always
                const bool shortl = n && n <= uint32_t(kShort) && staged;
                copy_upto32_lds(lds_addr(ring) + (shortl ? (ls & kRM) : 0u), lds_addr(cbuf) + (shortl ? cp - cs : 0u), shortl ? n : 0u);
                unsigned long long lg = __ballot(n && !shortl);
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    const uint32_t d0 = rl(ls, l), nn = rl(n, l), c0 = rl(cp, l);
                    if (c0 - cs + nn <= uint32_t(kCB)) {
                        for (uint32_t k = lane; k < nn; k += 64) ring[(d0 + k) & kRM] = cbuf[c0 - cs + k];
                    } else wave_copy_hbm(ring, d0, B.src + c0, nn, lane);
                }
            }
            PT(3);
            // ---- matches: from HBM here, through the slot otherwise
            const uint32_t mstart = sp + ll;
            const uint32_t ds = max(mstart, W0), de = min(mstart + ml, W1);
            const uint32_t mn = (act && ml && de > ds) ? de - ds : 0;
            const uint32_t s0 = ds - off;
            const bool far = mn && s0 < lowb;                                // never overlapping: off > kWin >= mn
            const bool near = mn && !far;
            if (__ballot(far)) {
                uint32_t need = far ? ((s0 + mn - 1) >> kWinLog) + 1 : 0;
                for (int o = 32; o; o >>= 1) need = max(need, uint32_t(__shfl_xor(int(need), o)));
                for (uint32_t spins = 0; ldv(&sy->F_vis) < need; ) if (spin_fail(sy, spins)) return;
                cbar();
                const bool fs = far && mn <= uint32_t(kShort);
                copy_upto32_hbm(lds_addr(ring) + (fs ? (ds & kRM) : 0u), B.dst + (fs ? s0 - B.a0 : 0u), fs ? mn : 0u);
                unsigned long long lg = __ballot(far && !fs);
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    wave_copy_hbm(ring, rl(ds, l), B.dst + rl(s0, l) - B.a0, rl(mn, l), lane);
                }
            }
            PT(4);
            Slot* s = slots + (produced % kNS);
            const unsigned long long nb = __ballot(near);
            if (near) {
                const uint32_t idx = __builtin_amdgcn_mbcnt_hi(uint32_t(nb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(nb), 0));
                s->dst[idx] = ds; s->ol[idx] = off | (mn << 16);
            }
            uint32_t mx = near ? mn : 0u;
            for (int o = 32; o; o >>= 1) mx = max(mx, uint32_t(__shfl_xor(int(mx), o)));
            const bool lastb = first + j0 + 64 > last;
            if (lane == 0) { s->n = uint32_t(__builtin_popcountll(nb)); s->epos = lastb ? W1 : min(W1, obase + B.a0); s->last = lastb ? 1u : 0u; s->maxlen = mx; }
            lds_fence();                                  // every byte of the batch is in the LDS before the slot is published
            produced++;
            if (lane == 0) stv(&sy->ready, produced);
            PT(5);
        }
        dcur = dnext; dnext = dnn; cur = nxt;
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
    const float roff = __builtin_amdgcn_rcpf(float(max(off, 1u)));
    unsigned long long undone = __ballot(act);
    uint32_t rounds = 0;
    while (undone) {
        const int f = __builtin_ctzll(undone);
        const uint32_t Df = rl(dst, f);
        const bool mine = ((undone >> lane) & 1) != 0;
        const bool ready = mine && (hi <= Df || lane == f);     // nothing that is still missing lies below Df
        const unsigned long long rb = __ballot(ready);
        if (__builtin_popcountll(rb) <= 3 || maxlen > uint32_t(kShort)) {
            // few entries (the usual state of a dependency chain) or long ones: the whole wave copies one entry at a time
            unsigned long long q = rb;
            while (q) {
                const int l = __builtin_ctzll(q); q &= q - 1;
                const uint32_t d0 = rl(dst, l), oo = rl(off, l), nn = rl(len, l);
                if (nn <= 64) {
                    // byte k of the entry = source byte k mod off (all of [d0 - off, d0) is final): one step for any overlap
                    const float ro = __builtin_bit_cast(float, rl(__builtin_bit_cast(uint32_t, roff), l));
                    const uint32_t k = uint32_t(lane);
                    const uint32_t qd = uint32_t((float(k) + 0.5f) * ro);      // k / off, exact for k, off < 2^16 apart from k >= off * 2^.. (k < 64 here)
                    const uint32_t km = oo >= 64 ? k : k - qd * oo;
                    const uint32_t sa = ra + ((d0 - oo + km) & kRM), da = ra + ((d0 + k) & kRM);
                    uint32_t v;
                    asm volatile("s_nop 1\n\tds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(sa) : "memory");
                    if (k < nn) st_lo<0>(da, v);
                } else wave_copy_ring(ring, d0, oo, nn, lane);
            }
        } else {
            lane_copy_ring(ring, dst, off, ready ? len : 0u);
        }
        undone &= ~rb;
        rounds++;
    }
    return rounds;
}

__device__ __forceinline__ void chain_wave(const Blk& B, ring_t ring, Slot* slots, XSync* sy, int lane)
{
    uint32_t cons = 0;
    __builtin_amdgcn_s_setprio(3);
    PROF_DECL
    for (uint32_t w = 0; w < B.nwin; w++) {
        for (;;) {
            for (uint32_t spins = 0; ldv(&sy->ready) <= cons; ) if (spin_fail(sy, spins)) return;
            cbar();
            PT(0);
            const Slot* s = slots + (cons % kNS);
            const uint32_t n = s->n, last = s->last, maxlen = s->maxlen;
            if (n) { const uint32_t rounds = chain_slot(ring, s, n, maxlen, lane); PADD(3, rounds); PADD(4, n); }
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

__global__ __launch_bounds__(kXT)
void lz4_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                     const uint8_t* work)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing + kGuard];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[kCB + kGuard];
    __shared__ Slot slots[kNS];
    __shared__ XSync sy;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint8_t* slot = work + size_t(b) * kSlotBytes;
    const ParHdr* hdr = reinterpret_cast<const ParHdr*>(slot);
    if (hdr->status != kParsed) return;
    const fourmc_block blk = blocks[b];
    Blk B;
    B.src = src_base + blk.src_off; B.dst = dst_base + blk.dst_off;
    B.wdesc = reinterpret_cast<const uint4*>(slot + kWdescOff);
    B.tok = reinterpret_cast<const uint32_t*>(slot + kTokOff);
    B.dbg = reinterpret_cast<unsigned long long*>(const_cast<uint8_t*>(slot) + kDbgOff);
    B.iend = blk.src_len; B.nseq = hdr->nseq; B.total = hdr->total; B.nwin = hdr->nwin; B.a0 = hdr->a0;
    if (threadIdx.x == 0) { sy.E_win = 0; sy.F_vis = 0; sy.ready = 0; sy.consumed = 0; sy.abort = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (wave == 0) literal_wave(B, ring, cbuf, slots, &sy, lane);
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
