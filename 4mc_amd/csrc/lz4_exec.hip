// 4mc_amd/csrc/lz4_exec.hip - K1x: executes the sequence records of lz4_parse.hip; the LZ77 copy loop of the reference
// (native/lz4/lz4.c:2060-2110, :2300-2325) as a dataflow over a 32 KiB window of the output held in LDS.
//
// One workgroup per block, output produced in WINDOWS of 1 KiB (lz4par.h).  What a match copies is the only thing that
// depends on earlier output.  Instead of walking the sequences in order, every output byte of the ring has a DONE BIT
// (4 KiB bitmap in LDS); a match may run as soon as the bits of its source are set, whoever produced them:
//   * kNW worker waves take windows round robin.  A lane is one sequence (clipped to the window): it decodes its token
//     from the wave's private copy of the stream (prefetched a window ahead), copies its literals into the ring, sets
//     their bits, and then copies its match when the source is ready - from the ring while the source is younger than
//     the ring guarantees, from the block's flushed output in HBM otherwise (always ready).  Lanes whose source is not
//     ready yet are retried in rounds; only they ever wait, and only for the bytes they need.
//   * ONE flush wave writes windows whose bits are all set to HBM with aligned 16-byte stores, clears the bits of ring
//     slots nobody may read any more, and publishes F_vis, which also bounds how far workers run ahead.
// All irregular, byte-granular accesses stay in the LDS (byte-unaligned wider LDS accesses cost 64 clk per instruction
// on gfx950, byte accesses 2-4: profiles/r02_ubench_lds_unaligned.txt); loads are always issued in batches before the
// stores that depend on them.  HBM sees the coalesced stream reads, the far match gathers and the 16-byte flush stores.
// Ring reuse: window w overwrites window w - kRW, which has been flushed and which no reader may touch any more because
// readers never reach further back than kRW - kAhead windows through the ring.  Every wait is bounded; a wave that
// waits too long aborts the block to the exact kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"

using namespace lz4par;

namespace {

constexpr int kNW    = 6;                     // worker waves
constexpr int kRW    = 32;                    // windows in the ring
constexpr int kRing  = kRW * kWin;
constexpr uint32_t kRM = kRing - 1;
constexpr int kAhead = 14;                    // windows a worker may be ahead of the flushed output (2 * kAhead <= kRW)
constexpr int kLag   = 6;                     // flush stores in flight before the oldest one is waited for
constexpr int kCB    = 2048;                  // bytes of the stream a worker stages per window
constexpr int kGuard = 64;                    // readable bytes behind the ring / the staged stream (batched reads overshoot)
constexpr int kShort = 32;                    // pieces up to this length are copied by their own lane
constexpr uint32_t kSpinLimit = 1u << 21;
constexpr int kXT    = 64 * (kNW + 1);
constexpr uint32_t kBW = kRing / 32;          // words of the done bitmap

struct XSync {
    uint32_t F_vis;                           // windows flushed to HBM and visible
    uint32_t abort;
};

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

// ---------------------------------------------------------------------------------------------- done bits
// bit (pos & kRM) of bm = the ring byte of output position pos is final
__device__ __forceinline__ void set_bits(uint32_t* bm, uint32_t pos, uint32_t n)        // n <= 32, inside one window
{
    if (!n) return;
    const uint32_t p = pos & kRM, i = p >> 5, b = p & 31;
    const unsigned long long m = (n >= 32 ? 0xffffffffull : ((1ull << n) - 1)) << b;
    cbar();
    __hip_atomic_fetch_or(&bm[i], uint32_t(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (m >> 32) __hip_atomic_fetch_or(&bm[i + 1], uint32_t(m >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool bits_set(const uint32_t* bm, uint32_t pos, uint32_t n)  // n <= 32, any position
{
    const uint32_t p = pos & kRM, i = p >> 5, b = p & 31;
    const unsigned long long m = (n >= 32 ? 0xffffffffull : ((1ull << n) - 1)) << b;
    const unsigned long long v = uint64_t(ldv(&bm[i])) | (uint64_t(ldv(&bm[(i + 1) & (kBW - 1)])) << 32);
    return (v & m) == m;
}
// bits of [lo, hi) inside the 32-bit word that starts at bit position word_lo (plain, unwrapped positions)
__device__ __forceinline__ uint32_t range_mask(uint32_t word_lo, uint32_t lo, uint32_t hi)
{
    const uint32_t a = max(lo, word_lo), e = min(hi, word_lo + 32);
    if (e <= a) return 0u;
    const uint32_t cnt = e - a;
    return (cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1)) << (a - word_lo);
}
// whole wave, wave-uniform range inside one window
__device__ __forceinline__ void wave_set_bits(uint32_t* bm, uint32_t pos, uint32_t n, int lane)
{
    const uint32_t p = pos & kRM, w0 = p & ~uint32_t(kWin - 1);
    if (lane < 32) {
        const uint32_t m = range_mask(w0 + 32 * lane, p, p + n);
        cbar();
        if (m) __hip_atomic_fetch_or(&bm[(w0 >> 5) + lane], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// whole wave, wave-uniform range anywhere, n <= kWin: at most 33 words, lane j looks at word first + j
__device__ __forceinline__ bool wave_bits_set(const uint32_t* bm, uint32_t pos, uint32_t n, int lane)
{
    const uint32_t p = pos & kRM, first = p >> 5;
    bool ok = true;
    if (lane < 34) {
        const uint32_t wi = first + lane;                                    // unwrapped word index
        const uint32_t m = range_mask(32 * wi, p, p + n);
        if (m) ok = (ldv(&bm[wi & (kBW - 1)]) & m) == m;
    }
    return __ballot(!ok) == 0;
}

struct Blk {
    const uint8_t* src; uint8_t* dst; const uint4* wdesc; const uint32_t* tok; unsigned long long* dbg;
    uint32_t iend, nseq, total, nwin, a0;
};

// stream byte at position p: from the wave's staged copy [cs, cs + kCB) or from HBM
typedef const __attribute__((address_space(3))) uint8_t* lds_bytes;   // keeps the two sides of the choice below apart (no flat loads)
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

// ------------------------------------------------------------------------------------------------ worker wave
__device__ __forceinline__ void worker_wave(const Blk& B, ring_t ring, uint32_t* bm, uint8_t* cbuf, XSync* sy, int ww, int lane)
{
    const uint32_t endp = B.total + B.a0;
    PROF_DECL

    // prefetch registers: descriptors two windows ahead, stream bytes and token positions one window ahead
    auto load_desc = [&](uint32_t w) -> uint4 {
        const uint32_t i = min(w + uint32_t(lane & 1), B.nwin);
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
                uint32_t wv[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < B.iend - g; i++) wv[i >> 2] |= uint32_t(B.src[g + i]) << (8 * (i & 3));
                v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            }
            d.c[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 3; q++) { const uint32_t i = first + 64u * q + lane; d.t[q] = i < B.nseq ? B.tok[i] : 0; }
    };

    uint32_t w = ww;
    if (w >= B.nwin) { PROF_OUT(B, ww, lane); return; }
    uint4 dcur = load_desc(w);
    uint4 dnext = load_desc(w + kNW);
    Data cur; load_data(dcur, cur);
    Data nxt = cur;
    for (; w < B.nwin; w += kNW) {
        const bool have_next = w + kNW < B.nwin;
        uint4 dnn = dnext;
        if (have_next) { load_data(dnext, nxt); dnn = load_desc(w + 2 * kNW); }
        const uint32_t first = rl(dcur.x, 0), opos0 = rl(dcur.y, 0), cs = rl(dcur.z, 0);
        const uint32_t last = min(rl(dcur.x, 1), B.nseq - 1);
        const uint32_t W0 = w << kWinLog, W1 = min(W0 + uint32_t(kWin), endp);
        const int lbw = int(w) + kAhead - kRW;
        const uint32_t lowb = lbw > 0 ? uint32_t(lbw) << kWinLog : 0u;     // the ring is guaranteed from here on
        PT(1);
        // may this window be produced yet?
        for (uint32_t spins = 0; ldv(&sy->F_vis) + kAhead <= w; ) if (spin_fail(sy, spins)) return;
        cbar();
        PT(0);
        // stage the stream
#pragma unroll
        for (int q = 0; q < kCB / 1024; q++) *reinterpret_cast<uint4*>(cbuf + 1024 * q + 16 * lane) = cur.c[q];
        uint32_t obase = opos0;
        for (uint32_t j0 = 0; first + j0 <= last; j0 += 64) {
            const uint32_t sidx = first + j0 + lane;
            const bool act = sidx <= last;
            uint32_t tp = j0 == 0 ? cur.t[0] : j0 == 64 ? cur.t[1] : j0 == 128 ? cur.t[2] : (act ? B.tok[sidx] : 0u);
            // ---- decode (every rule was checked by the parser)
            uint32_t ll = 0, ml = 0, off = 0, litpos = 0;
            if (act) {
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
            const uint32_t len = ll + ml;
            const uint32_t incl = xscan_add(len);
            const uint32_t sp = obase + (incl - len) + B.a0;                 // shifted output position of the sequence
            obase += rl(incl, 63);
            PT(1);
            // ---- literals
            {
                const uint32_t ls = max(sp, W0), le = min(sp + ll, W1);
                const uint32_t n = (act && le > ls) ? le - ls : 0;
                const uint32_t cp = litpos + (ls - sp);
                const bool staged = cp - cs + n <= uint32_t(kCB);           // cp >= cs always
                const bool shortl = n && n <= uint32_t(kShort) && staged;
                copy_upto32_lds(lds_addr(ring) + (shortl ? (ls & kRM) : 0u), lds_addr(cbuf) + (shortl ? cp - cs : 0u), shortl ? n : 0u);
                if (shortl) set_bits(bm, ls, n);
                unsigned long long lg = __ballot(n && !shortl);
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    const uint32_t d0 = rl(ls, l), nn = rl(n, l), c0 = rl(cp, l);
                    if (c0 - cs + nn <= uint32_t(kCB)) {
                        for (uint32_t k = lane; k < nn; k += 64) ring[(d0 + k) & kRM] = cbuf[c0 - cs + k];
                    } else wave_copy_hbm(ring, d0, B.src + c0, nn, lane);
                    wave_set_bits(bm, d0, nn, lane);
                }
            }
            PT(3);
            // ---- match
            const uint32_t mstart = sp + ll;
            const uint32_t ds = max(mstart, W0), de = min(mstart + ml, W1);
            const uint32_t mn = (act && ml && de > ds) ? de - ds : 0;
            const uint32_t s0 = ds - off;
            const uint32_t need = min(mn, off);                              // source bytes somebody else produces
            const bool in_ring = s0 >= lowb;
            const bool shortm = mn <= uint32_t(kShort);
            unsigned long long pending = __ballot(mn != 0);
            uint32_t idle = 0, spins = 0;
            while (pending) {
                const bool mine = ((pending >> lane) & 1) != 0;
                // short pieces: every lane for itself
                const bool ready = mine && shortm && (!in_ring || bits_set(bm, s0, need));
                cbar();
                const bool r1 = ready && in_ring, r2 = ready && !in_ring;
                if (__ballot(r1)) lane_copy_ring(ring, ds, off, r1 ? mn : 0u);
                if (__ballot(r2))        // far source: flushed long ago (kAhead bounds it), never overlapping
                    copy_upto32_hbm(lds_addr(ring) + (r2 ? (ds & kRM) : 0u), B.dst + (r2 ? s0 - B.a0 : 0u), r2 ? mn : 0u);
                if (ready) set_bits(bm, ds, mn);
                unsigned long long fin = __ballot(ready);
                // long pieces: the whole wave, one at a time
                unsigned long long lg = __ballot(mine && !shortm);
                while (lg) {
                    const int l = __builtin_ctzll(lg); lg &= lg - 1;
                    const uint32_t d0 = rl(ds, l), nn = rl(mn, l), ss = rl(s0, l), oo = rl(off, l), nd = rl(need, l);
                    if (ss >= lowb) { if (!wave_bits_set(bm, ss, nd, lane)) continue; cbar(); wave_copy_ring(ring, d0, oo, nn, lane); }
                    else wave_copy_hbm(ring, d0, B.dst + ss - B.a0, nn, lane);
                    wave_set_bits(bm, d0, nn, lane);
                    fin |= 1ull << l;
                }
                pending &= ~fin;
                PADD(6, 1);
                if (fin) PT(4); else PT(5);
                if (pending) {
                    PADD(7, fin ? 0 : 1);
                    if (fin) idle = 0;
                    else {
                        if (++idle > 2) __builtin_amdgcn_s_sleep(1);
                        if (++spins > kSpinLimit) { stv(&sy->abort, 1); return; }
                        if (ldv(&sy->abort)) return;
                    }
                }
            }
            PT(4);
        }
        dcur = dnext; dnext = dnn; cur = nxt;
    }
    PROF_OUT(B, ww, lane);
}

// ------------------------------------------------------------------------------------------------ flush wave
__device__ __forceinline__ void flush_wave(const Blk& B, ring_t ring, uint32_t* bm, XSync* sy, int lane)
{
    const uint32_t endp = B.total + B.a0;
    uint32_t cleared = 0;                       // windows whose bits have been cleared for the slot's next user
    PROF_DECL
    // publishing F_vis = x lets windows < x + kAhead start, i.e. reuse the slots of windows < x + kAhead - kRW; nobody reads
    // those any more (their readers sit in windows < x, all flushed)
    auto publish = [&](uint32_t x) {
        for (; cleared + uint32_t(kRW - kAhead) < x; cleared++)
            if (lane < 32) stv(&bm[(((cleared << kWinLog) & kRM) >> 5) + lane], 0u);
        lds_fence();
        if (lane == 0) stv(&sy->F_vis, x);
    };
    for (uint32_t f = 0; f < B.nwin; f++) {
        PT(1);
        // all bits of the window set?
        const uint32_t wpos = f << kWinLog;
        const uint32_t exp = lane < 32 ? range_mask(wpos + 32 * lane, max(wpos, B.a0), min(wpos + uint32_t(kWin), endp)) : 0u;
        for (uint32_t spins = 0;;) {
            const uint32_t v = lane < 32 ? ldv(&bm[((wpos & kRM) >> 5) + lane]) : 0u;
            if (__ballot((v & exp) != exp) == 0) break;
            if (spin_fail(sy, spins)) return;
        }
        cbar();
        PT(0);
        const uint32_t p0 = wpos + 16u * lane;
        const uint4 v = *reinterpret_cast<const uint4*>(const_cast<const uint8_t*>(ring) + (p0 & kRM));   // behind the barrier above
        uint8_t* g = B.dst + p0 - B.a0;        // 16-byte aligned by construction of a0
        if (p0 >= B.a0 && p0 + 16 <= endp) *reinterpret_cast<uint4*>(g) = v;
        else {
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t k = 0; k < 16; k++) if (p0 + k >= B.a0 && p0 + k < endp) g[k] = uint8_t(wv[k >> 2] >> (8 * (k & 3)));
        }
        if (f == 0 || f + 1 == B.nwin) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish(f + 1); }
        else { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); if (f >= uint32_t(kLag)) publish(f + 1 - kLag); }
    }
    PT(1);
    PROF_OUT(B, kNW, lane);
}

} // namespace

__global__ __launch_bounds__(kXT)
void lz4_exec_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks, uint32_t nblocks,
                     const uint8_t* work)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing + kGuard];
    __shared__ __attribute__((aligned(16))) uint32_t bm[kBW];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[kNW][kCB + kGuard];
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
    for (int i = threadIdx.x; i < int(kBW); i += kXT) bm[i] = 0;
    if (threadIdx.x == 0) { sy.F_vis = 0; sy.abort = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (wave < kNW) worker_wave(B, ring, bm, cbuf[wave], &sy, wave, lane);
    else {
        flush_wave(B, ring, bm, &sy, lane);
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
