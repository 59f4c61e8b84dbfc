// 4mc_amd/csrc/lz4hc_encode.hip — K3: batched LZ4 HC (hash-chain) block encode on gfx950,
// BYTE-IDENTICAL to LZ4_compress_HC of the reference at the levels 4mc reaches
// (4mc High = level 4, 4mc Ultra = level 8).
//
// Replaces native/4mc.c:301 with LZ4_compress_HC (:249-252) and native/jniCompressor.c:157
// (LZ4_compressHC2) -> native/lz4/lz4hc.c:958-973 -> :800-861 -> LZ4HC_compress_hashChain :553-788,
// search LZ4HC_InsertAndGetWiderMatch :239-447 (patternAnalysis off for <= 128 attempts :565,
// chainSwap off :461,:603,:648), tables LZ4HC_Insert :120-141.
//
// TWO wavefronts own one block.  LZ4HC_Insert puts every position into the tables, in order, whatever the
// parse does, so tables and candidate chains are a function of the input alone:
//   * the BUILDER wave inserts positions (64 per step, one read round trip per step: positions that
//     share a hash chain to each other by lane distance) and builds a look-ahead window for every
//     aligned group of 64 positions: the chains of the 64 positions walked at once, one per lane,
//     every candidate measured on the way (4-byte check, 32 bytes forward, 16 backward, all read in one
//     round trip per chain step), and the answer of the search that opens a sequence there;
//   * the PARSER wave runs the lazy three-match arbitration (lz4hc.c:592-732, mirrored statement by
//     statement) out of those windows in LDS - a ring of 3, the builder stays at most 2 ahead - and
//     emits the sequences.  Its searches only apply the limits that depend on the parse (how far it
//     may look back, the running `longest`); matches longer than the cached 32 / 16 bytes, and the
//     attempts beyond 8 of levels 5-8, touch memory again.
// The 384 KiB of match-finder state (32768 x u32 hash heads + 131072 x u16 chain deltas, twice the
// reference's chain so that inserting ahead of the parser overwrites nothing it needs) live in a
// per-block HBM workspace slot.  The reference's running "if (ml > longest)" over candidates in chain
// order equals max-length with earliest-wins ties, which is what the reductions compute.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devenc.h"

namespace {

constexpr int      kHashLog = 15;
constexpr uint32_t kIdx0 = 65536;            // table index of position 0 (LZ4HC_init_internal)
constexpr uint32_t kMaxDist = 65535;
constexpr int      kMinMatch = 4, kMfLimit = 12, kLastLit = 5, kOptimalML = 18;
constexpr int      kScore = 1024;
constexpr uint32_t kChainMask = 0x1FFFF;     // chain deltas of the last 128 Ki positions (the reference keeps 64 Ki): room for
                                             // inserting up to 64 Ki positions AHEAD of the search position
constexpr int      kWinK = 8;                // candidates cached per position of the look-ahead window
constexpr size_t   kWorkBytes = (size_t(4) << kHashLog) + 2 * (size_t(kChainMask) + 1);     // per block: heads + chain

// maximum over the wavefront on the DPP network (row shifts, then row broadcasts: the total lands in lane 63)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t wave_max(uint32_t v)            // values >= 0, identity 0
{
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v)); v = max(v, dpp0<0x143, 0xc>(v));
    return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}

__device__ __forceinline__ uint32_t hc_hash(uint32_t v) { return (v * 2654435761u) >> (32 - kHashLog); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// look-ahead window (LDS): for 64 consecutive positions, the first kWinK chain candidates with their measured lengths
struct HCWin {
    uint32_t cand[64 * kWinK];   // table index of the candidate
    uint8_t  fl[64 * kWinK];     // equal bytes after the first 4 (0..32); 0xFF: the first 4 bytes differ
    uint8_t  bl[64 * kWinK];     // equal bytes before (0..16)
    uint32_t next[64];           // chain continuation after the cached candidates (0: chain ended)
    uint8_t  nc[64];             // cached candidates of the position
    uint8_t  live[64];           // 0: no candidate of the position can give a match (none shares its first 4 bytes, chain ended)
    uint8_t  fm[64];             // the search that opens a sequence at this position (nothing to look back at, longest = 3),
    uint32_t fr[64];             // answered ahead: match length (0: none, 0xFF: ask hc_wider) and the candidate's table index
};

// Two wavefronts own one block.  The table state and the look-ahead windows depend on the input alone, so a BUILDER
// wave inserts positions and builds the windows of the aligned 64-position groups in order, a few ahead, while the
// PARSER wave runs the serial lazy parse out of them.  A ring of kRing windows in LDS; `built` = windows finished,
// `keep` = oldest window the parser may still open (it can step back one window at most without rebuilding).
constexpr int kRing = 3;
struct HCSync { uint32_t built, keep, done; };
__device__ __forceinline__ uint32_t ld_acq(uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_rel(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
constexpr uint32_t kSpinLimit = 1u << 24;    // a wait that long means the other wave is gone: give up instead of hanging

struct HC {
    const uint8_t* src;
    uint32_t* heads;       // [32768]
    uint16_t* chain;       // [131072]
    uint32_t* score;       // LDS [kScore], all 0xFFFFFFFF between uses
    HCWin*    win;         // LDS: the window in use (parser) / being built (builder)
    HCWin*    wins;        // LDS: kRing shared windows + one private to the parser
    HCSync*   sync;        // LDS
    uint32_t  wmax;        // parser: highest window opened so far
    bool      failed;
    uint32_t  ntu;         // next position to insert (runs AHEAD of the search position)
    uint32_t  wbase;       // first position of the window, 0xFFFFFFFF: none
    uint32_t  n, matchlimit, mflimit;
    int       lane;
    uint32_t  pl_byte, pl_dst, pl_n;   // literal run of the last sequence: read, not yet written (see hc_emit)
#ifdef K2_PROF   // side build (make prof): cycles in window build / searches / emission, call counts
    uint64_t  pt_build, pt_search, pt_emit, pt0; uint32_t n_build, n_search, n_emit, n_mem;
    uint64_t  px_emit, px_skip, px_first; uint32_t nx_emit, nx_skip;
#endif
};
#ifdef K2_PROF
#define K3PH(c, acc) do { const uint64_t t_ = __builtin_readcyclecounter(); (c).acc += t_ - (c).pt0; (c).pt0 = t_; } while (0)
#define K3CNT(c, f) do { (c).f++; } while (0)
#define K3X0() const uint64_t x0_ = __builtin_readcyclecounter()
#define K3X(c, acc) do { (c).acc += __builtin_readcyclecounter() - x0_; } while (0)
#else
#define K3PH(c, acc) do { } while (0)
#define K3CNT(c, f) do { } while (0)
#define K3X0() do { } while (0)
#define K3X(c, acc) do { } while (0)
#endif

// LZ4HC_Insert (lz4hc.c:120-141): positions [ntu, upto) enter the tables in order, 64 per step.  Inside one step the
// serial order only matters between positions with the same hash: each of those chains to its nearest predecessor in
// the step (a lane distance), only the first of a hash needs the old head from memory and only the last one becomes
// the new head - so a step is one read round trip however many lanes share a hash (byte runs put dozens into one).
// Lanes that share are found with the folded LDS scoreboard, their exact predecessor with one ballot per distinct hash.
__device__ __forceinline__ void hc_insert(HC& c, uint32_t upto)
{
    while (c.ntu < upto) {
        const uint32_t pos = c.ntu + c.lane;
        const bool on = pos < upto;
        const uint32_t h = on ? hc_hash(ld4(c.src + pos)) : 0xFFFFFFFFu;
        const uint32_t idx = pos + kIdx0;
        int prev = -1; bool last = on;
        if (on) {
            uint32_t* sc = &c.score[h & (kScore - 1)];
            atomicMin(sc, uint32_t(c.lane));
        }
        const bool lost = on && c.score[h & (kScore - 1)] != uint32_t(c.lane);      // an earlier lane sits in my bucket
        if (on) c.score[h & (kScore - 1)] = 0xFFFFFFFFu;
        for (unsigned long long rem = __ballot(lost); rem; ) {
            const uint32_t hv = uint32_t(__builtin_amdgcn_readlane(int(h), __builtin_ctzll(rem)));
            const unsigned long long m = __ballot(h == hv);
            if (h == hv) {
                const unsigned long long lower = m & ~(~0ull << c.lane), upper = m & (~1ull << c.lane);
                if (lower) prev = 63 - __builtin_clzll(lower);
                last = upper == 0;
            }
            rem &= ~m;
        }
        if (on) {
            uint32_t delta;
            if (prev >= 0) delta = uint32_t(c.lane - prev);
            else { delta = idx - c.heads[h]; if (delta > kMaxDist) delta = kMaxDist; }
            c.chain[idx & kChainMask] = uint16_t(delta);
            if (last) c.heads[h] = idx;
        }
        c.ntu = min(upto, c.ntu + 64u);
    }
}

// number of equal bytes at a / b (a > b), a stops at lim; the whole wavefront works on one pair
__device__ __forceinline__ uint32_t wave_count_fwd(const uint8_t* s, uint32_t a, uint32_t b, uint32_t lim, int lane)
{
    uint32_t n = 0;
    for (;;) {
        if (a + 1024 <= lim) {
            const U16B x = *reinterpret_cast<const U16B*>(s + a + 16 * lane);
            const U16B y = *reinterpret_cast<const U16B*>(s + b + 16 * lane);
            const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
            const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
            const unsigned long long bad = __ballot(eq < 16);
            if (bad) { const int l = __builtin_ctzll(bad); return n + 16 * l + __builtin_amdgcn_readlane(eq, l); }
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

// number of equal bytes going backwards from a / b (exclusive), at most `maxn`
__device__ __forceinline__ uint32_t wave_count_back(const uint8_t* s, uint32_t a, uint32_t b, uint32_t maxn, int lane)
{
    uint32_t n = 0;
    for (;;) {
        const uint32_t j = n + lane + 1;
        const bool same = (j <= maxn) && s[a - j] == s[b - j];
        const unsigned long long bad = ~__ballot(same);
        if (bad) return n + __builtin_ctzll(bad);
        n += 64;
    }
}

__device__ __forceinline__ uint32_t eq16_fwd(const U16B& x, const U16B& y)
{
    const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
    return d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
}
__device__ __forceinline__ uint32_t eq16_back(const U16B& x, const U16B& y)      // both end just before the compared positions
{
    const uint64_t d1 = x.b ^ y.b, d0 = x.a ^ y.a;
    return d1 ? uint32_t(__builtin_clzll(d1) >> 3) : (d0 ? 8u + uint32_t(__builtin_clzll(d0) >> 3) : 16u);
}

// Look-ahead window.  Every position enters the tables (LZ4HC_Insert inserts them all, in order, whatever the parse
// does), so the chain of position q is a function of the input alone: its first candidate is q - chain[q], the delta
// recorded when q itself was inserted.  That lets the wave walk the chains of 64 consecutive positions AT ONCE (one
// per lane, kWinK dependent steps for all of them together) and measure every candidate on the way, instead of one
// dependent chain step per memory round trip.  The serial parser below then reads its searches out of LDS.
__device__ __forceinline__ void hc_build_window(HC& c, uint32_t wb, int attempts, bool insert)
{
    const uint8_t* s = c.src;
    const int lane = c.lane;
    if (insert) hc_insert(c, min(wb + 64u, c.n - 3u));
    const uint32_t q = wb + lane;
    const bool act = q <= c.mflimit;                                  // no search starts beyond mflimit
    const uint32_t qi = q + kIdx0, lowest = (kIdx0 + 65536 > qi) ? kIdx0 : qi - kMaxDist;
    const uint32_t pat = ld4(s + (act ? q : 0u));
    const bool wide_f = act && q + 36 <= c.n, wide_b = act && q >= 16;
    U16B f0 = {0, 0}, f1 = {0, 0}, b0 = {0, 0};
    if (wide_f) { f0 = *reinterpret_cast<const U16B*>(s + q + 4); f1 = *reinterpret_cast<const U16B*>(s + q + 20); }
    if (wide_b) b0 = *reinterpret_cast<const U16B*>(s + q - 16);
    const uint32_t fcap = act ? min(32u, c.matchlimit - (q + 4)) : 0u;
    uint32_t mi = 0; bool go = false;
    if (act) {
        const uint32_t d = insert ? uint32_t(c.chain[qi & kChainMask]) : uint32_t(__hip_atomic_load(&c.chain[qi & kChainMask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        mi = qi - d;
        go = mi >= lowest;
        // a delta of 65535 is either that distance or "further" (capped at insertion): it is the former iff the
        // position there carries our hash
        if (go && d == kMaxDist) go = hc_hash(ld4(s + (mi - kIdx0))) == hc_hash(pat);
    }
    uint32_t cnt = 0; bool any = false;
    uint32_t best = 0, best_mi = 0; bool ask = attempts > kWinK;      // running "ml > longest" in chain order: the earliest of the longest
    const int K = attempts < kWinK ? attempts : kWinK;
    for (int k = 0; k < K; k++) {
        if (!__ballot(go)) break;
        if (go) {
            const uint32_t m = mi - kIdx0;
            const uint32_t d = insert ? uint32_t(c.chain[mi & kChainMask]) : uint32_t(__hip_atomic_load(&c.chain[mi & kChainMask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            // everything this candidate can be asked for is read at once, whether or not its first four bytes match:
            // one round trip per chain step instead of up to three dependent ones
            const bool wf = wide_f, wb2 = wide_b && m >= 16;
            const uint32_t c4 = ld4(s + m);
            U16B g0 = {0, 0}, g1 = {0, 0}, h0 = {0, 0};
            if (wf) { g0 = *reinterpret_cast<const U16B*>(s + m + 4); g1 = *reinterpret_cast<const U16B*>(s + m + 20); }
            if (wb2) h0 = *reinterpret_cast<const U16B*>(s + m - 16);
            uint32_t fl = 0xFF, bl = 0;
            if (c4 == pat) {
                if (wf) {
                    uint32_t e = eq16_fwd(f0, g0);
                    if (e == 16) e += eq16_fwd(f1, g1);
                    fl = min(e, fcap);
                } else { fl = 0; while (fl < fcap && s[q + 4 + fl] == s[m + 4 + fl]) fl++; }
                if (wb2) bl = eq16_back(b0, h0);
                else { const uint32_t bc = min(16u, min(q, m)); while (bl < bc && s[q - 1 - bl] == s[m - 1 - bl]) bl++; }
            }
            c.win->cand[lane * kWinK + k] = mi; c.win->fl[lane * kWinK + k] = uint8_t(fl); c.win->bl[lane * kWinK + k] = uint8_t(bl);
            cnt++; any |= fl != 0xFF;
            if (fl != 0xFF) {
                if (fl == 32 && q + kMinMatch + 32 < c.matchlimit) ask = true;      // longer than what was read
                if (kMinMatch + fl > best) { best = kMinMatch + fl; best_mi = mi; }
            }
            mi -= d;
            go = mi >= lowest;
        }
    }
    c.win->nc[lane] = uint8_t(cnt);
    c.win->fm[lane] = uint8_t(ask ? 0xFF : best); c.win->fr[lane] = best_mi;
    c.win->next[lane] = go ? mi : 0u;
    c.win->live[lane] = uint8_t((any ? 1 : 0) | ((go && attempts > K) ? 2 : 0));
    c.wbase = wb;
}

// parser: make the window of position ip current
__device__ __forceinline__ void hc_acquire(HC& c, uint32_t ip, int attempts)
{
    const uint32_t w = ip >> 6;
    if (w > c.wmax) { c.wmax = w; if (c.lane == 0) st_rel(&c.sync->keep, w - 1); }
    for (uint32_t spins = 0; ld_acq(&c.sync->built) <= w; ) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) { c.failed = true; break; }
    }
    if (w + 1 >= c.wmax) c.win = c.wins + (w % kRing);       // still in the ring (the builder never overwrites >= keep)
    else { c.win = c.wins + kRing; hc_build_window(c, w << 6, attempts, false); }   // stepped further back: rebuild it privately (tables are already ahead)
    c.wbase = w << 6;
}

// LZ4HC_InsertAndGetWiderMatch (lz4hc.c:239-447), no dictionary, no pattern analysis, no chain swap.
__device__ __forceinline__ int hc_wider(HC& c, uint32_t ip, uint32_t low, uint32_t high, int longest,
                                        uint32_t& mpos, uint32_t& spos, int attempts)
{
    const uint8_t* s = c.src;
    const int lane = c.lane;
    K3PH(c, pt_emit);
    if (c.wbase == 0xFFFFFFFFu || ip - c.wbase >= 64u) { hc_acquire(c, ip, attempts); K3CNT(c, n_build); K3PH(c, pt_build); }
    K3CNT(c, n_search);
    const uint32_t j = ip - c.wbase;
    const uint32_t ip_idx = ip + kIdx0;
    const uint32_t lowest = (kIdx0 + 65536 > ip_idx) ? kIdx0 : ip_idx - kMaxDist;
    const uint32_t lookback = ip - low;
    const uint32_t pattern = ld4(s + ip);
    int nc = int(c.win->nc[j]);
    uint32_t mi = c.win->next[j];
    if (nc == 0) { K3PH(c, pt_search); return longest; }               // empty chain (the common case in incompressible data)
    attempts -= nc;
    bool from_window = true;
    for (;;) {
        uint32_t m = 0, fl = 0, bk = 0; bool live = false, more_f = false, more_b = false;
        if (from_window) {
            // ---- the cached candidates: measured when the window was built; only limits that depend on the parse remain
            if (lane < nc) {
                const uint32_t f = c.win->fl[j * kWinK + lane], b = c.win->bl[j * kWinK + lane];
                m = c.win->cand[j * kWinK + lane] - kIdx0;
                live = f != 0xFF;
                if (live) {
                    fl = f; more_f = (f == 32) && (ip + kMinMatch + 32 < high);
                    const uint32_t maxb = min(lookback, m);
                    bk = min(b, maxb); more_b = (b == 16) && (maxb > 16);
                }
            }
            for (unsigned long long todo = __ballot(more_f); todo; todo &= todo - 1) {
                const int l = __builtin_ctzll(todo);
                const uint32_t mm = uint32_t(__builtin_amdgcn_readlane(int(m), l));
                const uint32_t extra = wave_count_fwd(s, ip + kMinMatch + 32, mm + kMinMatch + 32, high, lane);
                K3CNT(c, n_mem);
                if (lane == l) fl += extra;
            }
            for (unsigned long long todo = __ballot(more_b); todo; todo &= todo - 1) {
                const int l = __builtin_ctzll(todo);
                const uint32_t mm = uint32_t(__builtin_amdgcn_readlane(int(m), l));
                const uint32_t extra = wave_count_back(s, ip - 16, mm - 16, min(lookback, mm) - 16, lane);
                if (lane == l) bk += extra;
            }
        } else {
            // ---- levels with more attempts than the window caches: the rest of the chain, 64 candidates per step
            uint32_t cand = 0; nc = 0;
            while (mi >= lowest && attempts > 0 && nc < 64) {
                if (lane == nc) cand = mi;
                nc++; attempts--;
                mi -= uint32_t(uni(int(__hip_atomic_load(&c.chain[mi & kChainMask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))));   // (the builder wave is writing this table: not through a stale L1 line)
            }
            m = cand - kIdx0;
            live = lane < nc && ld4(s + (lane < nc ? m : 0u)) == pattern;
            if (live) {                                           // forward: up to 32 bytes here
                uint32_t a = ip + kMinMatch, b = m + kMinMatch;
                more_f = true;
                for (int it = 0; it < 4; it++) {
                    if (a + 8 > high) { while (a < high && s[a] == s[b]) { a++; b++; fl++; } more_f = false; break; }
                    const uint64_t x = ld8(s + a) ^ ld8(s + b);
                    if (x) { fl += uint32_t(__builtin_ctzll(x) >> 3); more_f = false; break; }
                    a += 8; b += 8; fl += 8;
                }
            }
            if (live && lookback) {                               // backward: up to 32 bytes here
                const uint32_t maxb = min(lookback, m);
                more_b = true;
                for (int it = 0; it < 4; it++) {
                    if (bk + 8 > maxb) { while (bk < maxb && s[ip - bk - 1] == s[m - bk - 1]) bk++; more_b = false; break; }
                    const uint64_t x = ld8(s + ip - bk - 8) ^ ld8(s + m - bk - 8);
                    if (x) { bk += uint32_t(__builtin_clzll(x) >> 3); more_b = false; break; }
                    bk += 8;
                }
            }
            for (unsigned long long todo = __ballot(more_f); todo; todo &= todo - 1) {
                const int l = __builtin_ctzll(todo);
                const uint32_t mm = uint32_t(__builtin_amdgcn_readlane(int(m), l));
                const uint32_t extra = wave_count_fwd(s, ip + kMinMatch + 32, mm + kMinMatch + 32, high, lane);
                if (lane == l) fl += extra;
            }
            for (unsigned long long todo = __ballot(more_b); todo; todo &= todo - 1) {
                const int l = __builtin_ctzll(todo);
                const uint32_t mm = uint32_t(__builtin_amdgcn_readlane(int(m), l));
                const uint32_t maxb = min(lookback, mm);
                const uint32_t extra = wave_count_back(s, ip - 32, mm - 32, maxb - 32, lane);
                if (lane == l) bk += extra;
            }
        }
        // ---- running "ml > longest" in chain order == max length, earliest candidate wins ties
        const uint32_t ml = live ? kMinMatch + fl + bk : 0u;
        uint32_t key = (live && int(ml) > longest) ? ((ml << 6) | uint32_t(63 - lane)) : 0u;
        key = wave_max(key);
        if (key) {
            const int l = 63 - int(key & 63);
            longest = int(key >> 6);
            const uint32_t bm = uint32_t(__builtin_amdgcn_readlane(int(m), l)), bb = uint32_t(__builtin_amdgcn_readlane(int(bk), l));
            mpos = bm - bb; spos = ip - bb;
        }
        from_window = false;
        if (!(mi != 0 && mi >= lowest && attempts > 0)) break;
    }
    K3PH(c, pt_search);
    return longest;
}

// LZ4HC_encodeSequence (lz4hc.c:467-548): returns true when `limited` and the output would overflow
__device__ __forceinline__ bool hc_emit(HC& c, const uint8_t* src, uint8_t* dst, uint32_t& ip, uint32_t& op, uint32_t& anchor,
                                        int ml, uint32_t match, bool limited, uint32_t cap, int lane)
{
    K3X0();
    const uint32_t lit = ip - anchor;
    const uint32_t token_pos = op++;
    if (limited && op + lit / 255 + lit + (2 + 1 + kLastLit) > cap) return true;
    uint32_t tok;
    if (lit >= 15) { tok = 0xF0; op += emit_len(dst + op, lit - 15, lane); }
    else tok = lit << 4;
    // a short literal run is read now and written at the NEXT emit: the parser never waits for the read to come back
    if (lit <= 64) {
        const uint32_t v = uint32_t(lane) < lit ? uint32_t(src[anchor + lane]) : 0u;
        if (uint32_t(lane) < c.pl_n) dst[c.pl_dst + lane] = uint8_t(c.pl_byte);
        c.pl_byte = v; c.pl_dst = op; c.pl_n = lit;
    } else {
        if (uint32_t(lane) < c.pl_n) dst[c.pl_dst + lane] = uint8_t(c.pl_byte);
        c.pl_n = 0;
        copy_bytes(dst + op, src + anchor, lit, lane);
    }
    op += lit;
    const uint32_t off = ip - match;
    if (lane == 0) { dst[op] = uint8_t(off); dst[op + 1] = uint8_t(off >> 8); }
    op += 2;
    const uint32_t mcode = uint32_t(ml) - kMinMatch;
    if (limited && op + mcode / 255 + (1 + kLastLit) > cap) return true;
    if (mcode >= 15) { tok += 15; op += emit_len(dst + op, mcode - 15, lane); }
    else tok += mcode;
    if (lane == 0) dst[token_pos] = uint8_t(tok);
    ip += uint32_t(ml);
    anchor = ip;
    K3X(c, px_emit); K3CNT(c, nx_emit);
    return false;
}

// builder wave: clear the tables, then positions and windows in order, at most kRing - 1 windows ahead of the parser
__device__ void lz4hc_build_block(const uint8_t* src, int n, int attempts, uint8_t* work, uint32_t* score, HCWin* wins, HCSync* sync, int lane)
{
    HC c; c.src = src; c.lane = lane; c.ntu = 0; c.score = score; c.wins = wins; c.sync = sync; c.win = wins; c.wbase = 0xFFFFFFFFu;
    c.wmax = 0; c.failed = false; c.pl_byte = 0; c.pl_dst = 0; c.pl_n = 0;
    c.n = uint32_t(n); c.matchlimit = n > kLastLit ? uint32_t(n) - kLastLit : 0u; c.mflimit = n > kMfLimit ? uint32_t(n) - kMfLimit : 0u;
    c.heads = reinterpret_cast<uint32_t*>(work);
    c.chain = reinterpret_cast<uint16_t*>(work + (size_t(4) << kHashLog));
    {   // LZ4HC_clearTables: heads = 0, chain = 0xFFFF
        uint4* w = reinterpret_cast<uint4*>(work);
        const uint32_t nh = uint32_t((size_t(4) << kHashLog) / 16), nt = uint32_t(kWorkBytes / 16);
        for (uint32_t i = lane; i < nt; i += 64) w[i] = i < nh ? make_uint4(0, 0, 0, 0) : make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int i = lane; i < kScore; i += 64) score[i] = 0xFFFFFFFFu;
    }
    if (n < kMfLimit + 1) return;
    const uint32_t nwin = (c.mflimit >> 6) + 1;
    for (uint32_t v = 0; v < nwin; v++) {
        uint32_t keep;
        for (uint32_t spins = 0; v >= (keep = ld_acq(&sync->keep)) + kRing; ) {
            if (ld_acq(&sync->done)) return;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) return;
        }
        if (ld_acq(&sync->done)) return;
        c.win = wins + (v % kRing);
        if (v + 1 < keep) hc_insert(c, min((v << 6) + 64u, c.n - 3u));          // the parser is past it: tables only
        else hc_build_window(c, v << 6, attempts, true);
        if (lane == 0) st_rel(&sync->built, v + 1);
    }
}

// parser wave
__device__ int lz4hc_encode_block(const uint8_t* src, uint8_t* dst, int n, int cap, int attempts,
                                  uint8_t* work, HCWin* wins, HCSync* sync, int lane)
{
    if (uint32_t(n) > 0x7E000000u) return 0;
    HC c; c.src = src; c.lane = lane; c.ntu = 0; c.score = nullptr; c.wins = wins; c.sync = sync; c.win = wins; c.wbase = 0xFFFFFFFFu;
    c.wmax = 0; c.failed = false;
    c.pl_byte = 0; c.pl_dst = 0; c.pl_n = 0;
#ifdef K2_PROF
    c.pt_build = c.pt_search = c.pt_emit = 0; c.n_build = c.n_search = c.n_emit = c.n_mem = 0; c.pt0 = __builtin_readcyclecounter();
    c.px_emit = c.px_skip = c.px_first = 0; c.nx_emit = c.nx_skip = 0;
#endif
    c.n = uint32_t(n); c.matchlimit = n > kLastLit ? uint32_t(n) - kLastLit : 0u; c.mflimit = n > kMfLimit ? uint32_t(n) - kMfLimit : 0u;
    c.heads = reinterpret_cast<uint32_t*>(work);
    c.chain = reinterpret_cast<uint16_t*>(work + (size_t(4) << kHashLog));
    const bool limited = cap < n + n / 255 + 16;
    const uint32_t ucap = uint32_t(cap);
    uint32_t ip = 0, anchor = 0, op = 0;

    if (n >= kMfLimit + 1) {
        const uint32_t mflimit = uint32_t(n) - kMfLimit, matchlimit = uint32_t(n) - kLastLit;
        int ml, ml2, ml3, ml0;
        uint32_t ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0, start0, ref0, dummy = 0;
        while (ip <= mflimit) {
            // the search that opens a sequence looks back at nothing: its answer was prepared with the window
            if (c.wbase == 0xFFFFFFFFu || ip - c.wbase >= 64u) { K3PH(c, pt_emit); hc_acquire(c, ip, attempts); K3CNT(c, n_build); K3PH(c, pt_build); }
            if (c.failed) return 0;
            K3X0();
            {
                const uint32_t f = c.win->fm[ip - c.wbase];
                if (f != 0xFF) { ml = f ? int(f) : kMinMatch - 1; ref = c.win->fr[ip - c.wbase] - kIdx0; }
                else ml = hc_wider(c, ip, ip, matchlimit, kMinMatch - 1, ref, dummy, attempts);
            }
            if (ml < kMinMatch) {
                // no match here: go straight to the next window position that has a candidate sharing its first four
                // bytes - a search anywhere in between returns "none" without side effects (tables are filled ahead)
                uint32_t nip = ip + 1;
                if (c.wbase != 0xFFFFFFFFu && nip - c.wbase < 64u) {
                    const uint32_t rel = nip - c.wbase;
                    const unsigned long long m = __ballot(c.win->live[lane] != 0) >> rel;
                    nip += m ? uint32_t(__builtin_ctzll(m)) : 64u - rel;
                }
                ip = nip;
                K3X(c, px_skip); K3CNT(c, nx_skip);
                continue;
            }
            K3X(c, px_first);
            start0 = ip; ref0 = ref; ml0 = ml;
        search2:
            if (ip + ml <= mflimit) ml2 = hc_wider(c, ip + ml - 2, ip, matchlimit, ml, ref2, start2, attempts);
            else ml2 = ml;
            if (ml2 == ml) {
                if (hc_emit(c, src, dst, ip, op, anchor, ml, ref, limited, ucap, lane)) return 0;
                continue;
            }
            if (start0 < ip && start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
            if (start2 - ip < 3) { ml = ml2; ip = start2; ref = ref2; goto search2; }
        search3:
            if (start2 - ip < kOptimalML) {
                int new_ml = ml;
                if (new_ml > kOptimalML) new_ml = kOptimalML;
                if (ip + new_ml > start2 + ml2 - kMinMatch) new_ml = int(start2 - ip) + ml2 - kMinMatch;
                const int correction = new_ml - int(start2 - ip);
                if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
            }
            if (start2 + ml2 <= mflimit) ml3 = hc_wider(c, start2 + ml2 - 3, start2, matchlimit, ml2, ref3, start3, attempts);
            else ml3 = ml2;
            if (ml3 == ml2) {
                if (start2 < ip + ml) ml = int(start2 - ip);
                if (hc_emit(c, src, dst, ip, op, anchor, ml, ref, limited, ucap, lane)) return 0;
                ip = start2;
                if (hc_emit(c, src, dst, ip, op, anchor, ml2, ref2, limited, ucap, lane)) return 0;
                continue;
            }
            if (start3 < ip + ml + 3) {
                if (start3 >= ip + ml) {
                    if (start2 < ip + ml) {
                        const int correction = int(ip + ml - start2);
                        start2 += correction; ref2 += correction; ml2 -= correction;
                        if (ml2 < kMinMatch) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                    }
                    if (hc_emit(c, src, dst, ip, op, anchor, ml, ref, limited, ucap, lane)) return 0;
                    ip = start3; ref = ref3; ml = ml3;
                    start0 = start2; ref0 = ref2; ml0 = ml2;
                    goto search2;
                }
                start2 = start3; ref2 = ref3; ml2 = ml3;
                goto search3;
            }
            if (start2 < ip + ml) {
                if (start2 - ip < kOptimalML) {
                    if (ml > kOptimalML) ml = kOptimalML;
                    if (ip + ml > start2 + ml2 - kMinMatch) ml = int(start2 - ip) + ml2 - kMinMatch;
                    const int correction = ml - int(start2 - ip);
                    if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
                } else ml = int(start2 - ip);
            }
            if (hc_emit(c, src, dst, ip, op, anchor, ml, ref, limited, ucap, lane)) return 0;
            ip = start2; ref = ref2; ml = ml2;
            start2 = start3; ref2 = ref3; ml2 = ml3;
            goto search3;
        }
    }
    if (uint32_t(lane) < c.pl_n) dst[c.pl_dst + lane] = uint8_t(c.pl_byte);
    {   // last literals (lz4hc.c:735-762)
        const uint32_t run = uint32_t(n) - anchor, add = (run + 255 - 15) / 255;
        if (limited && op + 1 + add + run > ucap) return 0;
        if (run >= 15) { if (lane == 0) dst[op] = 0xF0; op++; op += emit_len(dst + op, run - 15, lane); }
        else { if (lane == 0) dst[op] = uint8_t(run << 4); op++; }
        copy_bytes(dst + op, src + anchor, run, lane);
        op += run;
    }
#ifdef K2_PROF
    if (lane == 0 && n == (4 << 20) && op + 256 < uint32_t(n)) {
        uint64_t* o = reinterpret_cast<uint64_t*>(dst + n - 128);
        o[0] = c.pt_build; o[1] = c.pt_search; o[2] = c.pt_emit; o[3] = c.n_build; o[4] = c.n_search; o[5] = c.n_mem;
        o[6] = c.px_emit; o[7] = c.px_skip; o[8] = c.px_first; o[9] = c.nx_emit; o[10] = c.nx_skip;
    }
#endif
    return int(op);
}

__global__ __launch_bounds__(128)
void lz4hc_encode_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks,
                         uint32_t nblocks, uint8_t* work_base, int attempts, int container_mode)
{
    __shared__ uint32_t score[kScore];
    __shared__ HCWin wins[kRing + 1];
    __shared__ HCSync sync;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    if (threadIdx.x == 0) { sync.built = 0; sync.keep = 0; sync.done = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const fourmc_block blk = uniform_block(blocks[b]);
    const uint8_t* src = src_base + blk.src_off;
    const int n = int(blk.src_len);
    uint8_t* work = work_base + size_t(b) * kWorkBytes;
    if (threadIdx.x >= 64) {
        if (uint32_t(n) <= 0x7E000000u) lz4hc_build_block(src, n, attempts, work, score, wins, &sync, lane);
        return;
    }
    uint8_t* dst = dst_base + blk.dst_off;
    const int cap = container_mode ? n - 1 : int(blk.dst_cap);
    int r = lz4hc_encode_block(src, dst, n, cap, attempts, work, wins, &sync, lane);
    if (lane == 0) st_rel(&sync.done, 1u);
    if (container_mode && r <= 0) { copy_bytes(dst, src, uint32_t(n), lane); r = n; }
    if (lane == 0) blocks[b].result = r;
}

} // namespace

extern "C" size_t fourmc_lz4hc_work_bytes(uint32_t n) { return size_t(n) * kWorkBytes; }

extern "C" hipError_t fourmc_launch_lz4hc_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                                 void* d_work, int level, int container_mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    static const int searches[9] = {2, 2, 2, 4, 8, 16, 32, 64, 128};       // lz4hc.c:817-827, levels 0..8
    if (level < 1 || level > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lz4hc_encode_kernel, dim3(n), dim3(128), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n,
                       static_cast<uint8_t*>(d_work), searches[level], container_mode);
    return hipGetLastError();
}
