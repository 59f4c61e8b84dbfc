// 4mc_amd/csrc/lz4_encode.hip — K2: batched LZ4 "fast" block encode on gfx950, BYTE-IDENTICAL to
// LZ4_compress_default of the reference's 64-bit little-endian build.
//
// Replaces native/4mc.c:301 (capacity n-1, stored fallback) and native/jniCompressor.c:91
// (capacity LZ4_compressBound) -> native/lz4/lz4.c:1435 -> :1416 -> :1346-1367 -> :910-1302.
//
// The reference parse is a serial greedy walk whose bytes depend on the exact order of hash
// table reads and overwrites (lz4.c:1059,1207,1247), on the probe stride schedule
// (`step = searchMatchNb++ >> 6`, :1017-1027), on backward extension (:1080), the ip-2 refill and
// the immediate re-test (:1207-1259).  It is reproduced exactly, 64 positions at a time:
//   * one wavefront owns one block; its hash table (4096 x u32, or 8192 x u16 for blocks
//     < 65547 B, lz4.c:1353) lives in LDS and is zeroed per block like LZ4_initStream (:1348).
//     20 KiB of LDS per block (table + the sparse batch's scoreboard): 8 blocks per CU, which is what a
//     2048-block launch needs (a 16 KiB variant that tagged the table in the sparse batch too
//     measured 11 % slower and was not kept);
//   * "dense window": lane l takes position sp+l whatever role the walk will give it, prepares
//     candidate / 4-byte test / match extents against the table as it stands; lanes sharing a
//     slot are found on the table itself (lane tags, atomic max); a scalar walk of one readlane
//     per sequence chooses the hit lanes; sizes, positions and all bytes of the chosen sequences
//     are then written by all lanes at once, and the visited lanes enter the table (latest
//     position of a slot wins);
//   * "sparse batch": lane l speculatively executes probe k0+l of the running search at its
//     strided position (block start / end, long searches); a ballot finds the FIRST lane (serial
//     order) that hits or runs into the end-of-block limit, only lanes before it commit their
//     table writes, and the batch is cut at the first lane that shares a (folded) slot with an
//     earlier one - an LDS atomic-min scoreboard; the cut lane becomes lane 0 of the next batch;
//   * whatever the registers do not hold (long matches, long catch-up, output nearly full) goes
//     through one general sequence routine with wave-wide compares and copies.
// Per block HBM traffic: n bytes read (+ candidate re-reads that hit L1/L2), csize written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devenc.h"

namespace {

constexpr int      kHashLog   = 12;          // lz4.h:654
constexpr int      kSmallLim  = 65536 + 11;  // lz4.c:689 LZ4_64Klimit
constexpr uint32_t kMaxDist   = 65535;       // lz4.h:633
constexpr int      kMfLimit   = 12, kLastLit = 5, kMinLen = 13;
constexpr int      kScore     = 1024;        // entries of the same-slot scoreboard

template <bool U32TAB> __device__ __forceinline__ uint32_t hash_at(const uint8_t* p)
{
    if (U32TAB) return uint32_t(((ld8(p) << 24) * 889523592379ULL) >> (64 - kHashLog));   // lz4.c:764-769
    return (ld4(p) * 2654435761u) >> (32 - (kHashLog + 1));                                 // lz4.c:758-759
}

// offset of probe K from the first probe of a search: steps are 1 for the first 65 probes, then
// grow by one every 64 probes (lz4.c:1014-1023)
__device__ __forceinline__ uint32_t probe_offset(uint32_t K)
{
    if (K == 0) return 0;
    const uint32_t T = K - 1, m = T >> 6, r = T & 63;
    return 1 + T + 32 * m * (m - 1) + m * r;
}

__device__ __forceinline__ uint32_t U(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }   // pin to an SGPR

// 16 bytes at p, any alignment, as four dwords (one global_load_dwordx4)
struct Q16 { uint32_t d0, d1, d2, d3; };
__device__ __forceinline__ Q16 ld16(const uint8_t* p) { const U16B t = *reinterpret_cast<const U16B*>(p); return Q16{uint32_t(t.a), uint32_t(t.a >> 32), uint32_t(t.b), uint32_t(t.b >> 32)}; }
__device__ __forceinline__ uint64_t u64(uint32_t lo, uint32_t hi) { return (uint64_t(hi) << 32) | lo; }
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); }

// wave64 inclusive prefix sum on the DPP network (row shifts, then row broadcasts)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t scan_add(uint32_t v)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}

template <bool U32TAB> __device__ __forceinline__ uint32_t hash_of(uint64_t v)
{
    if (U32TAB) return uint32_t(((v << 24) * 889523592379ULL) >> (64 - kHashLog));
    return (uint32_t(v) * 2654435761u) >> (32 - (kHashLog + 1));
}

template <bool U32TAB>
__device__ int lz4_encode_block(const uint8_t* src, uint8_t* dst, const int n, const int cap,
                                uint32_t* tab32, uint32_t* score, const int lane)
{
    uint16_t* tab16 = reinterpret_cast<uint16_t*>(tab32);
#ifdef K2_PROF   // one-off phase profile (tools/k2_phases.py builds a side library with -DK2_PROF): cycles per phase
    uint64_t pt_search = 0, pt_ext = 0, pt_match = 0, pt_cur = 0, pt_tab = 0, pt_gather = 0, pt_emit = 0, pt_nwin = 0, pt_nseq = 0, pt_nslow = 0, pt_none = 0, pt_cross = 0, pt_dcut = 0, pt_scut = 0, pt_prep = 0, pt_gen = 0, pt_nit = 0, pt0 = __builtin_readcyclecounter(), pt1;
#define K2PH(acc) do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); pt1 = __builtin_readcyclecounter(); acc += pt1 - pt0; pt0 = pt1; __builtin_amdgcn_sched_barrier(0); } while (0)
#define K2CNT(c) do { c++; } while (0)
#else
#define K2PH(acc) do { } while (0)
#define K2CNT(c) do { } while (0)
#endif
    const bool limited = cap < n + n / 255 + 16;                       // lz4.c:1352
    uint32_t op = 0, anchor = 0;

    for (int i = lane; i < (1 << kHashLog); i += 64) tab32[i] = 0;     // LZ4_initStream
    for (int i = lane; i < kScore; i += 64) score[i] = 0xFFFFFFFFu;

    // A 32-bit table entry is position << cbits | check, the check being cbits hash bits of the 4 bytes at that
    // position: a probe whose check differs cannot be a 4-byte match and skips the (random, cache-line wide) read
    // of its candidate.  Entries with position 0 (empty slots read as position 0, lz4.c:1348) are always read.
    // Entries stay below 2^31: values above that are lane tags, written for a moment while a dense window finds out
    // which of its lanes share a slot.
    const int cbits = U32TAB ? max(__builtin_clz(uint32_t(n - 1) | 1u) - 1, 0) : 0;
    const uint32_t cmask = (1u << cbits) - 1;
    auto chk_of = [&](uint32_t w) -> uint32_t { return (U32TAB && cbits) ? (w * 2654435761u) >> (32 - cbits) : 0u; };
    auto enc = [&](uint32_t pos, uint32_t ck) -> uint32_t { return U32TAB ? (pos << cbits) | ck : pos; };
    auto tab_get = [&](uint32_t h) -> uint32_t { return U32TAB ? tab32[h] : uint32_t(tab16[h]); };
    auto tab_put = [&](uint32_t h, uint32_t pos, uint32_t ck) { if (U32TAB) tab32[h] = enc(pos, ck); else tab16[h] = uint16_t(pos); };

    if (n >= kMinLen) {
        const uint32_t un = uint32_t(n);
        const uint32_t lim = un - kMfLimit + 1;                        // mflimitPlusOne
        const uint32_t matchlimit = un - kLastLit;
        if (lane == 0) tab_put(hash_at<U32TAB>(src), 0, chk_of(ld4(src)));

        // One sequence (lz4.c:1080-1200): catch-up, literals, offset, match length.  `ip`/`cand` are the hit; `back`
        // (0..4) bytes before them and `fwd` bytes after ip+4 are already known equal, `*_more` says the comparison
        // has to go on in memory.  Literals at [wsp, wsp+64) come from `wbyte` (lane l holds src[wsp+l]).
        // Returns 0 and the end of the match in ip_out, 1 = go to the last literals, 2 = output full.
        auto sequence = [&](uint32_t ip, uint32_t cand, const bool rt_hit, const uint32_t back, const bool back_more,
                            uint32_t fwd, const bool fwd_more, const uint32_t wsp, const uint32_t wbyte, uint32_t& ip_out) -> int {
            uint32_t token_pos, tok;
            if (rt_hit) { token_pos = op++; tok = 0; }                 // lz4.c:1250-1256: zero literals, straight to _next_match
            else {
                const uint32_t maxback = min(ip - anchor, cand);
                uint32_t b = min(back, maxback);
                if (back_more && maxback > back) {
                    for (;;) {
                        const uint32_t j = b + uint32_t(lane) + 1;
                        const bool ok = (j <= maxback) && src[ip - j] == src[cand - j];
                        const unsigned long long bad = ~__ballot(ok);
                        const int t = bad ? __builtin_ctzll(bad) : 64;
                        b += t;
                        if (t < 64) break;
                    }
                }
                b = U(b);
                ip = U(ip - b); cand = U(cand - b);
                fwd += b;                                              // everything in [new ip, old ip + 4 + fwd) is equal
                const uint32_t lit = ip - anchor;                      // literals (lz4.c:1083-1107)
                token_pos = op++;
                if (limited && op + lit + (2 + 1 + kLastLit) + lit / 255 > uint32_t(cap)) return 2;
                if (lit >= 15) { tok = 0xF0; op += emit_len(dst + op, lit - 15, lane); }
                else tok = lit << 4;
                if (wsp != 0xFFFFFFFFu && ip > wsp && ip <= wsp + 64) {
                    if (anchor < wsp) copy_bytes(dst + op, src + anchor, wsp - anchor, lane);
                    const uint32_t q = wsp + lane;
                    if (q >= anchor && q < ip) dst[op + (q - anchor)] = uint8_t(wbyte);
                } else copy_bytes(dst + op, src + anchor, lit, lane);
                op += lit;
            }
            // _next_match (lz4.c:1109-1200)
            const uint32_t off = ip - cand, off_pos = op;
            op += 2;
            uint32_t mcode = fwd;
            if (fwd_more) {
                uint32_t a = ip + 4 + mcode, b = cand + 4 + mcode;
                for (;;) {
                    if (a + 1024 <= matchlimit && mcode >= 64) {
                        const U16B x = *reinterpret_cast<const U16B*>(src + a + 16 * lane);
                        const U16B y = *reinterpret_cast<const U16B*>(src + b + 16 * lane);
                        const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
                        const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3)
                                               : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
                        const unsigned long long bad = __ballot(eq < 16);
                        if (bad) {
                            const int l = __builtin_ctzll(bad);
                            mcode += 16 * l + __builtin_amdgcn_readlane(eq, l);
                            break;
                        }
                        mcode += 1024; a += 1024; b += 1024;
                    } else {
                        const uint32_t i = a + lane;
                        const bool same = (i < matchlimit) && src[i] == src[b + lane];
                        const unsigned long long bad = ~__ballot(same);
                        if (bad) { mcode += __builtin_ctzll(bad); break; }
                        mcode += 64; a += 64; b += 64;
                    }
                }
            }
            mcode = U(mcode);
            ip = U(ip + mcode + 4);
            if (limited && op + (1 + kLastLit) + (mcode + 240) / 255 > uint32_t(cap)) return 2;
            uint32_t tok_add;
            if (mcode >= 15) { tok_add = 15; op += emit_len(dst + op, mcode - 15, lane); }
            else tok_add = mcode;
            if (lane < 3) {                                            // token, offset low, offset high: one store
                const uint32_t at = lane == 0 ? token_pos : off_pos + uint32_t(lane) - 1;
                const uint32_t v  = lane == 0 ? tok + tok_add : (lane == 1 ? off : off >> 8);
                dst[at] = uint8_t(v);
            }
            anchor = ip; op = U(op);
            ip_out = ip;
            return ip >= lim ? 1 : 0;
        };

        // Parser state: `sp` is the next position to act on.  With `retest` it is the position right after a match,
        // whose ip-2 refill and immediate re-test (lz4.c:1207-1259) are still owed and whose search starts at sp+1;
        // otherwise it is probe number k0 of the running search.
        uint32_t sp = 1, k0 = 0; bool retest = false;
        uint32_t pw_sp = 0xFFFFFFFFu, pwbyte = 0;                      // previous dense window: start, and the byte each lane held
        for (;;) {
            if (sp >= 4 && sp + 132 <= un && k0 <= 32) {
                // ------------------------------------------------------------ dense window
                // Probes 0..65 of a search are one byte apart, so around a fresh search the parse walks consecutive
                // positions.  Lane l takes position sp+l, whatever role the parse will give it (probe, re-test,
                // refill, or inside a match), and prepares everything against the table as it stands: candidate,
                // 4-byte test, up to 4 equal bytes backwards and 24 (56 for the lanes that need it) forwards.  A scalar loop of ballots and
                // readlanes then only CHOOSES the sequences of the window (which hit lanes the greedy walk lands
                // on); sizes, output positions and all bytes of the chosen sequences are produced afterwards by all
                // lanes at once.  A lane whose table slot is also touched by an earlier lane of the window
                // ("dirty") cannot trust its candidate: when the walk reaches it as a probe the window ends there.
                const uint32_t sp0 = sp;
                const uint32_t pos = sp0 + uint32_t(lane);
                const Q16 q0 = ld16(src + pos - 4), q1 = ld16(src + pos + 12);          // [pos-4, pos+28)
                K2PH(pt_cur); K2CNT(pt_nwin);
                const uint64_t v8 = u64(q0.d1, q0.d2);
                const uint32_t h = hash_of<U32TAB>(v8);
                uint32_t h2 = 0xFFFFFFFFu;                              // slot of the owed refill of sp-2
                const uint32_t ck = chk_of(q0.d1);
                uint32_t ck2 = 0;
                if (retest) {
                    const uint32_t w2 = __builtin_amdgcn_alignbit(q0.d1, q0.d0, 16);
                    h2 = rl(hash_of<U32TAB>(u64(w2, __builtin_amdgcn_alignbit(q0.d2, q0.d1, 16))), 0);
                    ck2 = rl(chk_of(w2), 0);
                }
                if (retest && lane == 0) tab_put(h2, sp0 - 2, ck2);    // the owed refill, ahead of every read of this window
                const uint32_t ent = tab_get(h);
                const uint32_t c = U32TAB ? ent >> cbits : ent;
                const bool pass = !U32TAB || (c + kMaxDist >= pos && (c == 0 || (ent & cmask) == ck));
                // Lanes that share a table slot.  32-bit table: the slot itself is the scoreboard - every lane tags it
                // (atomic max of a value above any entry: the lowest lane wins), the losers tag once more among
                // themselves, the winner puts the entry back.  That gives, exactly: the first lane of a slot (clean),
                // the second (its only predecessor in the window is `pred`), and later ones ("dirty": the window ends
                // in front of one the walk reaches as a probe).  16-bit table: a folded min-scoreboard, first lane
                // clean, all others dirty.
                bool dirty, second = false; int pred = 0;
                if (U32TAB) {
                    atomicMax(&tab32[h], 0x80000000u | uint32_t(63 - lane));
                    pred = 63 - int(tab32[h] & 63);
                    const bool first = pred == lane;
                    dirty = false;
                    if (__ballot(!first)) {
                        if (!first) atomicMax(&tab32[h], 0x80000040u | uint32_t(63 - lane));
                        second = !first && 63 - int(tab32[h] & 63) == lane;
                        dirty = !first && !second;
                    }
                    if (first) tab32[h] = ent;
                } else {
                    uint32_t* sc = &score[h & (kScore - 1)];
                    atomicMin(sc, uint32_t(lane));
                    dirty = (*sc != uint32_t(lane));
                    *sc = 0xFFFFFFFFu;
                }
                K2PH(pt_tab);
                // info: lane of the match end | back bytes << 8 (5: four and possibly more) | raw back << 12 |
                // back_more << 15 | fwd_more << 16
                uint32_t info;
                bool hit = false, slow = false;
                uint32_t fw = 0, bk = 0;
                if (pass && c >= 4) {
                    const Q16 c0 = ld16(src + c - 4), c1 = ld16(src + c + 12);
                    hit = c0.d1 == q0.d1;
                    const uint32_t xb = q0.d0 ^ c0.d0;                  // byte 3 (MSB) is position -1
                    bk = xb ? uint32_t(__builtin_clz(xb) >> 3) : 4u;
                    const uint64_t x0 = u64(q0.d2, q0.d3) ^ u64(c0.d2, c0.d3), x1 = u64(q1.d0, q1.d1) ^ u64(c1.d0, c1.d1),
                                   x2 = u64(q1.d2, q1.d3) ^ u64(c1.d2, c1.d3);
                    fw = x0 ? uint32_t(__builtin_ctzll(x0) >> 3) : (x1 ? 8u + uint32_t(__builtin_ctzll(x1) >> 3)
                                           : (x2 ? 16u + uint32_t(__builtin_ctzll(x2) >> 3) : 24u));
                }
                // matches that run past the 28 bytes held: those lanes (only) read 32 more of both sides
                const bool sat = hit && fw == 24;
                if (__ballot(sat)) {
                    if (sat) {
                        const Q16 q2 = ld16(src + pos + 28), q3 = ld16(src + pos + 44), c2 = ld16(src + c + 28), c3 = ld16(src + c + 44);
                        const uint64_t x3 = u64(q2.d0, q2.d1) ^ u64(c2.d0, c2.d1), x4 = u64(q2.d2, q2.d3) ^ u64(c2.d2, c2.d3),
                                       x5 = u64(q3.d0, q3.d1) ^ u64(c3.d0, c3.d1), x6 = u64(q3.d2, q3.d3) ^ u64(c3.d2, c3.d3);
                        fw = x3 ? 24u + uint32_t(__builtin_ctzll(x3) >> 3) : (x4 ? 32u + uint32_t(__builtin_ctzll(x4) >> 3)
                                : (x5 ? 40u + uint32_t(__builtin_ctzll(x5) >> 3) : (x6 ? 48u + uint32_t(__builtin_ctzll(x6) >> 3) : 56u)));
                    }
                }
                if (!pass) info = uint32_t(lane) + 4;
                else if (c >= 4) {
                    slow = fw == 56;
                    info = (uint32_t(lane) + 4 + fw) | ((bk == 4 && c > 4 ? 5u : bk) << 8) | (bk << 12) | (uint32_t(bk == 4) << 15) | (uint32_t(fw == 56) << 16);
                } else {
                    hit = ld4(src + c) == q0.d1;
                    slow = true;                                        // nothing known: compare in memory
                    info = (uint32_t(lane) + 4) | (1u << 15) | (1u << 16);
                }
                // A second lane of a slot sees its predecessor's position instead of the table entry if the walk made the
                // predecessor a probe.  Hit on neither: an ordinary probe.  Otherwise it is decided when the walk gets there.
                const uint32_t wpred = __shfl(q0.d1, pred);              // (every lane takes part: the source lane must be active)
                const bool hitA = second && wpred == q0.d1;
                const unsigned long long m_cond = __ballot(second && (hitA || hit));
                const unsigned long long m_hit = __ballot(hit) & ~m_cond, m_dirty = __ballot(dirty), m_slow = __ballot(hit && slow);
                const uint32_t wbyte = q0.d1;
                K2PH(pt_gather);
                // The first sequence may own literals of the previous window: they are still in registers if that was
                // a dense window too (pw_sp); anything older goes through the general path.  Near the end of the
                // output buffer every sequence does, for its exact capacity checks.
                const uint32_t pw_lo = (pw_sp != 0xFFFFFFFFu && pw_sp + 64 >= sp0) ? pw_sp : sp0;
                const bool capok = !limited || op + 320 <= uint32_t(cap);
                unsigned long long stop2 = m_dirty | m_slow | m_cond | (capok ? 0ull : m_hit);
                unsigned long long stop1 = stop2;
                const int scut = (!retest && k0 > 1) ? 65 - int(k0) : 64;   // first lane that is not one byte on from its predecessor
                if (scut < 64) stop1 |= 1ull << scut;
                if (anchor < pw_lo) stop1 |= m_hit & (0ull - m_hit);
                const unsigned long long hd2 = m_hit | stop2;
                unsigned long long hd = m_hit | stop1, stp = stop1;
                unsigned long long selw = 0, xint = 0;                  // chosen hit lanes; lanes inside directly emitted matches
                int cur = 0, anc = int(anchor - sp0);                   // lane of the next action; anchor as a lane number
                bool any = retest, fresh = true;                        // a search has begun in this window; no sequence yet
                bool gen = false;                                       // take one general step next (a resolved second lane)
                int stoplane = 64, status = 0;
                // Where the walk goes from lane l when a search starts there with nothing pending (anchor == l), for
                // every l at once: the end of the first hit at or after l (| that hit lane << 8), or 0x8000 | the lane
                // it has to stop at (64: none).  The scalar walk below is then one readlane per sequence.
                uint32_t nxt;
                {
                    const unsigned long long shm = (m_hit | stop2) >> lane;
                    const int E = lane + (shm ? __builtin_ctzll(shm) : 0);
                    const uint32_t infE = __shfl(info | (uint32_t((stop2 >> lane) & 1) << 17), E);
                    const bool term = !shm || ((infE >> 17) & 1) || (((infE >> 8) & 7) == 5 && E - lane >= 5);
                    nxt = term ? (0x8000u | (shm ? uint32_t(E) : 64u)) : ((infE & 255) | (uint32_t(E) << 8));
                }
                // the same for a lane as a chosen HIT: where the walk lands behind its match, and what it does there
                uint32_t nextE, fin;
                {
                    const uint32_t land = info & 255;
                    const uint32_t nl = __shfl(nxt, int(land & 63));
                    const bool endn = land >= 64 || (nl & 0x8000u);
                    nextE = endn ? uint32_t(lane) : (nl >> 8) & 63;
                    fin = land >= 64 ? 0x4000u : nl;
                }
                K2PH(pt_prep);
                for (;;) {
                    unsigned long long sel = 0;
                    K2CNT(pt_nit);
                    int reason = -1, e = 0;                             // 0: no event left, 1: stop at lane e, 2: match leaves the window
                    const int anc0 = anc;
                    if (gen || anc != cur || hd != hd2) {
                        gen = false;
                        // the running search began before this window, or first-sequence stops apply: one general step
                        const unsigned long long ev = hd & (~0ull << cur);
                        if (!ev) reason = 0;
                        else {
                            e = __builtin_ctzll(ev);
                            const uint32_t inf = rl(info, e);
                            if (((stp >> e) & 1) || min(int((inf >> 8) & 7), e - anc) == 5) reason = 1;   // 5: the catch-up goes on in memory
                            else {
                                sel = 1ull << e;
                                anc = cur = int(inf & 255);
                                hd = hd2; stp = stop2;
                                if (cur >= 64) reason = 2;
                            }
                        }
                    }
                    K2PH(pt_gen);
                    if (reason < 0) {
                        // The walk proper, over the HIT lanes: nextE is the next chosen hit behind a hit (itself when taking it ends
                        // the walk), so the chain is a straight run of s_bitset1 + v_readlane pairs without a branch per sequence
                        // (re-marking the last hit is harmless), tested for its end every six; `fin` says what ended it.
                        K2CNT(pt_nseq);
                        const uint32_t t = rl(nxt, cur);
                        if (t & 0x8000u) { e = int(t & 127); reason = e == 64 ? 0 : 1; }
                        else {
                            uint32_t E = (t >> 8) & 63;
                            for (;;) {
                                asm volatile("s_bitset1_b64 %0, %1" : "+s"(sel) : "s"(E));
                                const uint32_t n1 = rl(nextE, int(E));
                                if (n1 == E) break;
                                E = n1;
                            }
                            asm volatile("s_bitset1_b64 %0, %1" : "+s"(sel) : "s"(E));
                            const uint32_t f = rl(fin, int(E));
                            cur = anc = int(rl(info, int(E)) & 255);
                            if (f & 0x4000u) reason = 2;
                            else { e = int(f & 127); reason = e == 64 ? 0 : 1; }
                        }
                    }
                    K2PH(pt_ext);
                    if (sel) {
                        // sizes, positions and bytes of the chosen sequences, all at once (lz4.c:1080-1200)
                        const bool mine = (sel >> lane) & 1;
                        const unsigned long long below = sel & ~(~0ull << lane);
                        const int P = below ? 63 - __builtin_clzll(below) : 0;
                        const int endP = int(__shfl(info, P) & 255);
                        const int myanc = below ? endP : anc0;           // my anchor: the end of the sequence chosen before me
                        const uint32_t b = uint32_t(min(int((info >> 8) & 7), lane - myanc));
                        const uint32_t lit = uint32_t(lane - myanc) - b, mc = (info & 255) - uint32_t(lane) - 4 + b;
                        const uint32_t xl = lit >= 15 ? 1u : 0u;
                        const uint32_t size = mine ? 3 + lit + xl + (mc >= 15 ? 1u : 0u) : 0u;
                        const uint32_t incl = scan_add(size);
                        const uint32_t opb = op + incl - size;
                        const uint32_t pk = uint32_t(myanc + 128) | (b << 9) | (xl << 12);
                        // a literal lane belongs to the next chosen hit lane E at or after it, if it lies in that
                        // sequence's [anchor, ip)
                        const unsigned long long ahead = sel >> lane;
                        const int E = lane + (ahead ? __builtin_ctzll(ahead) : 0);
                        const uint32_t opbE = __shfl(opb, E), pkE = __shfl(pk, E);
                        const int aE = int(pkE & 511) - 128, ipnE = E - int((pkE >> 9) & 7);
                        if (ahead && lane >= aE && lane < ipnE) dst[opbE + 1 + ((pkE >> 12) & 1) + uint32_t(lane - aE)] = uint8_t(wbyte);
                        const int E1 = __builtin_ctzll(sel);
                        const int a1 = int(rl(pk, E1) & 511) - 128;
                        if (a1 < 0) {                                   // literals still held by the previous window's lanes
                            const uint32_t pk1 = rl(pk, E1), opb1 = rl(opb, E1);
                            const int pl = int(pw_sp - sp0) + lane, ipn1 = E1 - int((pk1 >> 9) & 7);
                            if (pl >= a1 && pl < min(0, ipn1)) dst[opb1 + 1 + ((pk1 >> 12) & 1) + uint32_t(pl - a1)] = uint8_t(pwbyte);
                        }
                        if (mine) {
                            const uint32_t off = pos - c;
                            uint32_t o = opb;
                            dst[o++] = uint8_t((min(lit, 15u) << 4) | min(mc, 15u));
                            if (xl) dst[o++] = uint8_t(lit - 15);
                            o += lit;
                            dst[o] = uint8_t(off); dst[o + 1] = uint8_t(off >> 8);
                            if (mc >= 15) dst[o + 2] = uint8_t(mc - 15);
                        }
                        op = U(op + rl(incl, 63));
                        anchor = sp0 + uint32_t(anc);
                        selw |= sel; any = true; fresh = false;
                    }
                    K2PH(pt_emit);
                    if (reason == 0) {                                  // every remaining lane was a probe
                        K2CNT(pt_none);
                        const int s = any ? anc + 1 : -int(k0);
                        k0 = uint32_t(64 - s); sp = sp0 + 64; retest = false;
                        break;
                    }
                    if (reason == 2) {
                        const uint32_t ip = sp0 + uint32_t(cur);
                        if (ip >= lim) { status = 1; break; }
                        sp = ip; k0 = 0; retest = true;                 // the next window pays the refill
                        K2CNT(pt_cross);
                        break;
                    }
                    // A dirty lane that is the re-test right after a match, and whose slot the refill two bytes back has
                    // just written (byte runs: same five bytes): its candidate is that refill, nothing else can have
                    // written the slot since.
                    const bool run_rt = ((m_dirty >> e) & 1) && any && e == anc && e >= 2 && rl(h, e - 2) == rl(h, e);
                    if (!run_rt && (((m_dirty >> e) & 1) || (fresh && e == scut))) { // the window ends in front of lane e
                        stoplane = e;
                        if ((m_dirty >> e) & 1) K2CNT(pt_dcut); else K2CNT(pt_scut);
                        if (any && e == anc) { k0 = 0; retest = true; }
                        else { k0 = uint32_t(e - (any ? anc + 1 : -int(k0))); retest = false; }
                        sp = sp0 + uint32_t(e);
                        break;
                    }
                    uint32_t s_cand = rl(c, e), s_inf = rl(info, e);
                    if (run_rt) {
                        if (rl(q0.d1, e - 2) != rl(q0.d1, e)) {         // no match: the search goes on behind the re-test
                            cur = e + 1;
                            if (cur < 64) continue;
                            k0 = uint32_t(64 - (anc + 1)); sp = sp0 + 64; retest = false;
                            break;
                        }
                        s_cand = sp0 + uint32_t(e) - 2; s_inf = (uint32_t(e) + 4) | (1u << 15) | (1u << 16);
                    } else if ((m_cond >> e) & 1) {
                        // second lane of its slot: did the walk put its predecessor j into the table?  Yes if j is a probe
                        // of the running search (j >= anchor lane) or lies outside every match chosen so far.
                        const int j = int(rl(uint32_t(pred), e));
                        const unsigned long long below = selw & ~(~0ull << lane);
                        const int P = below ? 63 - __builtin_clzll(below) : 0;
                        const int endP = int(__shfl(info, P) & 255);
                        const bool inside = (below && lane < endP && lane != endP - 2) || ((xint >> lane) & 1);
                        const bool visited = j >= anc || !rl(uint32_t(inside), j);
                        const bool use = visited ? rl(uint32_t(hitA), e) : rl(uint32_t(hit), e);
                        if (!use) {                                     // an ordinary probe after all: the search goes on behind it
                            cur = e + 1;
                            if (cur < 64) continue;
                            k0 = uint32_t(64 - (any ? anc + 1 : -int(k0))); sp = sp0 + 64; retest = false;
                            break;
                        }
                        if (visited) { s_cand = sp0 + uint32_t(j); s_inf = (uint32_t(e) + 4) | (1u << 15) | (1u << 16); }
                        else if (capok && !((m_slow >> e) & 1) && min(int((s_inf >> 8) & 7), e - anc) != 5) {   // a plain hit on its table entry: back to the walk, which takes it as such
                            stop2 &= ~(1ull << e); stp &= ~(1ull << e); gen = true;
                            continue;
                        }
                    }
                    {   // a hit that needs the general path
                        K2CNT(pt_nslow);
                        const uint32_t inf = s_inf;
                        uint32_t ip;
                        status = sequence(sp0 + uint32_t(e), s_cand, any && e == anc, (inf >> 12) & 7, (inf >> 15) & 1,
                                          (inf & 255) - uint32_t(e) - 4, (inf >> 16) & 1, sp0, wbyte, ip);
                        if (status) break;
                        any = true; fresh = false;
                        if (ip - sp0 >= 64) {
                            xint |= ~1ull << e;
                            sp = ip; k0 = 0; retest = true;
                            break;
                        }
                        cur = anc = int(ip - sp0);
                        xint |= (~1ull << e) & ~(~0ull << cur) & ~(1ull << (cur - 2));
                        hd = hd2; stp = stop2;
                    }
                }
                if (status == 2) return 0;
                if (status == 1) goto last_literals;
                // commit: every lane the walk passed as a probe, re-test or refill enters the table; the latest
                // position of a slot wins (a refill lane may share its slot with an earlier probe)
                {
                    const unsigned long long below = selw & ~(~0ull << lane);
                    const int P = below ? 63 - __builtin_clzll(below) : 0;
                    const int endP = int(__shfl(info, P) & 255);
                    const bool inside = (below && lane < endP && lane != endP - 2) || ((xint >> lane) & 1);
                    if (lane < stoplane && !inside) {
                        if (U32TAB) atomicMax(&tab32[h], enc(pos, ck));
                        else for (;;) { tab16[h] = uint16_t(pos); asm volatile("" ::: "memory"); if (uint32_t(tab16[h]) >= pos) break; }
                    }
                }
                pw_sp = sp0; pwbyte = wbyte;
                sp = U(sp); k0 = U(k0);
                K2PH(pt_match);
                continue;
            }
            // ---------------------------------------------------------------- sparse batch (lz4.c:1014-1076)
            // 64 probes of the running search at their strided positions: start and end of a block, and searches
            // that found nothing for a while.  The first lane (serial order) that hits or reaches the end limit ends
            // the batch; it is cut at the first lane that shares a table slot with an earlier one.
            {
                const bool rt = retest;                                 // lane 0 is the re-test of sp, the search starts at sp+1
                pw_sp = 0xFFFFFFFFu;
                uint32_t pos, next;
                if (k0 == 0) { pos = sp + lane; next = pos + 1; }
                else { pos = sp + probe_offset(k0 + lane) - probe_offset(k0); next = sp + probe_offset(k0 + lane + 1) - probe_offset(k0); }
                const bool in_range = next <= lim;                     // else: this probe ends the block
                const bool wide = sp >= 4;
                const uint32_t rp = in_range ? pos : (wide ? 4u : 0u);   // keep speculative reads in bounds
                uint64_t v8, ipx = 0; uint32_t ipb = 0;
                if (wide) { const Q16 q = ld16(src + rp - 4); ipb = q.d0; v8 = u64(q.d1, q.d2); ipx = u64(q.d2, q.d3); }
                else v8 = ld8(src + rp);
                const uint32_t h = hash_of<U32TAB>(v8);
                uint32_t h2 = 0xFFFFFFFFu;
                const uint32_t ck = chk_of(uint32_t(v8));
                uint32_t ck2 = 0;
                if (rt) {
                    const uint32_t w2 = __builtin_amdgcn_alignbit(uint32_t(v8), ipb, 16);
                    h2 = rl(hash_of<U32TAB>(u64(w2, __builtin_amdgcn_alignbit(uint32_t(v8 >> 32), uint32_t(v8), 16))), 0);
                    ck2 = rl(chk_of(w2), 0);
                }
                uint32_t ent = tab_get(h);
                if (h == h2) ent = enc(sp - 2, ck2);                   // the refill of ip-2 comes before every probe of this batch
                const uint32_t c = U32TAB ? ent >> cbits : ent;
                const bool pass = in_range && (!U32TAB || (c + kMaxDist >= pos && (c == 0 || (ent & cmask) == ck)));
                uint32_t* sc = &score[h & (kScore - 1)];
                atomicMin(sc, uint32_t(lane));
                const bool shared = (*sc != uint32_t(lane));
                *sc = 0xFFFFFFFFu;
                uint32_t cw = 0, cb = 0; uint64_t cx = 0; bool cregs = false;
                if (!pass) { }
                else if (wide && c >= 4) { const Q16 q = ld16(src + c - 4); cb = q.d0; cw = q.d1; cx = u64(q.d2, q.d3); cregs = true; }
                else cw = ld4(src + c);
                const bool hit = pass && cw == uint32_t(v8);
                const unsigned long long m_cut  = __ballot(shared);
                const unsigned long long m_term = __ballot(!in_range);
                const unsigned long long m_hit  = __ballot(hit);
                const int cut = m_cut ? __builtin_ctzll(m_cut) : 64;
                const unsigned long long ev = (m_term | m_hit) & ((cut >= 64) ? ~0ull : ((1ull << cut) - 1));
                const int e = ev ? __builtin_ctzll(ev) : cut;
                const bool e_is_hit = ev && ((m_hit >> e) & 1) && !((m_term >> e) & 1);
                if (rt && lane == 0) tab_put(h2, sp - 2, ck2);
                if (lane < e || (lane == e && e_is_hit)) tab_put(h, pos, ck);
                K2PH(pt_search);
                if (ev) {
                    if (!e_is_hit) goto last_literals;
                    const uint32_t ip = rl(pos, e), cand = rl(c, e);
                    uint32_t back = 0, fwd = 0; bool back_more = true, fwd_more = true;
                    if (rl(uint32_t(cregs), e)) {
                        const uint32_t xb = rl(ipb, e) ^ rl(cb, e);
                        back = xb ? uint32_t(__builtin_clz(xb) >> 3) : 4u;
                        back_more = back == 4;
                        const uint64_t x = u64(rl(uint32_t(ipx), e), rl(uint32_t(ipx >> 32), e)) ^ u64(rl(uint32_t(cx), e), rl(uint32_t(cx >> 32), e));
                        const uint32_t eq = x ? uint32_t(__builtin_ctzll(x) >> 3) : 8u;
                        const uint32_t room = matchlimit - (ip + 4);   // ip < lim  =>  room >= 3
                        fwd = min(eq, room);
                        fwd_more = !((eq < 8) || (room <= 8));
                    }
                    uint32_t ipn;
                    const int status = sequence(ip, cand, rt && e == 0, back, back_more, fwd, fwd_more, 0xFFFFFFFFu, 0, ipn);
                    if (status == 2) return 0;
                    if (status == 1) goto last_literals;
                    sp = U(ipn); k0 = 0; retest = true;
                    K2PH(pt_search);
                    continue;
                }
                // nobody hit: lanes [0, e) were probes (lane 0 possibly the re-test)
                sp = U(sp + (k0 == 0 ? uint32_t(e) : probe_offset(k0 + uint32_t(e)) - probe_offset(k0)));
                k0 = U(k0 + uint32_t(e) - (rt ? 1u : 0u));
                retest = false;
                K2PH(pt_search);
            }
        }
    }
last_literals:
    {
        const uint32_t run = uint32_t(n) - anchor;                      // lz4.c:1266-1293
        if (limited && op + run + 1 + (run + 255 - 15) / 255 > uint32_t(cap)) return 0;
        if (run >= 15) {
            if (lane == 0) dst[op] = 0xF0;
            op++;
            op += emit_len(dst + op, run - 15, lane);
        } else {
            if (lane == 0) dst[op] = uint8_t(run << 4);
            op++;
        }
        copy_bytes(dst + op, src + anchor, run, lane);
        op += run;
    }
#ifdef K2_PROF
    if (lane == 0 && n == (4 << 20)) { uint64_t* c = reinterpret_cast<uint64_t*>(dst + n - 32); c[0] = pt_search; c[1] = pt_ext; c[2] = pt_match; uint64_t* d = reinterpret_cast<uint64_t*>(dst + n - 128); d[0] = pt_cur; d[1] = pt_tab; d[2] = pt_gather; d[3] = pt_emit; d[4] = pt_nwin; d[5] = pt_nseq; d[6] = pt_nslow; uint64_t* f = reinterpret_cast<uint64_t*>(dst + n - 192); f[0] = pt_none; f[1] = pt_cross; f[2] = pt_dcut; f[3] = pt_scut; f[4] = pt_prep; f[5] = pt_gen; f[6] = pt_nit; }
#endif
    return int(op);
}

// container_mode = 0: result = LZ4_compress_default(src, dst, src_len, dst_cap).
// container_mode = 1: one iteration of fourMCcompressFilename's loop (native/4mc.c:301-329):
//   capacity src_len-1; a result <= 0 stores the block raw (payload = input, result = src_len).
__global__ __launch_bounds__(64)
void lz4_encode_fast_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                            fourmc_block* blocks, uint32_t nblocks, int container_mode)
{
    __shared__ uint32_t tab[1 << kHashLog];
    __shared__ uint32_t score[kScore];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    const int n = int(blk.src_len);
    const int cap = container_mode ? n - 1 : int(blk.dst_cap);
    const int lane = threadIdx.x;
    int r;
    if (uint32_t(n) > 0x7E000000u) r = 0;                               // lz4.c:1324
    else if (n == 0) {                                                  // lz4.c:1325-1335
        const bool limited = cap < 16;
        if (limited && cap <= 0) r = 0; else { if (lane == 0) dst[0] = 0; r = 1; }
    }
    else if (n < kSmallLim) r = lz4_encode_block<false>(src, dst, n, cap, tab, score, lane);
    else                    r = lz4_encode_block<true>(src, dst, n, cap, tab, score, lane);
    if (container_mode && r <= 0) {
        copy_bytes(dst, src, uint32_t(n), lane);
        r = n;
    }
    if (lane == 0) blocks[b].result = r;
}

} // namespace

extern "C" hipError_t fourmc_launch_lz4_encode_fast(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                                    uint32_t n, int container_mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4_encode_fast_kernel, dim3(n), dim3(64), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n,
                       container_mode);
    return hipGetLastError();
}
