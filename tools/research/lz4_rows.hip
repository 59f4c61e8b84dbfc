// 4mc_amd/csrc/lz4_rows.hip — K1r: row-parallel LZ4 block decode on gfx950 (wave64), the default fast path of
// fourmc_launch_lz4_decode since round 3.
//
// Replaces LZ4_decompress_safe(in, out, csize, usize) (native/4mc.c:661, native/jniDecompressor.c:88 ->
// native/lz4/lz4.c:2345-2350 -> :1936-2339) for every block whose stream is regular; anything else (rule
// violations, odd end-of-block shapes, sizes beyond 4 MiB) is handed to the exact walker of lz4_decode.hip
// (result = kRetry), which reproduces the reference's accept / reject set and negative return codes.
//
// What is serial in an LZ4 block is only WHERE the tokens are.  The compressed stream is cut into aligned rows of 64
// bytes and four waves per block work as a pipeline through LDS:
//   PRE   every byte of a row is decoded as if a token started there (literal count, match length, offset, next
//         token); five rounds of pointer doubling over ds_bpermute give, for each of the 64 possible entry points of
//         the row, where the token chain leaves the row and how many literal / match bytes it produces on the way
//         (one packed dword per entry: the row's TABLE).  No dependence on other rows: rows are processed four at a
//         time, straight from global memory (unaligned dword loads).
//   WALK  the only serial chain left: one LDS look-up per row (entry -> exit, output position, match-space position).
//         Tokens the tables do not cover (length continuations beyond one byte, the end of the block) are decoded
//         here one at a time under the strict rules of the reference's safe loop (lz4.c:2120-2330).
//   POST  once a row's entry point is known: marks the row's true tokens (a v_readlane chain), prefix-sums their
//         sizes (DPP), stores the row's LITERALS straight from the stream bytes the lanes hold (a byte scatter: the
//         output position of a literal is its stream position plus a per-token constant, carried to the lanes by one
//         max-scan), and publishes one record per token for the copier: {offset, where its match goes}.
//   COPY  executes the matches in MATCH SPACE: the match bytes of all sequences laid end to end.  Lane l of a step
//         produces match byte g + l; its owner token is found by a max-scan over a ring of start marks, source and
//         destination follow from the token's record.  Lanes whose source is already stored go first, the others
//         follow in passes (overlapping matches read the period instead: no dependence inside a match).
// Literal stores (POST) and match loads (COPY) meet in the CU's L1; a record is published to the copier only after
// the stores of its literals have completed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include "lz4par.h"

namespace {

constexpr int kRetry = lz4par::kRetryCode;
constexpr int kNR  = 8;            // rows in the table / field ring
constexpr int kNQ  = 16;           // records in flight between POST / WALK and COPY
constexpr int kOwn = 2048;         // match-space ring of start marks (entries)
constexpr int kSpan = 512;         // output bytes one copier window may span
constexpr uint32_t kMaxRowLits = 2047, kMaxRowMatch = 511;    // a row beyond these goes token by token (four rows of marks fit the ring)
constexpr int kTailGuard = 16 + 337 + 64;   // rows whose tokens may touch the last 16 stream bytes are never batched
constexpr uint32_t kSpinLimit = 1u << 19;
#ifndef K1R_NAP
#define K1R_NAP 6                   // s_sleep units (64 clk) between two looks at a progress word
#endif
#ifndef K1R_NAP_PRE
#define K1R_NAP_PRE 32              // PRE waiting for ring space, WALK waiting for tables: the roles that run ahead of the others
#endif
#ifndef K1R_NAP_WALK
#define K1R_NAP_WALK 16
#endif
#ifndef K1R_STEPS
#define K1R_STEPS 4                 // 64-lane sub-steps the copier keeps in flight per iteration
#endif

struct RowSlot { uint32_t tab[64]; uint4 fld[64]; };        // fld: {x, y, token-chain mask of the entry (lo, hi)}
constexpr int kRecTok = 24;        // token records per row record (a row of 64 bytes holds 21 tokens at most)
struct Shared {
    uint16_t own[kOwn];            // own[m & (kOwn-1)] = mark of the token whose match starts at match-space position m (see COPY)
    RowSlot  rows[kNR];
    uint2    rec[kNQ][kRecTok];    // per record and token (by rank in its row): {offset | mpos << 16, D | (mpos >> 16) << 22 | overlap << 31}
    uint4    res[kNR][2];          // WALK -> POST, per visited row: {row + 1, entry | q << 6, output position, match-space position}, {chain mask lo, hi, -, -}
    uint2    pub[kNQ];             // POST / WALK -> COPY, per record: {q + 1 once ready, end of the record in match space}
    uint16_t scr[kSpan + 8];         // COPY: the output bytes [bound0, bound0 + kSpan) of the window being executed: 0 literal /
                                   // older data (in memory), 0x200 match byte still to come, 0x100 | byte once produced
    // progress words; pairs that one reader wants together sit in one 8-byte word
    uint2    pw;                   // x: PRE, rows [0, x) have tables and fields in the ring;  y: WALK, rows [0, y) are decided
    uint2    cc;                   // COPY: x records consumed, y match-space bytes consumed
    uint32_t post_rows;            // POST: rows [0, post_rows) are done with their ring slots
    uint32_t total_q;              // WALK: number of records of the block once known, else 0xFFFFFFFF
    uint32_t failed;
    int      end_value;
};
static_assert(sizeof(Shared) <= 20480, "eight blocks per CU need <= 20 KiB of LDS each");

__device__ __forceinline__ uint32_t ldv(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void stv(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return uint32_t(__builtin_amdgcn_readlane(int(v), int(l))); }
// LDS operations of one wave execute in order; what has to be stopped is the compiler moving them
#define LDS_ORDER() asm volatile("" ::: "memory")

// per-role cycle counters of the profiling side build (make rprof; tools/k1r_prof.py): slot 7 = the role's whole time,
// slots 0..6 = time spent in its wait sites / counts
#ifdef K1R_PROF
struct Prof {
    unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void add(int i, unsigned long long since) { t[i] += __builtin_amdgcn_s_memtime() - since; }
    __device__ __forceinline__ void count(int i, unsigned long long n = 1) { t[i] += n; }
};
#else
struct Prof {
    __device__ __forceinline__ unsigned long long now() const { return 0; }
    __device__ __forceinline__ void add(int, unsigned long long) {}
    __device__ __forceinline__ void count(int, unsigned long long = 1) {}
};
#endif

__device__ __forceinline__ uint2 ldv2(const uint2* p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return make_uint2(rfl(uint32_t(v)), rfl(uint32_t(v >> 32)));
}
// bounded wait on LDS words: `poll` re-reads what `cond` looks at; false = somebody failed (or the wait ran out): the
// role leaves and the exact kernel decides
template <int NAP = K1R_NAP, class P, class F> __device__ __forceinline__ bool wait_until(Shared* S, Prof& pf, int site, P poll, F cond)
{
    if (cond()) return true;
    poll();
    if (cond()) { LDS_ORDER(); return true; }
    const unsigned long long t0 = pf.now();
    for (uint32_t spins = 0;;) {
        // a waiting wave costs the working ones issue slots and LDS reads: nap between looks (the roles that run ahead nap longer),
        // look at the failure word only now and then
        if ((spins & 7) == 0 && rfl(ldv(&S->failed))) return false;
        __builtin_amdgcn_s_sleep(NAP);
        poll();
        if (cond()) break;
        if (++spins > kSpinLimit) { stv(&S->failed, 1); return false; }
    }
    pf.add(site, t0);
    LDS_ORDER();
    return true;
}

// every byte the roles touch in memory is GLOBAL memory: say so in the pointer types (a generic pointer costs FLAT
// instructions, which also count on lgkmcnt and make every later wait a full one)
typedef __attribute__((address_space(1))) uint8_t gbyte;
typedef __attribute__((address_space(1))) const uint8_t cgbyte;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld4u(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32_u*>(p); }
__device__ __forceinline__ u32x4 ld16g(cgbyte* p) { return *reinterpret_cast<__attribute__((address_space(1))) const u32x4_u*>(p); }
__device__ __forceinline__ void st16g(gbyte* p, u32x4 v) { *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(p) = v; }

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t scan_add(uint32_t v)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t scan_max(uint32_t v)
{
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v)); v = max(v, dpp0<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ uint32_t bperm(uint32_t addr4, uint32_t v) { return uint32_t(__builtin_amdgcn_ds_bpermute(int(addr4), int(v))); }

// ---------------------------------------------------------------------------------------------------------- PRE
// field word x: offset | stream byte << 24;   y: match | overlap << 9 | regular << 10 | lit-ext << 11 | literals << 16;   z, w: chain mask
__device__ void role_pre(Shared* S, Prof& pf, cgbyte* src, int csize, int lane)
{
    const int nrows = (csize + 63) >> 6;
    const int lastpos = csize - 4;
    int post_seen = 0;
    auto request = [&](int r0, uint32_t (&w)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = ld4u(src + min((r0 + u) * 64 + lane, lastpos));
    };
    auto group = [&](int r0, const uint32_t (&w)[4]) -> bool {
        post_seen = int(rfl(uint32_t(post_seen)));
        if (!wait_until<K1R_NAP_PRE>(S, pf, 0, [&] { post_seen = int(rfl(ldv(&S->post_rows))); }, [&] { return post_seen + kNR >= r0 + 4; })) return false;
        uint32_t offpos[4], L[4], lext[4], wo[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t b = w[u] & 0xff, b1 = (w[u] >> 8) & 0xff, L0 = b >> 4;
            lext[u] = L0 == 15 ? 1u : 0u;
            L[u] = L0 + (lext[u] ? b1 : 0u);
            offpos[u] = uint32_t(lane) + 1 + lext[u] + L[u];
            wo[u] = ld4u(src + min((r0 + u) * 64 + int(offpos[u]), lastpos));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t b = w[u] & 0xff, b1 = (w[u] >> 8) & 0xff, M0 = b & 15;
            const uint32_t off = wo[u] & 0xffff, e1 = (wo[u] >> 16) & 0xff;
            const bool mext = M0 == 15;
            const uint32_t ml = M0 + 4 + (mext ? e1 : 0u);
            const bool reg = !(lext[u] && b1 == 255) && !(mext && e1 == 255);
            const uint32_t nxt = reg ? offpos[u] + 2 + (mext ? 1u : 0u) : uint32_t(lane);
            const uint32_t pz = reg ? (L[u] << 16 | ml) : 0u;
            const bool nonterm = reg && nxt < 64;
            const uint32_t h0 = nonterm ? nxt : uint32_t(lane);
            // pointer doubling over ds_bpermute.  One register carries the hop AND the sizes summed along it: literals << 19 |
            // match bytes << 6 | hop lane (21 tokens of <= 269 / 273 bytes: 13 bits each), so a round moves both with one permute;
            // the regular tokens on the chain that starts at each lane travel beside it as a 64-bit mask (what POST needs of a row)
            const uint32_t pzp = (pz >> 16) << 19 | (pz & 0xffff) << 6;
            uint32_t v = (nonterm ? pzp : 0u) | h0, h4 = h0 << 2;
            // (the mask starts with the lane itself and the lane it hops to: k rounds cover 2^k hops, the last node included;
            // an irregular token may get its bit - it has no sizes, which is what POST takes a token by)
            uint32_t mlo = ((reg && lane < 32) ? 1u << lane : 0u) | (h0 < 32 ? 1u << h0 : 0u);
            uint32_t mhi = ((reg && lane >= 32) ? 1u << (lane - 32) : 0u) | (h0 >= 32 ? 1u << (h0 - 32) : 0u);
            auto round = [&] {
                const uint32_t t = bperm(h4, v);
                mlo |= bperm(h4, mlo); mhi |= bperm(h4, mhi);
                v = (v & ~63u) + t;                     // sums add, the hop becomes the hop's hop
                h4 = (t & 63) << 2;
            };
#pragma unroll
            for (int k = 0; k < 4; k++) round();       // 2^4 hops: chains of up to 16 tokens
            if (__ballot(bperm(h4, h4) != h4)) round(); // (uniform) a chain of 17..21 tokens needs the fifth round
            const uint32_t tz = bperm(h4, pzp);
            const uint32_t tot = (v & ~63u) + (tz & ~63u);
            uint32_t X = bperm(h4, nxt);
            uint32_t lits = tot >> 19, mls = (tot >> 6) & 0x1fff;
            if (lits > kMaxRowLits || mls > kMaxRowMatch) { X = uint32_t(lane); lits = 0; mls = 0; }
            RowSlot& rs = S->rows[(r0 + u) & (kNR - 1)];
            rs.tab[lane] = X << 21 | lits << 10 | mls;
            const uint32_t ovl = (reg && off < ml) ? 1u : 0u;
            rs.fld[lane] = make_uint4(off | b << 24, pz | ovl << 9 | (reg ? 1u : 0u) << 10 | lext[u] << 11, mlo, mhi);
        }
        LDS_ORDER();
        if (lane == 0) stv(&S->pw.x, uint32_t(r0 + 4));
        return true;
    };
    uint32_t wa[4], wb[4];
    request(0, wa);
    for (int r0 = 0; r0 < nrows; r0 += 8) {
        // rows the walk has passed already (it copies long literal runs itself) need no tables: go on where it is
        const int walked = int(min(rfl(ldv(&S->pw.y)), uint32_t(nrows)));
        if (walked >= r0 + 16) {
            r0 = (walked & ~7) - 8;                                     // the loop adds 8 (nobody asks for the tables of the rows passed over)
            request(r0 + 8, wa);
            continue;
        }
        request(r0 + 4, wb);
        if (!group(r0, wa)) return;
        if (r0 + 4 >= nrows) break;
        request(r0 + 8, wa);
        if (!group(r0 + 4, wb)) return;
    }
}

// ---------------------------------------------------------------------------------------------------------- WALK
// literal runs the walk copies itself (general sequences): 16 bytes per lane, two granules in flight; a small register
// footprint matters more here than the last GB/s (all four roles share one register allocation: 64 VGPRs for 8 waves / SIMD)
__device__ __attribute__((noinline)) void lean_copy(gbyte* dst, cgbyte* src, int n, int lane)
{
    const int head = min(n, int((16 - (uintptr_t(dst) & 15)) & 15));
    if (lane < head) dst[lane] = src[lane];
    int k = head;
    for (; k + 2048 <= n; k += 2048) {
        const u32x4 v0 = ld16g(src + k + 16 * lane), v1 = ld16g(src + k + 1024 + 16 * lane);
        st16g(dst + k + 16 * lane, v0);
        st16g(dst + k + 1024 + 16 * lane, v1);
    }
    for (; k + 1024 <= n; k += 1024) { const u32x4 v = ld16g(src + k + 16 * lane); st16g(dst + k + 16 * lane, v); }
    for (; k < n; k += 64) { const int i = k + lane; if (i < n) dst[i] = src[i]; }
}
struct Win {                       // 64 stream bytes around the position the walk decodes token by token
    cgbyte* src; int csize; int la_pos; uint32_t la; int lane;
    __device__ __forceinline__ void reload(int p) { la_pos = p; const int a = p + lane; la = a < csize ? uint32_t(src[a]) : 0u; }
    __device__ __forceinline__ uint32_t get(int p) { if (p < la_pos || p >= la_pos + 64) reload(p); return rdl(la, uint32_t(p - la_pos)); }
};
// length continuation bytes (lz4.c:1903-1928): whole runs of 255 per step via a ballot over the window
__device__ __forceinline__ bool more_len(Win& s, int& ip, int lim, bool check_first, int& len)
{
    if (check_first && ip >= lim) return false;
    for (;;) {
        if (ip < s.la_pos || ip >= s.la_pos + 64) s.reload(ip);
        const int l0 = ip - s.la_pos;
        const unsigned long long not255 = __ballot(s.la != 255u) >> l0;
        const int avail = 64 - l0;
        int n, add; bool done;
        if (not255 == 0) { n = avail; add = 255 * avail; done = false; }
        else { const int t = __builtin_ctzll(not255); n = t + 1; add = 255 * t + int(rdl(s.la, uint32_t(l0 + t))); done = true; }
        if (ip + n > lim) return false;
        ip += n;
        len = len > 0x40000000 - add ? 0x40000000 : len + add;
        if (done) return true;
    }
}

__device__ void role_walk(Shared* S, Prof& pf, cgbyte* src, gbyte* dst, int csize, int cap, int lane)
{
    const int iend = csize, oend = cap;
    const int rlast = (csize - kTailGuard) >> 6;            // negative: no row is batched
    Win win; win.src = src; win.csize = csize; win.lane = lane; win.la = 0; win.la_pos = -(1 << 30);
    int p = 0, op = 0;
    uint32_t mb = 0, q = 0, wpos = 0;
    int row_general = -1, pre_seen = 0, post_seen = 0;
    uint32_t cq_seen = 0, cg_seen = 0;
    auto fail = [&]() { if (lane == 0) stv(&S->failed, 1); };
    auto decided = [&](uint32_t upto) { if (upto > wpos) { wpos = upto; LDS_ORDER(); if (lane == 0) stv(&S->pw.y, wpos); } };
    for (;;) {
        p = int(rfl(uint32_t(p))); op = int(rfl(uint32_t(op))); mb = rfl(mb); q = rfl(q); wpos = rfl(wpos);
        pre_seen = int(rfl(uint32_t(pre_seen))); post_seen = int(rfl(uint32_t(post_seen))); cq_seen = rfl(cq_seen); cg_seen = rfl(cg_seen);
        const int r = p >> 6; const uint32_t e = uint32_t(p) & 63;
        decided(uint32_t(r));
        if (r <= rlast && r != row_general) {
            if (!wait_until<K1R_NAP_WALK>(S, pf, 0, [&] { pre_seen = int(rfl(ldv(&S->pw.x))); }, [&] { return pre_seen > r; })) return;
            const uint32_t t = rfl(S->rows[r & (kNR - 1)].tab[e]);
            const uint4 fe = S->rows[r & (kNR - 1)].fld[e];               // same round trip: the entry's chain mask
            const uint32_t X = t >> 21, lits = (t >> 10) & 0x7ff, mls = t & 0x3ff;
            if (X != e && op + int(lits + mls) + 80 <= oend) {
                if (!wait_until(S, pf, 1, [&] { post_seen = int(rfl(ldv(&S->post_rows))); }, [&] { return post_seen + kNR > r; })) return;
                if (lane == 0) { S->res[r & (kNR - 1)][1] = make_uint4(fe.z, fe.w, 0, 0); S->res[r & (kNR - 1)][0] = make_uint4(uint32_t(r + 1), e | q << 6, uint32_t(op), mb); }
                decided(uint32_t(r + 1));
                op += int(lits + mls); mb += mls; p = r * 64 + int(X); q++;
                continue;
            }
        }
        // ------------------------------------------------------------ one sequence under the strict rules
        row_general = r;
        pf.count(4);
        const unsigned long long tg = pf.now();
        int ip = p;
        if (ip >= iend) { fail(); return; }
        const uint32_t token = win.get(ip); ip++;
        int lit = int(token >> 4), mlen = int(token & 15);
        if (lit == 15) { if (!more_len(win, ip, iend - 15, true, lit)) { fail(); return; } }
        if (op + lit > oend - 12 || ip + lit > iend - 8) {
            if (ip + lit != iend || op + lit > oend) { fail(); return; }
            lean_copy(dst + op, src + ip, lit, lane);                  // the block's last sequence: literals only
            if (lane == 0) S->end_value = op + lit;
            break;
        }
        const int lit_ip = ip;
        ip += lit;
        const int op2 = op + lit;
        const int off = int(win.get(ip)) | (int(win.get(ip + 1)) << 8);
        ip += 2;
        if (mlen == 15) { if (!more_len(win, ip, iend - 4, false, mlen)) { fail(); return; } }
        mlen += 4;
        if (off == 0 || off > op2 || op2 + mlen > oend - 5) { fail(); return; }
        lean_copy(dst + op, src + lit_ip, lit, lane);
        if (!wait_until(S, pf, 2, [&] { const uint2 c = ldv2(&S->cc); cq_seen = c.x; cg_seen = c.y; },
                        [&] { return cq_seen + kNQ > q && mb < cg_seen + kOwn; })) return;
        if (lane == 0) {
            S->own[mb & (kOwn - 1)] = uint16_t(((q & 511) << 6) + 1);
            S->rec[q & (kNQ - 1)][0] = make_uint2(uint32_t(off) | mb << 16,
                                                  uint32_t(op2 - int(mb)) | (mb >> 16) << 22 | (off < mlen ? 1u : 0u) << 31);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the literals are in memory before the record is seen
        if (lane == 0) S->pub[q & (kNQ - 1)] = make_uint2(q + 1, mb + uint32_t(mlen));
        op = op2 + mlen; mb += uint32_t(mlen); p = ip; q++;
        pf.add(3, tg);
    }
    LDS_ORDER();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) { stv(&S->total_q, q); stv(&S->pw.y, 0xFFFFFFFFu); }
}

// ---------------------------------------------------------------------------------------------------------- POST
// Four rows per iteration, each complete in itself (a literal run that leaves its row is fetched from the stream by the
// row that owns the token), so that the four dependent chains of a row - the token walk, two DPP scans, the LDS round
// trips - overlap with those of its neighbours instead of adding up.
__device__ void role_post(Shared* S, Prof& pf, cgbyte* src, gbyte* dst, int csize, int lane)
{
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const int nrows = (csize + 63) >> 6;
    uint32_t pend_q = kNone, pend_m = 0;             // lanes 0..3: the records of the previous group, their literal stores issued
    uint32_t pre_seen = 0, wpos_seen = 0, cq_seen = 0, cg_seen = 0;      // progress of the other roles as last read
    uint32_t clean_from = 0, clean_to = 0;                               // rows [clean_from, clean_to) are known not to be visited
    (void)pre_seen;
    auto fail = [&]() { if (lane == 0) stv(&S->failed, 1); };
    auto publish = [&]() {
        LDS_ORDER();
        if (lane < 4 && pend_q != kNone) S->pub[pend_q & (kNQ - 1)] = make_uint2(pend_q + 1, pend_m);
        pend_q = kNone;
    };
    auto poll_pw = [&] { const uint2 v = ldv2(&S->pw); pre_seen = v.x; wpos_seen = v.y; };
    for (int R = 0, n = 0; R < nrows; R += n) {
        wpos_seen = rfl(wpos_seen); cq_seen = rfl(cq_seen); cg_seen = rfl(cg_seen); clean_from = rfl(clean_from); clean_to = rfl(clean_to);
        unsigned long long tp = pf.now();
        // the group: the rows the walk has decided, four at most (waiting for a full group could wait for the walk while the
        // walk waits for the copier and the copier for this group's records)
        if (wpos_seen < uint32_t(R + 4)) {
            poll_pw();
            if (wpos_seen <= uint32_t(R)) {          // nothing decided yet: nothing to overlap the drain with
                if (__ballot(pend_q != kNone)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish(); }
                if (!wait_until(S, pf, 7, poll_pw, [&] { return wpos_seen > uint32_t(R); })) return;
            }
        }
        n = int(min(wpos_seen - uint32_t(R), 4u));
        // The walk writes a row's word only while that row is less than kNR rows above post_rows: seen from row R, the decided rows
        // from R + kNR on were not visited (the walk is past them and never returns).  Long literal runs make such stretches: pass them.
        if (wpos_seen > clean_to && uint32_t(R + kNR) < wpos_seen) {
            if (uint32_t(R + kNR) > clean_to) clean_from = uint32_t(R + kNR);
            clean_to = wpos_seen;
        }
        if (uint32_t(R) >= clean_from && uint32_t(R) < clean_to) {
            n = int(min(clean_to, uint32_t(nrows)) - uint32_t(R));
            LDS_ORDER();
            if (lane == 0) stv(&S->post_rows, uint32_t(R + n));
            continue;
        }
        LDS_ORDER();
        pf.add(0, tp); tp = pf.now();
        // one round trip: the walk's words and the fields of the four rows
        // (the progress word is read first: rows it shows as produced are in the ring when the field reads below execute)
        const unsigned long long pwv = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&S->pw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        LDS_ORDER();
        const uint32_t f4x = S->rows[(R + 4) & (kNR - 1)].fld[lane].x;   // stream bytes of the row behind the group (literal runs reaching into it)
        uint4 rs[4], rm[4]; uint2 f[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            rs[u] = S->res[(R + u) & (kNR - 1)][0]; rm[u] = S->res[(R + u) & (kNR - 1)][1];
            f[u] = *reinterpret_cast<const uint2*>(&S->rows[(R + u) & (kNR - 1)].fld[lane]);
        }
        bool vis[4]; uint32_t q[4], mb[4]; int op[4]; unsigned long long tokmask[4];
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            vis[u] = u < n && rfl(rs[u].x) == uint32_t(R + u + 1);
            q[u] = rfl(rs[u].y) >> 6; op[u] = int(rfl(rs[u].z)); mb[u] = rfl(rs[u].w);
            tokmask[u] = vis[u] ? ((unsigned long long)rfl(rm[u].y) << 32 | rfl(rm[u].x)) : 0ull;
            any |= vis[u];
        }
        pf.add(1, tp); tp = pf.now();
        if (any) {
            pf.add(2, tp); tp = pf.now();
            uint32_t pz[4], incl[4], excl[4], mls_t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pz[u] = ((tokmask[u] >> lane) & 1) ? (f[u].y & 0x01FF01FFu) : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) incl[u] = scan_add(pz[u]);
            uint32_t q_last = 0, m_end = 0; unsigned long long bad = 0;
            uint32_t D[4], mpos[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                excl[u] = incl[u] - pz[u];
                mls_t[u] = rdl(incl[u], 63) & 0xffff;
                D[u] = uint32_t(op[u]) - mb[u] + (incl[u] >> 16);        // destination of a match byte = match-space position + D
                mpos[u] = mb[u] + (excl[u] & 0xffff);
                const uint32_t off = f[u].x & 0xffff;
                bad |= __ballot(pz[u] != 0 && (off == 0 || off > mpos[u] + D[u]));
                if (vis[u]) { q_last = q[u]; m_end = mb[u] + mls_t[u]; }
            }
            if (bad) { fail(); return; }
            pf.add(3, tp); tp = pf.now();
            auto room = [&] { return cq_seen + kNQ > q_last && m_end <= cg_seen + kOwn; };
            auto poll_cc = [&] { const uint2 c = ldv2(&S->cc); cq_seen = c.x; cg_seen = c.y; };
            if (!room()) {
                poll_cc();
                if (!room()) {                       // the copier has to catch up: it may be waiting for records held back here
                    if (__ballot(pend_q != kNone)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish(); }
                    if (!wait_until(S, pf, 7, poll_cc, room)) return;
                }
            }
            pf.add(4, tp); tp = pf.now();
            // the previous group's literal stores were issued a whole group ago: complete them (normally no wait) and show its
            // records to the copier BEFORE this group's stores go out, so that no count of issued stores is needed
            if (__ballot(pend_q != kNone)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish(); }
            // ---- records for the copier (rows that were not visited have no tokens: nothing is written for them)
            uint32_t new_q = kNone, new_m = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (pz[u] != 0) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(tokmask[u] >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(tokmask[u]), 0u));
                    S->own[mpos[u] & (kOwn - 1)] = uint16_t(((q[u] & 511) << 6 | rank) + 1);
                    S->rec[q[u] & (kNQ - 1)][rank] = make_uint2((f[u].x & 0xffff) | mpos[u] << 16,
                                                                D[u] | (mpos[u] >> 16) << 22 | ((f[u].y >> 9) & 1) << 31);
                }
                if (vis[u] && lane == u) { new_q = q[u]; new_m = mb[u] + mls_t[u]; }
            }
            // ---- literals: the latest token at or before a stream lane owns it (max-scan of keys ordered by literal start)
            uint32_t km[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t litlo = uint32_t(lane) + 1 + ((f[u].y >> 11) & 1);
                const uint32_t dl = (excl[u] >> 16) + (excl[u] & 0xffff) + 512 - litlo;   // output position of stream lane 0, relative to op - 512
                km[u] = scan_max(pz[u] != 0 ? (litlo << 23 | (pz[u] >> 16) << 14 | dl) : 0u);
            }
            const uint32_t pre_now = rfl(uint32_t(pwv));
#pragma unroll
            for (int u = 0; u < 4; u++) {                               // rows without tokens: no lane passes the test
                if (uint32_t(lane) - (km[u] >> 23) < ((km[u] >> 14) & 0x1ff)) dst[lane + int(km[u] & 0x3fff) + op[u] - 512] = uint8_t(f[u].x >> 24);
            }
            // the row's last token: its literals may run on into the next rows.  Their first 64 bytes are the next row's stream
            // bytes - held by this wave (the next row of the group, or the row behind it), lane for lane
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k63 = rdl(km[u], 63);
                const uint32_t lo63 = k63 >> 23, end63 = lo63 + ((k63 >> 14) & 0x1ff);      // row-relative stream positions
                if (end63 <= 64) continue;                              // (uniform) nothing outside the row
                int from_pos = int(max(lo63, 64u));
                if (pre_now > uint32_t(R + u + 1)) {
                    const uint32_t nb = (u < 3 ? f[u < 3 ? u + 1 : 0].x : f4x) >> 24;
                    if (uint32_t(lane + 64) - lo63 < ((k63 >> 14) & 0x1ff)) dst[lane + 64 + int(k63 & 0x3fff) + op[u] - 512] = uint8_t(nb);
                    from_pos = 128;
                }
                if (end63 > uint32_t(from_pos)) {                       // rare: beyond what is at hand - straight from the stream
                    cgbyte* from = src + (R + u) * 64;
                    gbyte* to = dst + int(k63 & 0x3fff) + op[u] - 512;
                    for (int k = from_pos + lane; k < int(end63); k += 64) { const uint8_t v = from[k]; to[k] = v; }
                }
            }
            pf.add(5, tp); tp = pf.now();
            pend_q = new_q; pend_m = new_m;
        }
        LDS_ORDER();
        if (lane == 0) stv(&S->post_rows, uint32_t(R + n));
        pf.add(6, tp);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish();
}

// ---------------------------------------------------------------------------------------------------------- COPY
// Start marks are 16 bits: ((q & 511) << 6 | token lane) + 1, 0 = none.  The copier clears the marks it has consumed, so
// every mark it reads belongs to a record in flight, and compares them relative to the oldest unconsumed record (at
// most kNQ records are in flight: the 9 bits of q never wrap inside one comparison).
template <int K>
__device__ void role_copy(Shared* S, Prof& pf, gbyte* dst, int lane)
{
    uint32_t g = 0, qa = 0, cqv = 0, ext = 0, ckraw = 0, idle = 0;
    for (uint32_t spins = 0;;) {
        g = rfl(g); qa = rfl(qa); cqv = rfl(cqv); ext = rfl(ext); ckraw = rfl(ckraw); idle = rfl(idle);
        // one round trip: the publication words of all record slots (lanes 0..7) and the start marks of the next K * 64
        // match-space positions
        unsigned long long tc = pf.now();
        const bool look = ext - g < 64u * K;                              // a full window is there already: no need to look for records
        uint2 pb = make_uint2(0, 0);
        if (look) pb = S->pub[lane & (kNQ - 1)];
        LDS_ORDER();                                   // issued in this order: a record seen ready has its marks in place
        uint32_t mark[K];
#pragma unroll
        for (int u = 0; u < K; u++) mark[u] = S->own[(g + 64u * u + uint32_t(lane)) & (kOwn - 1)];
        if (look) {   // records consumed so far: [cqv, qa) with an end at or below g
            const uint32_t ql = cqv + ((uint32_t(lane) - cqv) & (kNQ - 1));
            cqv += uint32_t(__builtin_popcountll(__ballot(lane < kNQ && ql < qa && pb.y <= g)));
            // records that became ready: slots qa, qa + 1, ... in a row
            const uint32_t qn = qa + ((uint32_t(lane) - qa) & (kNQ - 1));
            const uint32_t ok = uint32_t(__ballot(lane < kNQ && pb.x == qn + 1)) & ((1u << kNQ) - 1);
            const uint32_t rot = uint32_t(((unsigned long long)ok | (unsigned long long)ok << kNQ) >> (qa & (kNQ - 1))) & ((1u << kNQ) - 1);
            const uint32_t n = uint32_t(__builtin_ctz(~rot));             // 0..kNQ
            if (n) { ext = rdl(pb.y, (qa + n - 1) & (kNQ - 1)); qa += n; }
        }
        LDS_ORDER();
        if (lane == 0) *reinterpret_cast<volatile unsigned long long*>(&S->cc) = (unsigned long long)cqv | (unsigned long long)g << 32;
        const bool ended = rfl(ldv(&S->total_q)) == qa;
        if (ext == g) {
            if (ended) break;                                           // every record consumed
            if (rfl(ldv(&S->failed))) return;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) { stv(&S->failed, 1); return; }
            continue;
        }
        spins = 0; idle = 0;
        pf.count(1); pf.count(2, min(ext - g, 64u * K));
        uint32_t avail = min(ext - g, 64u * K);
        // ---- owners.  Everything below is written as batches of K independent operations without branches in between:
        // the compiler issues the K LDS / memory operations of a batch back to back and waits once (a branch around each
        // would serialise them, one round trip after the other)
        // marks are compared relative to the record before the oldest unconsumed one: the carried owner (of the last byte
        // consumed) may belong to it
        const uint32_t base = ((cqv - 1) & 511) << 6;
        uint32_t dest[K], sp[K], km[K], nl[K];
#pragma unroll
        for (int u = 0; u < K; u++) {
            nl[u] = avail > 64u * u ? min(avail - 64u * u, 64u) : 0u;
            km[u] = scan_max(mark[u] ? ((mark[u] - 1 - base) & 0x7FFF) + 1 : 0u);
        }
        uint32_t ck = ckraw ? ((ckraw - 1 - base) & 0x7FFF) + 1 : 0u;
#pragma unroll
        for (int u = 0; u < K; u++) { km[u] = max(km[u], ck); const uint32_t last = rdl(km[u], nl[u] ? nl[u] - 1 : 0u); ck = nl[u] ? last : ck; }
        uint2 rc[K];
#pragma unroll
        for (int u = 0; u < K; u++) {
            const uint32_t x = km[u] - 1 + base;                        // (q & 511) << 6 | rank
            rc[u] = S->rec[(x >> 6) & (kNQ - 1)][min(x & 63, uint32_t(kRecTok - 1))];
        }
        unsigned long long anyovl = 0;
#pragma unroll
        for (int u = 0; u < K; u++) {
            const uint32_t m = g + 64u * u + uint32_t(lane);
            dest[u] = m + (rc[u].y & 0x3FFFFF);
            sp[u] = dest[u] - (rc[u].x & 0xffff);
            anyovl |= __ballot(int(rc[u].y) < 0 && uint32_t(lane) < nl[u]);
        }
        if (anyovl) {                                                   // overlapping matches: read the period, not the match itself
#pragma unroll
            for (int u = 0; u < K; u++) {
                const uint32_t m = g + 64u * u + uint32_t(lane);
                const uint32_t off = rc[u].x & 0xffff, D = rc[u].y & 0x3FFFFF;
                const uint32_t mpos = rc[u].x >> 16 | ((rc[u].y >> 22) & 63) << 16;
                const uint32_t rel = m - mpos;
                if (int(rc[u].y) < 0 && rel >= off) sp[u] = mpos + D - off + rel % max(off, 1u);
            }
        }
        // the window ends where its output would leave the scratch span (literal runs between the matches count)
        const uint32_t bound0 = rdl(dest[0], 0);
        {
            uint32_t keep_total = avail; bool cut = false;
#pragma unroll
            for (int u = 0; u < K; u++) {
                const unsigned long long out = __ballot(uint32_t(lane) < nl[u] && dest[u] - bound0 >= uint32_t(kSpan));
                if (out && !cut) { cut = true; keep_total = 64u * u + uint32_t(__builtin_ctzll(out)); }
            }
            if (cut) {                                                  // rare: long literal runs inside the window
                avail = keep_total;                                     // >= 1: lane 0 of the first sub-step is always inside
                const uint32_t lu = (avail - 1) >> 6, ll = (avail - 1) & 63;
                ck = 0;
#pragma unroll
                for (int u = 0; u < K; u++) { nl[u] = avail > 64u * u ? min(avail - 64u * u, 64u) : 0u; if (uint32_t(u) == lu) ck = rdl(km[u], ll); }
            }
        }
        ckraw = ((ck - 1 + base) & 0x7FFF) + 1;                         // ck != 0: the window has at least one byte, its owner a mark
        pf.add(4, tc); tc = pf.now();
        // ---- execute.  Sources inside the window: a match byte another lane of the window produces travels through the
        // scratch; anything else (literals, older output) is in memory already - records are admitted only once their
        // literals are.  The marks of the window are consumed: cleared.
#pragma unroll
        for (int i = 0; i < kSpan * 2 / 1024; i++) reinterpret_cast<uint4*>(S->scr)[lane + 64 * i] = make_uint4(0, 0, 0, 0);
        LDS_ORDER();
        bool lv[K];
#pragma unroll
        for (int u = 0; u < K; u++) {
            lv[u] = uint32_t(lane) < nl[u];
            if (lv[u]) { S->scr[dest[u] - bound0] = 0x200; S->own[(g + 64u * u + uint32_t(lane)) & (kOwn - 1)] = 0; }
        }
        LDS_ORDER();
        uint32_t t[K], val[K];
#pragma unroll
        for (int u = 0; u < K; u++) t[u] = S->scr[(lv[u] && sp[u] >= bound0) ? sp[u] - bound0 : uint32_t(kSpan)];     // scr[kSpan] stays 0
        bool now[K]; unsigned long long pend[K];
#pragma unroll
        for (int u = 0; u < K; u++) { now[u] = lv[u] && t[u] == 0; pend[u] = __ballot(lv[u] && t[u] != 0); }
        pf.add(5, tc); tc = pf.now();
#pragma unroll
        for (int u = 0; u < K; u++) val[u] = dst[now[u] ? sp[u] : 0u];
#pragma unroll
        for (int u = 0; u < K; u++) if (now[u]) { dst[dest[u]] = uint8_t(val[u]); S->scr[dest[u] - bound0] = uint16_t(0x100u | val[u]); }
        pf.add(6, tc); tc = pf.now();
        for (uint32_t rounds = 0;; rounds++) {
            unsigned long long any = 0;
#pragma unroll
            for (int u = 0; u < K; u++) any |= pend[u];
            if (!any) break;
            if (rounds > 64u * K) { stv(&S->failed, 1); return; }       // every round completes a lane: more means broken marks
            pf.count(3);
            LDS_ORDER();
#pragma unroll
            for (int u = 0; u < K; u++) {
                if (!pend[u]) continue;                                 // (uniform) most rounds concern one or two sub-steps
                const uint32_t tt = S->scr[((pend[u] >> lane) & 1) ? sp[u] - bound0 : uint32_t(kSpan)];
                const bool got = (tt & 0x100) != 0;
                if (got) { dst[dest[u]] = uint8_t(tt); S->scr[dest[u] - bound0] = uint16_t(tt); }
                pend[u] &= ~__ballot(got);
            }
        }
        g += avail;
        pf.add(0, tc);
    }
}

__global__ __launch_bounds__(256, 8)
void lz4_decode_rows_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                            fourmc_block* blocks, uint32_t nblocks, int container_mode, unsigned long long* prof)
{
    __shared__ __attribute__((aligned(16))) Shared S;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (container_mode && blk.src_len == blk.dst_cap) {                 // stored block (native/4mc.c:635-642)
        if (wave == 0) { wave_copy(dst, src, int(blk.src_len), lane); if (lane == 0) blocks[b].result = int(blk.src_len); }
        return;
    }
    if (blk.src_len < 8 || blk.src_len > lz4par::kSrcMax || blk.dst_cap < 64 || blk.dst_cap > lz4par::kDstMax) {
        if (threadIdx.x == 0) blocks[b].result = kRetry;
        return;
    }
    for (uint32_t i = threadIdx.x; i < uint32_t(kOwn) / 2; i += 256) reinterpret_cast<uint32_t*>(S.own)[i] = 0;
    if (threadIdx.x < uint32_t(kNR)) S.res[threadIdx.x][0] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < uint32_t(kNQ)) S.pub[threadIdx.x] = make_uint2(0, 0);
    if (threadIdx.x == 0) {
        S.scr[kSpan] = 0; S.pw = make_uint2(0, 0); S.cc = make_uint2(0, 0); S.post_rows = 0; S.total_q = 0xFFFFFFFFu; S.failed = 0; S.end_value = kRetry;
    }
    __syncthreads();
    const int csize = int(blk.src_len), cap = int(blk.dst_cap);
    Prof pf;
    const unsigned long long t_role = pf.now();
    cgbyte* gsrc = (cgbyte*)src; gbyte* gdst = (gbyte*)dst;
    if (wave == 0) role_pre(&S, pf, gsrc, csize, lane);
    else if (wave == 1) role_walk(&S, pf, gsrc, gdst, csize, cap, lane);
    else if (wave == 2) role_post(&S, pf, gsrc, gdst, csize, lane);
    else role_copy<K1R_STEPS>(&S, pf, gdst, lane);
    pf.add(7, t_role);
#ifdef K1R_PROF
    if (prof && lane == 0) for (int i = 0; i < 8; i++) prof[(size_t(b) * 4 + wave) * 8 + i] = pf.t[i];
#endif
    __syncthreads();
    if (threadIdx.x == 0) blocks[b].result = S.failed ? kRetry : S.end_value;
}


// ================================================================================================ K1w: one lane per sequence
// The third shape of the LZ4 fast path (FOURMC_DECODE=lanes).  The row pipeline and the wave trio move one byte per lane and
// step, which costs 3.4 - 4.7 wave instructions per output byte; here a lane owns a whole SEQUENCE and moves it in pieces of
// 16 / 8 / 4 / 2 / 1 bytes, so that 64 sequences (some 600 bytes of text) cost about 200 instructions.  Two waves per block:
//   WALK  the serial part and nothing else: per window of 64 stream bytes every lane decodes the token that would start at its
//         byte (where the next one starts), a v_readlane chain marks the true tokens, and their stream POSITIONS go into a queue
//         in LDS.  Tokens the window rules do not cover (length continuations beyond one byte, long literal runs, the end of the
//         block) are decoded one at a time under the strict rules of the reference's safe loop (lz4.c:2120-2330) and queued with
//         their lengths.  The walk never needs an output position.
//   EXEC  takes up to 64 queued tokens: every lane reads its token and offset from the stream, a prefix sum of the sizes gives
//         each sequence its place, literals go out first (stream -> output), then the matches in passes: a match is copied once
//         everything in front of the first unfinished match of the batch covers its source (one pass for most batches of text).
//         Output-side rules (room behind a sequence, offsets) are checked here.  General tokens are executed by the whole wave.
// Irregular input of any kind ends in kRetry: the exact walker of lz4_decode.hip decides.
#ifndef K1WX_HOPS
#define K1WX_HOPS 4                  // v_readlane hops of the walk between two looks at its end (a window holds 21 tokens at most)
#define K1WX_ROUNDS 6
#endif
#ifndef K1WX_NAP_ROOM
#define K1WX_NAP_ROOM 16
#endif
#ifndef K1WX_NAP_PLAN
#define K1WX_NAP_PLAN 4
#endif
#ifndef K1WX_NAP_WALK
#define K1WX_NAP_WALK 32
#endif
constexpr int kTQ = 2048;                // token queue (entries)
constexpr int kLaneMatchMax = 80;        // longer matches / literal runs are general tokens (executed by the whole wave)
constexpr int kLRing = 4096, kLChunk = 1024, kLAhead = 2048;   // WALK's stream ring: bytes, refill granule (64 lanes x 16 B), staged ahead of the cursor
struct LShared {
    uint8_t  ring[kLRing];               // the compressed stream around WALK's cursor (aligned 16-byte granules, one KiB ahead of use)
    uint32_t tq[kTQ];                    // stream position of a token; a general token takes four entries: 1 << 31 | literal start, literals, offset, match length (0: last literals)
    uint32_t head, tail, total, failed;  // WALK: entries [0, head) written, their number once the block ends; EXEC: entries [0, tail) consumed
    int      end_value;
};

template <class SH, class P, class F> __device__ __forceinline__ bool lwait(SH* S, Prof& pf, int nap, P poll, F cond)
{
    if (cond()) return true;
    poll();
    if (cond()) { LDS_ORDER(); return true; }
    const unsigned long long t0 = pf.now();
    for (uint32_t spins = 0;;) {
        if ((spins & 7) == 0 && rfl(ldv(&S->failed))) return false;
        if (nap <= 2) __builtin_amdgcn_s_sleep(2); else if (nap <= 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
        poll();
        if (cond()) break;
        if (++spins > kSpinLimit) { stv(&S->failed, 1); return false; }
    }
    pf.add(0, t0);
    LDS_ORDER();
    return true;
}

template <class SH, int TQ, int RING, int AHEAD, int MLMAX>
__device__ void lanes_walk(SH* S, Prof& pf, cgbyte* src, const int csize, const int lane)
{
    constexpr int kTQ = TQ, kLRing = RING, kLAhead = AHEAD;

    const int iend = csize;
    Win win; win.src = src; win.csize = csize; win.lane = lane; win.la = 0; win.la_pos = -(1 << 30);
    int ip = 0;
    uint32_t head = 0, tail_seen = 0;
    // aligned 16-byte granules are safe to read whenever they hold at least one stream byte
    const int delta = int(uintptr_t(src) & 15), qend = delta + csize;
    cgbyte* const abase = src - delta;
    auto fetch = [&](int q) -> u32x4 { const int g = q + 16 * lane; u32x4 v = {0, 0, 0, 0}; if (g < qend) v = *reinterpret_cast<__attribute__((address_space(1))) const u32x4*>(abase + g); return v; };
    int fill_hi = 0;
    u32x4 pend = fetch(0);
    auto fail = [&]() { if (lane == 0) stv(&S->failed, 1); };
    auto room = [&](uint32_t n) -> bool {
        return lwait(S, pf, K1WX_NAP_WALK, [&] { tail_seen = rfl(ldv(&S->tail)); }, [&] { return head + n <= tail_seen + uint32_t(kTQ); });
    };
    for (;;) {
        ip = int(rfl(uint32_t(ip))); head = rfl(head); tail_seen = rfl(tail_seen); fill_hi = int(rfl(uint32_t(fill_hi)));
        if (ip + 64 + 16 <= iend) {
            // ---- a window: lane j = the token that would start at stream byte ip + j (four bytes per lane, from the ring: the
            // stream is staged a KiB ahead, a window costs LDS reads instead of a trip to memory in the middle of the chain)
            {
                const int q = ip + delta;
                if (q >= fill_hi + kLChunk) { fill_hi = q & ~(kLChunk - 1); pend = fetch(fill_hi); }
                while (fill_hi < q + kLAhead && fill_hi < qend) {
                    *reinterpret_cast<u32x4*>(S->ring + ((fill_hi + 16 * lane) & (kLRing - 1))) = pend;
                    fill_hi += kLChunk;
                    pend = fetch(fill_hi);
                }
            }
            const int q0 = ip + delta + lane;
            LDS_ORDER();
            const uint32_t w = uint32_t(S->ring[q0 & (kLRing - 1)]) | uint32_t(S->ring[(q0 + 1) & (kLRing - 1)]) << 8 |
                               uint32_t(S->ring[(q0 + 2) & (kLRing - 1)]) << 16 | uint32_t(S->ring[(q0 + 3) & (kLRing - 1)]) << 24;
            const uint32_t b = w & 0xff, b1 = (w >> 8) & 0xff, L0 = b >> 4, M0 = b & 15;
            const bool lext = L0 == 15, mext = M0 == 15;
            const uint32_t L = L0 + (lext ? b1 : 0u);
            const uint32_t offpos = uint32_t(lane) + 1 + (lext ? 1u : 0u) + L;
            const uint32_t wo = uint32_t(__shfl(int(w), int(offpos & 63)));
            const uint32_t e1 = (wo >> 16) & 0xff;
            const uint32_t ml = M0 + 4 + (mext ? e1 : 0u);
            const uint32_t nxt = offpos + 2 + (mext ? 1u : 0u);
            const bool ok = !(lext && b1 == 255) && !(mext && e1 == 255) && nxt <= 64 && ml <= uint32_t(MLMAX);
            const uint32_t jump = ok ? nxt : 128u + uint32_t(lane);     // 64: the window ends behind this token; >= 128: no window token
            const uint32_t hop = lane == 63 ? 63u : min(jump, 63u);
            unsigned long long tokmask = 0;
            uint32_t pos = 0;
            for (int round = 0; round < K1WX_ROUNDS && pos != 63; round++) {
#pragma unroll
                for (int i = 0; i < K1WX_HOPS; i++) {
                    asm volatile("s_bitset1_b64 %0, %1" : "+s"(tokmask) : "s"(pos));
                    pos = rdl(hop, pos);
                }
            }
            tokmask &= ~(1ull << 63);
            {   // where the chain left the window: behind its last token (64), or at a byte that is no window token
                const uint32_t last = 63u - uint32_t(__builtin_clzll(tokmask | 1ull));
                const uint32_t j = rdl(jump, last);
                if (j >= 128) { tokmask &= ~(1ull << last); pos = last; } else pos = j;
            }
            if (tokmask) {
                const uint32_t cnt = uint32_t(__builtin_popcountll(tokmask));
                pf.count(1); pf.count(2, cnt);
                if (!room(cnt)) return;
                if ((tokmask >> lane) & 1) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(tokmask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(tokmask), 0u));
                    S->tq[(head + rank) & (kTQ - 1)] = uint32_t(ip) + uint32_t(lane);
                }
                head += cnt;
                LDS_ORDER();
                if (lane == 0) stv(&S->head, head);
                ip += int(pos);
                continue;
            }
        }
        // ---- one sequence under the strict rules (stream side; EXEC checks the output side)
        pf.count(3);
        if (ip >= iend) { fail(); return; }
        const uint32_t token = win.get(ip); ip++;
        int lit = int(token >> 4), mlen = int(token & 15);
        if (lit == 15) { if (!more_len(win, ip, iend - 15, true, lit)) { fail(); return; } }
        const bool last = ip + lit > iend - 8;
        if (last && ip + lit != iend) { fail(); return; }
        const int lit_ip = ip;
        int off = 0;
        if (!last) {
            ip += lit;
            off = int(win.get(ip)) | (int(win.get(ip + 1)) << 8);
            ip += 2;
            if (mlen == 15) { if (!more_len(win, ip, iend - 4, false, mlen)) { fail(); return; } }
            mlen += 4;
            if (off == 0) { fail(); return; }
        } else mlen = 0;
        if (!room(4)) return;
        if (lane < 4) S->tq[(head + uint32_t(lane)) & (kTQ - 1)] = lane == 0 ? (0x80000000u | uint32_t(lit_ip)) : (lane == 1 ? uint32_t(lit) : (lane == 2 ? uint32_t(off) : uint32_t(mlen)));
        head += 4;
        LDS_ORDER();
        if (lane == 0) stv(&S->head, head);
        if (last) break;
    }
    LDS_ORDER();
    if (lane == 0) stv(&S->total, head);
}

// dst[d .. d+n) = dst[d-off ..], byte-serial semantics, by the whole wave: 16 bytes per lane where the distance allows
__device__ __attribute__((noinline)) void lanes_big_match(gbyte* dst, int d, int off, int n, int lane)
{
    if (off < 16) {
        // the output is periodic: the first 64 bytes straight from the period, then the distance is the largest multiple of
        // the period that fits 64 (>= 50)
        const int r = lane % off;
        if (lane < min(n, 64)) { const uint8_t v = dst[d - off + r]; dst[d + lane] = v; }
        if (n <= 64) return;
        const int D = (64 / off) * off;
        d += 64; n -= 64; off = D;
    }
    // a step moves `step` bytes, a multiple of 16 not larger than the distance: no step reads what it writes
    const int step = min(off & ~15, 1024);
    int k = 0;
    for (; k + step <= n; k += step) {
        if (16 * lane < step) { const u32x4 v = ld16g(dst + d - off + k + 16 * lane); *reinterpret_cast<__attribute__((address_space(1))) u32x4_u*>(dst + d + k + 16 * lane) = v; }
    }
    while (k < n) {                                                     // what is left (less than a step): bytes
        const int lim = min(n - k, min(off, 64));
        if (lane < lim) { const uint8_t v = dst[d - off + k + lane]; dst[d + k + lane] = v; }
        k += lim;
    }
}

__device__ void lanes_exec(LShared* S, Prof& pf, cgbyte* src, gbyte* dst, const int cap, const int lane)
{
    const int oend = cap;
    uint32_t tail = 0, head_seen = 0, total = 0xFFFFFFFFu;
    int op = 0;
    auto fail = [&]() { if (lane == 0) stv(&S->failed, 1); };
    for (;;) {
        tail = rfl(tail); head_seen = rfl(head_seen); op = int(rfl(uint32_t(op)));
        if (head_seen == tail) {
            auto poll = [&] { total = rfl(ldv(&S->total)); LDS_ORDER(); head_seen = rfl(ldv(&S->head)); };
            if (!lwait(S, pf, 2, poll, [&] { return head_seen != tail || total == tail; })) return;
            if (head_seen == tail) break;                               // total == tail: every token executed
        }
        const uint32_t n = min(head_seen - tail, 64u);
        const uint32_t e = uint32_t(lane) < n ? S->tq[(tail + uint32_t(lane)) & (kTQ - 1)] : 0u;
        const unsigned long long gm = __ballot(uint32_t(lane) < n && (e >> 31));
        const uint32_t g = gm ? uint32_t(__builtin_ctzll(gm)) : n;
        if (g == 0) {
            // ---- a general token, by the whole wave
            pf.count(5);
            const int lit_ip = int(rdl(e, 0) & 0x7FFFFFFFu), lit = int(rdl(e, 1)), off = int(rdl(e, 2)), mlen = int(rdl(e, 3));
            tail += 4;
            LDS_ORDER();
            if (lane == 0) stv(&S->tail, tail);
            if (mlen == 0) {                                            // the block's last sequence: literals only
                if (op + lit > oend) { fail(); return; }
                lean_copy(dst + op, src + lit_ip, lit, lane);
                op += lit;
                continue;
            }
            if (op + lit > oend - 12) { fail(); return; }
            const int op2 = op + lit;
            if (off > op2 || op2 + mlen > oend - 5) { fail(); return; }
            lean_copy(dst + op, src + lit_ip, lit, lane);
            lanes_big_match(dst, op2, off, mlen, lane);
            op = op2 + mlen;
            continue;
        }
        // ---- up to 64 window tokens, one per lane
        pf.count(1); pf.count(2, g);
        const bool act = uint32_t(lane) < g;
        const uint32_t pos = act ? e : 0u;
        const uint32_t w = ld4u(src + pos);
        const uint32_t b = w & 0xff, b1 = (w >> 8) & 0xff, L0 = b >> 4, M0 = b & 15;
        const uint32_t lextn = L0 == 15 ? 1u : 0u;
        const uint32_t L = act ? L0 + (lextn ? b1 : 0u) : 0u;
        const uint32_t wo = ld4u(src + pos + 1 + lextn + L);
        const uint32_t off = wo & 0xffff;
        const uint32_t ml = act ? M0 + 4 + (M0 == 15 ? (wo >> 16) & 0xff : 0u) : 0u;
        tail += g;
        LDS_ORDER();
        if (lane == 0) stv(&S->tail, tail);
        const uint32_t sz = L + ml;
        const uint32_t incl = scan_add(sz);
        const uint32_t ostart = uint32_t(op) + incl - sz, mdest = ostart + L;
        const uint32_t T = rdl(incl, 63);
        if (__ballot(act && (off == 0 || off > mdest || ostart + L > uint32_t(oend - 12) || mdest + ml > uint32_t(oend - 5))) || oend < 12) { fail(); return; }
        // pieces of a run of n bytes (n < 32 after the 16-byte loop): all loads, then all stores - one round trip
        auto copy_run = [&](const bool on, cgbyte* from, gbyte* to, uint32_t nb) {
            uint32_t k = 0;
            while (__ballot(on && nb - k >= 32)) { if (on && nb - k >= 32) { const u32x4 v = ld16g(from + k); *reinterpret_cast<__attribute__((address_space(1))) u32x4_u*>(to + k) = v; k += 16; } }
            const uint32_t r = on ? nb - k : 0u;                         // < 32
            typedef uint64_t u64_u __attribute__((aligned(1)));
            typedef uint16_t u16_u __attribute__((aligned(1)));
            u32x4 v16 = {0, 0, 0, 0}; uint64_t v8 = 0; uint32_t v4 = 0, v2 = 0, v1 = 0;
            const uint32_t o16 = k, o8 = k + (r & 16), o4 = o8 + (r & 8), o2 = o4 + (r & 4), o1 = o2 + (r & 2);
            if (r & 16) v16 = ld16g(from + o16);
            if (r & 8) v8 = *reinterpret_cast<__attribute__((address_space(1))) const u64_u*>(from + o8);
            if (r & 4) v4 = ld4u(from + o4);
            if (r & 2) v2 = *reinterpret_cast<__attribute__((address_space(1))) const u16_u*>(from + o2);
            if (r & 1) v1 = from[o1];
            if (r & 16) *reinterpret_cast<__attribute__((address_space(1))) u32x4_u*>(to + o16) = v16;
            if (r & 8) *reinterpret_cast<__attribute__((address_space(1))) u64_u*>(to + o8) = v8;
            if (r & 4) *reinterpret_cast<__attribute__((address_space(1))) u32_u*>(to + o4) = v4;
            if (r & 2) *reinterpret_cast<__attribute__((address_space(1))) u16_u*>(to + o2) = uint16_t(v2);
            if (r & 1) to[o1] = uint8_t(v1);
        };
        copy_run(act && L != 0, src + pos + 1 + lextn, dst + ostart, L);
        // ---- matches.  Which earlier matches of the batch a source touches is fixed: destinations ascend with the lane, so they are
        // the lanes [u0, u1) found by two binary searches over the wave (ends <= source start, starts < source end); a match goes
        // in the first pass in which none of those is unfinished.  What a match reads of ITSELF (offset < length) is no dependence
        // on others: such a lane copies piece by piece, each piece behind its own stores (pieces of 16 or 8 bytes; below 8 the
        // whole wave expands the period).
        const bool overlap = act && off < ml;
        const uint32_t s0 = mdest - off, s1 = overlap ? mdest : s0 + ml;
        const uint32_t mend = act ? mdest + ml : 0xFFFFFFFFu, mdst = act ? mdest : 0xFFFFFFFFu;
        unsigned long long dep = 0;
        if (__ballot(act && s1 > uint32_t(op))) {                        // (uniform) some source reaches into this batch
            uint32_t u0 = 0, u1 = 0;
#pragma unroll
            for (int step = 32; step; step >>= 1) {
                const uint32_t e0 = bperm((u0 + uint32_t(step) - 1) << 2, mend), e1 = bperm((u1 + uint32_t(step) - 1) << 2, mdst);
                if (e0 <= s0) u0 += uint32_t(step);
                if (e1 < s1) u1 += uint32_t(step);
            }
            u0 = min(u0, 63u); u1 = min(u1, 64u);
            const unsigned long long below1 = u1 >= 64 ? ~0ull : ~(~0ull << u1), below0 = ~(~0ull << u0);
            dep = act && u1 > u0 ? below1 & ~below0 : 0ull;
        }
        unsigned long long done = ~__ballot(act);
        const unsigned long long tpass = pf.now();
        for (uint32_t pass = 0; ~done; pass++) {
            if (pass > 64) { fail(); return; }
            pf.count(3);
            const bool go = !((done >> lane) & 1) && (dep & ~done) == 0;
            const unsigned long long gomask = __ballot(go);
            copy_run(go && !overlap, (cgbyte*)(dst + s0), dst + mdest, ml);
            const unsigned long long ov = __ballot(go && overlap);
            if (ov) {
                pf.count(4, uint64_t(__builtin_popcountll(ov)));
                // offsets of 8 and more: in the lane, sequentially (a piece never reads what it writes itself)
                const bool seq = go && overlap && off >= 8;
                if (__ballot(seq)) {
                    typedef uint64_t u64_u __attribute__((aligned(1)));
                    uint32_t k = 0;
                    while (__ballot(seq && off >= 16 && ml - k >= 16)) { if (seq && off >= 16 && ml - k >= 16) { const u32x4 v = ld16g((cgbyte*)(dst + s0 + k)); *reinterpret_cast<__attribute__((address_space(1))) u32x4_u*>(dst + mdest + k) = v; k += 16; } }
                    while (__ballot(seq && ml - k >= 8)) { if (seq && ml - k >= 8) { const uint64_t v = *reinterpret_cast<__attribute__((address_space(1))) const u64_u*>(dst + s0 + k); *reinterpret_cast<__attribute__((address_space(1))) u64_u*>(dst + mdest + k) = v; k += 8; } }
                    if (seq && ((ml - k) & 4)) { const uint32_t v = ld4u((cgbyte*)(dst + s0 + k)); *reinterpret_cast<__attribute__((address_space(1))) u32_u*>(dst + mdest + k) = v; k += 4; }
                    if (seq && ((ml - k) & 2)) { typedef uint16_t u16_u __attribute__((aligned(1))); const uint16_t v = *reinterpret_cast<__attribute__((address_space(1))) const u16_u*>(dst + s0 + k); *reinterpret_cast<__attribute__((address_space(1))) u16_u*>(dst + mdest + k) = v; k += 2; }
                    if (seq && ((ml - k) & 1)) { const uint8_t v = dst[s0 + k]; dst[mdest + k] = v; }
                }
                for (unsigned long long m = __ballot(go && overlap && off < 8); m; m &= m - 1) {   // short periods: by the whole wave, one at a time
                    const uint32_t f = uint32_t(__builtin_ctzll(m));
                    lanes_big_match(dst, int(rdl(mdest, f)), int(rdl(off, f)), int(rdl(ml, f)), lane);
                }
            }
            done |= gomask;
        }
        pf.add(6, tpass);
        op += int(T);
    }
    if (lane == 0) S->end_value = op;
}

__global__ __launch_bounds__(128)
void lz4_decode_lanes_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                             fourmc_block* blocks, uint32_t nblocks, int container_mode, unsigned long long* prof)
{
    __shared__ __attribute__((aligned(16))) LShared S;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (container_mode && blk.src_len == blk.dst_cap) {                 // stored block (native/4mc.c:635-642)
        if (wave == 0) { wave_copy(dst, src, int(blk.src_len), lane); if (lane == 0) blocks[b].result = int(blk.src_len); }
        return;
    }
    if (blk.src_len < 8 || blk.src_len > lz4par::kSrcMax || blk.dst_cap < 64 || blk.dst_cap > lz4par::kDstMax) {
        if (threadIdx.x == 0) blocks[b].result = kRetry;
        return;
    }
    if (threadIdx.x == 0) { S.head = 0; S.tail = 0; S.total = 0xFFFFFFFFu; S.failed = 0; S.end_value = kRetry; }
    __syncthreads();
    Prof pf;
    const unsigned long long t_role = pf.now();
    if (wave == 0) lanes_walk<LShared, kTQ, kLRing, kLAhead, kLaneMatchMax>(&S, pf, (cgbyte*)src, int(blk.src_len), lane);
    else lanes_exec(&S, pf, (cgbyte*)src, (gbyte*)dst, int(blk.dst_cap), lane);
    pf.add(7, t_role);
#ifdef K1R_PROF
    if (prof && lane == 0) for (int i = 0; i < 8; i++) prof[(size_t(b) * 4 + wave) * 8 + i] = pf.t[i];
#endif
    __syncthreads();
    if (threadIdx.x == 0) blocks[b].result = S.failed ? kRetry : S.end_value;
}


// ================================================================================================ K1wx: the walk in front of the window copier
// FOURMC_DECODE=wx.  Four waves per block: WALK as in K1w (token positions into a queue; the serial part and nothing else), then the
// execute pipeline of the 4mz decoder (zstd_exec.inc) fed from that queue:
//   SL    up to 64 queued tokens, one per lane: token and offset from the stream, prefix sums place every sequence in the output,
//         in the group's literal bytes and in MATCH SPACE; one record per sequence for PLAN, then the group's literals - one byte
//         per lane, owner by max-scan over a byte map of literal starts, read from where the token's literals lie in the stream.
//         A group reaches PLAN only after its literal stores have completed.  General tokens: literals by the whole wave, the match
//         as a group of one.
//   PLAN / EXEC   the window copier (window_copier.inc: 256 match-space positions per window, stores one window late).
constexpr int kYQ = 4, kYOwn = 1024, kYSpan = 512, kYSteps = 4, kYTQ = 512, kYRing = 2048, kYAhead = 1024;
constexpr uint32_t kYLitCap = 1023, kYMatCap = 511;
struct YShared {
    uint16_t own[kYOwn];                 // start marks in match space: ((q & 511) << 6 | lane) + 1
    uint2    rec[kYQ][64];               // {offset | mpos[9:0] << 22, D | mpos[18:10] << 22 | overlap << 31}
    uint2    plan[2][kYSteps * 64];      // PLAN -> EXEC
    uint8_t  ring[kYRing];               // WALK: the compressed stream around its cursor
    uint32_t tq[kYTQ];                   // WALK -> SL: token positions (general tokens: four entries, see K1w)
    uint16_t scr[2][kYSpan + 8];         // EXEC: two windows' worth of produced bytes
    uint8_t  lmark[1024 + 64];           // SL: lane + 1 at the literal-stream start of each sequence of the group
    uint2    pub[kYQ];                   // SL -> PLAN: {q + 1 once the group's literals are in memory, end of the group in match space}
    uint2    phdr[2];
    uint2    cc;                         // PLAN: x groups consumed, y match-space bytes consumed
    uint32_t head, tail, total;          // the token queue
    uint32_t plan_ready, plan_total, exec_done;
    uint32_t total_q, failed;
    int      end_value;
};
static_assert(sizeof(YShared) <= 20480, "eight blocks per CU need <= 20 KiB of LDS each");

template <class P, class F> __device__ __forceinline__ bool ywait(YShared* S, int nap, P poll, F cond, unsigned long long& waited)
{
    if (cond()) return true;
    poll();
    if (cond()) { LDS_ORDER(); return true; }
    for (uint32_t spins = 0;;) {
        if ((spins & 7) == 0 && rfl(ldv(&S->failed))) return false;
        if (nap <= 2) __builtin_amdgcn_s_sleep(2); else if (nap <= 4) __builtin_amdgcn_s_sleep(4); else if (nap <= 8) __builtin_amdgcn_s_sleep(8); else if (nap <= 16) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(32);
        poll();
        waited++;
        if (cond()) break;
        if (++spins > kSpinLimit) { stv(&S->failed, 1); return false; }
    }
    LDS_ORDER();
    return true;
}

#include "window_copier.inc"

__device__ void y_sl(YShared* S, cgbyte* src, gbyte* dst, const int cap, const int lane, unsigned long long& waited)
{
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const int oend = cap;
    uint32_t tail = 0, head_seen = 0, total = kNone;
    uint32_t q = 0, mb = 0, cq_seen = 0, cg_seen = 0, pend_q = kNone, pend_m = 0;
    int op = 0;
    auto fail = [&]() { if (lane == 0) stv(&S->failed, 1); };
    auto publish = [&] {       // the pending group's stores have completed (the caller waited): show it to PLAN
        if (pend_q != kNone) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) S->pub[pend_q & (kYQ - 1)] = make_uint2(pend_q + 1, pend_m);
            pend_q = kNone;
        }
    };
    auto flush_pending = [&] { if (pend_q != kNone) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); publish(); } };
    auto room = [&] { return cq_seen + kYQ > q && mb <= cg_seen + uint32_t(kYOwn - 512); };
    auto wait_room = [&]() -> bool {      // a record slot is free and the group's start marks fit the ring
        if (room()) return true;
        flush_pending();                   // PLAN may be waiting for what is held back here
        return ywait(S, K1WX_NAP_ROOM, [&] { const uint2 c = ldv2(&S->cc); cq_seen = c.x; cg_seen = c.y; }, room, waited);
    };
    auto mark_of = [&](uint32_t l) { return uint16_t(((q & 511) << 6 | l) + 1); };
    for (;;) {
        tail = rfl(tail); head_seen = rfl(head_seen); op = int(rfl(uint32_t(op))); q = rfl(q); mb = rfl(mb);
        cq_seen = rfl(cq_seen); cg_seen = rfl(cg_seen); pend_q = rfl(pend_q); pend_m = rfl(pend_m);
        if (head_seen == tail) {
            auto poll = [&] { total = rfl(ldv(&S->total)); LDS_ORDER(); head_seen = rfl(ldv(&S->head)); };
            poll();
            if (head_seen == tail && total != tail) {
                flush_pending();
                if (!ywait(S, 2, poll, [&] { return head_seen != tail || total == tail; }, waited)) return;
            }
            if (head_seen == tail) break;                               // total == tail: every token taken
        }
        const uint32_t n = min(head_seen - tail, 64u);
        const uint32_t e = uint32_t(lane) < n ? S->tq[(tail + uint32_t(lane)) & (kYTQ - 1)] : 0u;
        const unsigned long long gm = __ballot(uint32_t(lane) < n && (e >> 31));
        const uint32_t g = gm ? uint32_t(__builtin_ctzll(gm)) : n;
        if (g == 0) {
            // ---- a general token: literals by the whole wave, the match as a group of one
            const int lit_ip = int(rdl(e, 0) & 0x7FFFFFFFu), lit = int(rdl(e, 1)), off = int(rdl(e, 2)), mlen = int(rdl(e, 3));
            tail += 4;
            LDS_ORDER();
            if (lane == 0) stv(&S->tail, tail);
            if (mlen == 0) {                                            // the block's last sequence: literals only
                if (op + lit > oend) { fail(); return; }
                lean_copy(dst + op, src + lit_ip, lit, lane);
                op += lit;
                continue;
            }
            if (op + lit > oend - 12) { fail(); return; }
            const int op2 = op + lit;
            if (off > op2 || op2 + mlen > oend - 5) { fail(); return; }
            if (!wait_room()) return;
            flush_pending();                                            // (in front of this group's stores, as below)
            lean_copy(dst + op, src + lit_ip, lit, lane);
            if (lane == 0) {
                S->own[mb & (kYOwn - 1)] = mark_of(0);
                S->rec[q & (kYQ - 1)][0] = make_uint2(uint32_t(off) | (mb & 0x3FF) << 22, uint32_t(op2 - int(mb)) | ((mb >> 10) & 0x1FF) << 22 | (off < mlen ? 1u : 0u) << 31);
            }
            pend_q = q; pend_m = mb + uint32_t(mlen);
            q++; mb += uint32_t(mlen); op = op2 + mlen;
            continue;
        }
        // ---- up to 64 window tokens, one per lane; the group ends where its literals or the matches in front of a sequence pass the caps
        const bool in = uint32_t(lane) < g;
        const uint32_t pos = in ? e : 0u;
        const uint32_t w = ld4u(src + pos);
        const uint32_t b = w & 0xff, b1 = (w >> 8) & 0xff, L0 = b >> 4, M0 = b & 15;
        const uint32_t lextn = L0 == 15 ? 1u : 0u;
        const uint32_t L_all = in ? L0 + (lextn ? b1 : 0u) : 0u;
        const uint32_t litpos = pos + 1 + lextn;
        const uint32_t wo = ld4u(src + litpos + L_all);
        const uint32_t off = wo & 0xffff;
        const uint32_t ml_all = in ? M0 + 4 + (M0 == 15 ? (wo >> 16) & 0xff : 0u) : 0u;
        const uint32_t lli0 = scan_add(L_all), mli0 = scan_add(ml_all);
        const unsigned long long okm = __ballot(in && lli0 <= kYLitCap && mli0 - ml_all <= kYMatCap);
        const uint32_t cnt = ~okm ? uint32_t(__builtin_ctzll(~okm)) : 64u;          // >= 1: a window token has at most 62 literals
        const bool act = uint32_t(lane) < cnt;
        const uint32_t L = act ? L_all : 0u, ml = act ? ml_all : 0u;
        tail += cnt;
        LDS_ORDER();
        if (lane == 0) stv(&S->tail, tail);
        const uint32_t Lsum = rdl(lli0, cnt - 1), Msum = rdl(mli0, cnt - 1);
        const uint32_t lstart = lli0 - L_all, mexcl = mli0 - ml_all;
        const uint32_t ostart = uint32_t(op) + lstart + mexcl, mdest = ostart + L;
        if (oend < 12 || __ballot(act && (off == 0 || off > mdest || ostart + L > uint32_t(oend - 12) || mdest + ml > uint32_t(oend - 5)))) { fail(); return; }
        if (!wait_room()) return;
        if (act) {
            const uint32_t mpos = mb + mexcl;
            S->own[mpos & (kYOwn - 1)] = mark_of(uint32_t(lane));
            S->rec[q & (kYQ - 1)][lane] = make_uint2(off | (mpos & 0x3FF) << 22, (mdest - mpos) | ((mpos >> 10) & 0x1FF) << 22 | (off < ml ? 1u : 0u) << 31);
        }
        // the previous group's literal stores were issued an iteration ago: complete them (normally no wait) and publish it BEFORE
        // this group's stores go out
        flush_pending();
        if (Lsum) {
            const uint32_t lr = L | lstart << 10 | mexcl << 20;
            if (act && L) S->lmark[lstart] = uint8_t(lane + 1);
            LDS_ORDER();
            uint32_t carry = 0;
            for (uint32_t j0 = 0; j0 < Lsum; j0 += 64u * kYSteps) {
                uint32_t m[kYSteps], lw[kYSteps], lp[kYSteps], v[kYSteps];
#pragma unroll
                for (int u = 0; u < kYSteps; u++) m[u] = S->lmark[min(j0 + 64u * u + uint32_t(lane), 1024u + 63u)];
#pragma unroll
                for (int u = 0; u < kYSteps; u++) { m[u] = max(scan_max(m[u]), carry); carry = rdl(m[u], 63); }
#pragma unroll
                for (int u = 0; u < kYSteps; u++) { const uint32_t a4 = ((m[u] - 1) & 63) << 2; lw[u] = bperm(a4, lr); lp[u] = bperm(a4, litpos); }
#pragma unroll
                for (int u = 0; u < kYSteps; u++) {
                    const uint32_t j = j0 + 64u * u + uint32_t(lane);
                    v[u] = src[j < Lsum ? lp[u] + (j - ((lw[u] >> 10) & 1023)) : 0u];
                }
#pragma unroll
                for (int u = 0; u < kYSteps; u++) { const uint32_t j = j0 + 64u * u + uint32_t(lane); if (j < Lsum) dst[uint32_t(op) + j + (lw[u] >> 20)] = uint8_t(v[u]); }
            }
            LDS_ORDER();
            *reinterpret_cast<uint4*>(S->lmark + 16 * lane) = make_uint4(0, 0, 0, 0);
            LDS_ORDER();
        }
        pend_q = q; pend_m = mb + Msum;
        q++; op += int(Lsum + Msum); mb += Msum;
    }
    flush_pending();
    if (lane == 0) S->end_value = op;
    LDS_ORDER();
    if (lane == 0) stv(&S->total_q, q);
}

__global__ __launch_bounds__(256, 8)
void lz4_decode_wx_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base,
                          fourmc_block* blocks, uint32_t nblocks, int container_mode, unsigned long long* prof,
                          const uint32_t* pick, uint32_t want)
{
    __shared__ __attribute__((aligned(16))) YShared S;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    if (pick && *pick != want) return;                                  // a full launch of short streams is the wave trio's (lz4_decode.hip: lz4_pick_kernel)
    const fourmc_block blk = uniform_block(blocks[b]);
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (container_mode && blk.src_len == blk.dst_cap) {                 // stored block (native/4mc.c:635-642)
        if (wave == 0) { wave_copy(dst, src, int(blk.src_len), lane); if (lane == 0) blocks[b].result = int(blk.src_len); }
        return;
    }
    if (blk.src_len < 8 || blk.src_len > lz4par::kSrcMax || blk.dst_cap < 64 || blk.dst_cap > lz4par::kDstMax) {
        if (threadIdx.x == 0) blocks[b].result = kRetry;
        return;
    }
    for (uint32_t i = threadIdx.x; i < uint32_t(kYOwn) / 2; i += 256) reinterpret_cast<uint32_t*>(S.own)[i] = 0;
    for (uint32_t i = threadIdx.x; i < sizeof(S.lmark) / 4; i += 256) reinterpret_cast<uint32_t*>(S.lmark)[i] = 0;
    if (threadIdx.x < uint32_t(kYQ)) S.pub[threadIdx.x] = make_uint2(0, 0);
    if (threadIdx.x == 0) {
        S.scr[0][kYSpan] = 0; S.scr[1][kYSpan] = 0; S.cc = make_uint2(0, 0); S.head = 0; S.tail = 0; S.total = 0xFFFFFFFFu;
        S.plan_ready = 0; S.exec_done = 0; S.plan_total = 0xFFFFFFFFu; S.total_q = 0xFFFFFFFFu; S.failed = 0; S.end_value = kRetry;
    }
    __syncthreads();
    Prof pf;
    unsigned long long waited = 0;
    const unsigned long long t_role = pf.now();
    if (wave == 0) lanes_walk<YShared, kYTQ, kYRing, kYAhead, 1023>(&S, pf, (cgbyte*)src, int(blk.src_len), lane);
    else if (wave == 1) y_sl(&S, (cgbyte*)src, (gbyte*)dst, int(blk.dst_cap), lane, waited);
    else if (wave == 2) wc_plan<YShared, kYSteps, kYQ, kYOwn, kYSpan, K1WX_NAP_PLAN>(&S, lane, waited);
    else wc_exec<YShared, kYSteps, kYSpan, false>(&S, (wc_g8*)dst, lane, waited);
    pf.add(7, t_role); pf.count(6, waited);
#ifdef K1R_PROF
    if (prof && lane == 0) for (int i = 0; i < 8; i++) prof[(size_t(b) * 4 + wave) * 8 + i] = pf.t[i];
#endif
    __syncthreads();
    if (threadIdx.x == 0) blocks[b].result = S.failed ? kRetry : S.end_value;
}

} // namespace

#ifdef K1R_PROF
static unsigned long long* g_prof = nullptr; static uint32_t g_prof_blocks = 0;
#endif
#ifdef FOURMC_RESEARCH      // the row pipeline and the lane-per-sequence path: measured alternatives, side build only
extern "C" hipError_t fourmc_launch_lz4_rows(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                             uint32_t n, int container_mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    unsigned long long* prof = nullptr;
#ifdef K1R_PROF
    if (n > g_prof_blocks) { if (g_prof) (void)hipFree(g_prof); g_prof = nullptr; if (hipMalloc(&g_prof, size_t(n) * 32 * 8) == hipSuccess) g_prof_blocks = n; }
    prof = g_prof;
#endif
    hipLaunchKernelGGL(lz4_decode_rows_kernel, dim3(n), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, prof);
    return hipGetLastError();
}

#ifdef K1R_PROF
// profiling side build only: the counters of the last launch, [block][role: pre, walk, post, copy][8]
extern "C" int fourmc_gpu_debug_rows_prof(unsigned long long* host, uint32_t nblocks)
{
    if (!g_prof || nblocks > g_prof_blocks) return -1;
    return hipMemcpy(host, g_prof, size_t(nblocks) * 32 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

extern "C" hipError_t fourmc_launch_lz4_lanes(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                              uint32_t n, int container_mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    unsigned long long* prof = nullptr;
#ifdef K1R_PROF
    if (n > g_prof_blocks) { if (g_prof) (void)hipFree(g_prof); g_prof = nullptr; if (hipMalloc(&g_prof, size_t(n) * 32 * 8) == hipSuccess) g_prof_blocks = n; }
    prof = g_prof;
#endif
    hipLaunchKernelGGL(lz4_decode_lanes_kernel, dim3(n), dim3(128), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, prof);
    return hipGetLastError();
}

#endif

extern "C" hipError_t fourmc_launch_lz4_wx(const void* d_src, void* d_dst, fourmc_block* d_blocks,
                                           uint32_t n, int container_mode, hipStream_t stream, const uint32_t* pick, uint32_t want)
{
    if (n == 0) return hipSuccess;
    unsigned long long* prof = nullptr;
#ifdef K1R_PROF
    if (n > g_prof_blocks) { if (g_prof) (void)hipFree(g_prof); g_prof = nullptr; if (hipMalloc(&g_prof, size_t(n) * 32 * 8) == hipSuccess) g_prof_blocks = n; }
    prof = g_prof;
#endif
    hipLaunchKernelGGL(lz4_decode_wx_kernel, dim3(n), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n, container_mode, prof, pick, want);
    return hipGetLastError();
}
