// 4mc_amd/csrc/lz4_encode.hip — K2: batched LZ4 "fast" block encode on gfx950, BYTE-IDENTICAL to
// LZ4_compress_default of the reference's 64-bit little-endian build.
//
// Replaces native/4mc.c:301 (capacity n-1, stored fallback) and native/jniCompressor.c:91
// (capacity LZ4_compressBound) -> native/lz4/lz4.c:1435 -> :1416 -> :1346-1367 -> :910-1302.
//
// The reference parse is a serial greedy walk whose bytes depend on the exact order of hash
// table reads and overwrites (lz4.c:1059,1207,1247), on the probe stride schedule
// (`step = searchMatchNb++ >> 6`, :1017-1027) and on backward extension (:1080).  It is
// reproduced exactly, but 64 probes at a time:
//   * one wavefront owns one block; its hash table (4096 x u32, or 8192 x u16 for blocks
//     < 65547 B, lz4.c:1353) lives in LDS and is zeroed per block like LZ4_initStream (:1348);
//   * lane l speculatively executes probe k0+l of the current search (its position follows from
//     the closed form of the stride schedule), reads its candidate from the LDS table and tests
//     the 4-byte match; a ballot finds the FIRST lane (serial order) that hits, or that runs
//     into the end-of-block limit; only lanes before it commit their table writes;
//   * two probes of one batch that fall into the same table slot would see each other's write in
//     the serial order, so the batch is cut at the first lane that shares a (folded) slot with an
//     earlier lane - detected with an LDS atomic-min scoreboard; the cut lane simply becomes
//     lane 0 of the next batch.  False sharing of the scoreboard only shortens a batch;
//   * match extension (backwards and forwards) and literal / length emission are wave-wide
//     compares + ballots and wave-wide byte copies.
// Per block HBM traffic: n bytes read (+ candidate re-reads that hit L2/MALL), csize written.
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
    uint64_t pt_search = 0, pt_ext = 0, pt_match = 0, pt0 = __builtin_readcyclecounter(), pt1;
#define K2PH(acc) do { pt1 = __builtin_readcyclecounter(); acc += pt1 - pt0; pt0 = pt1; } while (0)
#else
#define K2PH(acc) do { } while (0)
#endif
    const bool limited = cap < n + n / 255 + 16;                       // lz4.c:1352
    uint32_t op = 0, anchor = 0;

    for (int i = lane; i < (1 << kHashLog); i += 64) tab32[i] = 0;     // LZ4_initStream
    for (int i = lane; i < kScore; i += 64) score[i] = 0xFFFFFFFFu;

    auto tab_get = [&](uint32_t h) -> uint32_t { return U32TAB ? tab32[h] : uint32_t(tab16[h]); };
    auto tab_put = [&](uint32_t h, uint32_t v) { if (U32TAB) tab32[h] = v; else tab16[h] = uint16_t(v); };

    if (n >= kMinLen) {
        const uint32_t un = uint32_t(n);
        const uint32_t lim = un - kMfLimit + 1;                        // mflimitPlusOne
        const uint32_t matchlimit = un - kLastLit;
        if (lane == 0) tab_put(hash_at<U32TAB>(src), 0);
        uint32_t sp = 1;                                               // first probe of the search
        // Cursor window: lane l keeps the 16 bytes [wsp+l-4, wsp+l+12) of the input.  Loaded once per ~50
        // bytes of progress, it feeds (through lane permutes / readlane, no memory round trip) the probe
        // words of the following searches, the extension bytes around a hit, short literal runs and the
        // ip-2 / ip refill after a match.
        Q16 W = {0, 0, 0, 0}; uint32_t wsp = 0; bool wvalid = false;
        // After a match the reference refills ip-2, re-tests ip at once (lz4.c:1207-1259) and only then starts the
        // next search at ip+1.  Here that re-test is lane 0 of the next search's first batch (`retest`): same table
        // order (ip-2, then ip, then ip+1 ...), one candidate round trip instead of two.
        bool retest = false;
        for (;;) {
            // ------------------------------------------------------------ search (lz4.c:1014-1076)
            uint32_t ip, cand;
            bool rt_hit = false;
            uint64_t e_ipx = 0, e_cx = 0; uint32_t e_ipb = 0, e_cb = 0; bool e_regs = false;   // extension data of the hit
            // Probe width: candidate checks are random 64 KiB-window gathers (one cache line each), the
            // real cost of a batch.  In compressible data a match turns up within a few probes, so a
            // search starts 16 lanes wide and doubles after every batch that found nothing.
            uint32_t width = 16;
            for (uint32_t k0 = 0;; ) {
                const bool act = uint32_t(lane) < width;
                uint32_t pos, next;
                if (k0 == 0) { pos = sp + lane; next = pos + 1; }       // the first 65 probes of a search are 1 apart
                else { pos = sp + probe_offset(k0 + lane); next = sp + probe_offset(k0 + lane + 1); }
                const bool in_range = next <= lim;                     // else: this probe ends the block
                // One 16-byte load per lane covers everything needed on the cursor side: [pos-4, pos) for the
                // backward extension, [pos, pos+8) for the hash, [pos+4, pos+12) for the forward extension
                // (in_range lanes have pos+12 <= n; the first few positions of a block take the plain path).
                const bool wide = sp >= 4;
                const uint32_t rp = in_range ? pos : (wide ? 4u : 0u);   // keep speculative reads in bounds
                uint64_t v8, ipx = 0; uint32_t ipb = 0;
                if (wide && k0 == 0) {
                    // probes of a search's first (stride-1) batch: from the cursor window when they are
                    // inside it, else re-centre the window on this search
                    if (!(wvalid && sp >= wsp && sp + width <= wsp + 64 && wsp + 64 + 12 <= un)) {
                        W = ld16(src + rp - 4); wsp = sp; wvalid = (sp + 64 + 12 <= un);
                    }
                    Q16 q = W;
                    if (wvalid && sp != wsp) {
                        const int sl = (lane + int(sp - wsp)) & 63;
                        q.d0 = __shfl(W.d0, sl); q.d1 = __shfl(W.d1, sl); q.d2 = __shfl(W.d2, sl); q.d3 = __shfl(W.d3, sl);
                    }
                    ipb = q.d0; v8 = u64(q.d1, q.d2); ipx = u64(q.d2, q.d3);
                } else if (wide) {
                    const Q16 q = ld16(src + rp - 4);
                    ipb = q.d0; v8 = u64(q.d1, q.d2); ipx = u64(q.d2, q.d3);
                } else v8 = ld8(src + rp);
                const uint32_t h = hash_of<U32TAB>(v8);
                // re-test batch: position sp-2 enters the table first; its 8 bytes are bytes 2..9 of lane 0's 16
                const bool rt = retest && k0 == 0;
                uint32_t h2 = 0xFFFFFFFFu;
                if (rt) h2 = rl(hash_of<U32TAB>(u64(__builtin_amdgcn_alignbit(uint32_t(v8), ipb, 16), __builtin_amdgcn_alignbit(uint32_t(v8 >> 32), uint32_t(v8), 16))), 0);
                uint32_t c = 0, cw = 0; uint64_t cx = 0; uint32_t cb = 0; bool shared = false, cregs = false;
                if (act) {
                    c = tab_get(h);
                    if (h == h2) c = sp - 2;                           // the refill of ip-2 comes before every probe of this batch
                    // scoreboard: does an earlier lane of this batch touch the same (folded) slot?
                    uint32_t* sc = &score[h & (kScore - 1)];
                    atomicMin(sc, uint32_t(lane));
                    shared = (*sc != uint32_t(lane));
                    *sc = 0xFFFFFFFFu;
                    // candidate side, same shape: [c-4, c+12) in one load (c + 12 <= n always holds)
                    if (wide && c >= 4) { const Q16 q = ld16(src + c - 4); cb = q.d0; cw = q.d1; cx = u64(q.d2, q.d3); cregs = true; }
                    else cw = ld4(src + c);
                }
                const bool hit = act && in_range && (!U32TAB || c + kMaxDist >= pos) && cw == uint32_t(v8);
                const unsigned long long m_cut  = __ballot(shared);
                const unsigned long long m_term = __ballot(act && !in_range);
                const unsigned long long m_hit  = __ballot(hit);
                const int cut = m_cut ? __builtin_ctzll(m_cut) : int(width);
                // first lane (serial order) that ends this batch
                const unsigned long long ev = (m_term | m_hit) & ((cut >= 64) ? ~0ull : ((1ull << cut) - 1));
                const int e = ev ? __builtin_ctzll(ev) : cut;
                const bool e_is_hit = ev && ((m_hit >> e) & 1) && !((m_term >> e) & 1);
                // commit table writes of the probes that really happen (after the ip-2 refill of a re-test batch)
                if (rt && lane == 0) tab_put(h2, sp - 2);
                if (lane < e || (lane == e && e_is_hit)) tab_put(h, pos);
                if (ev) {
                    if (!e_is_hit) goto last_literals;
                    ip   = __builtin_amdgcn_readlane(pos, e);
                    cand = __builtin_amdgcn_readlane(c, e);
                    if (rl(uint32_t(cregs), e)) {
                        e_regs = true;
                        e_ipx = u64(rl(uint32_t(ipx), e), rl(uint32_t(ipx >> 32), e));
                        e_cx  = u64(rl(uint32_t(cx), e), rl(uint32_t(cx >> 32), e));
                        e_ipb = rl(ipb, e); e_cb = rl(cb, e);
                    }
                    rt_hit = rt && e == 0;                            // the immediate re-test hit: no literals, no catch-up
                    break;
                }
                k0 = U(k0 + e);
                if (rt) { sp = U(sp + 1); k0 = U(k0 - 1); retest = false; }   // the search proper starts one past the re-test
                if (e == int(width) && width < 64) width *= 2;
            }
            retest = false;
            K2PH(pt_search);
            // ------------------------------------------------------------ catch up (lz4.c:1080)
            {
                // forward bytes already known from the hit registers (relative to the ORIGINAL ip)
                uint32_t fwd_known = 0; bool fwd_done = false;
                if (e_regs) {
                    const uint64_t x = e_ipx ^ e_cx;
                    const uint32_t eq = x ? uint32_t(__builtin_ctzll(x) >> 3) : 8u;
                    const uint32_t room = matchlimit - (ip + 4);       // ip < lim  =>  room >= 3
                    fwd_known = min(eq, room);
                    fwd_done = (eq < 8) || (room <= 8);
                }
                uint32_t token_pos, tok;      // the token byte is written once both nibbles are known
                if (rt_hit) { token_pos = op++; tok = 0; }             // lz4.c:1250-1256: zero literals, straight to _next_match
                else {
                const uint32_t maxback = min(ip - anchor, cand);
                uint32_t back = 0; bool more = maxback > 0;
                if (e_regs && more && cand >= 4) {                     // first 4 bytes from registers
                    const uint32_t x = e_ipb ^ e_cb;                   // byte 3 (MSB) is position -1
                    const uint32_t eq = x ? uint32_t(__builtin_clz(x) >> 3) : 4u;
                    back = min(eq, maxback);
                    more = (eq == 4 && maxback > 4);
                }
                while (more) {
                    const uint32_t j = back + uint32_t(lane) + 1;
                    const bool ok = (j <= maxback) && src[ip - j] == src[cand - j];
                    const unsigned long long bad = ~__ballot(ok);
                    const int b = bad ? __builtin_ctzll(bad) : 64;
                    back += b;
                    if (b < 64) break;
                }
                back = U(back);
                ip = U(ip - back); cand = U(cand - back);
                // everything in [new ip, old ip + 4 + fwd_known) is equal: the forward count restarts from
                // the new ip+4, so the `back` bytes just walked over are already part of it
                if (e_regs) fwd_known += back;
                // ---------------------------------------------------------- literals (lz4.c:1083-1107)
                {
                    const uint32_t lit = ip - anchor;
                    token_pos = op++;
                    if (limited && op + lit + (2 + 1 + kLastLit) + lit / 255 > uint32_t(cap)) return 0;
                    if (lit >= 15) { tok = 0xF0; op += emit_len(dst + op, lit - 15, lane); }
                    else tok = lit << 4;
                    if (wvalid && anchor + 1 >= wsp && ip <= wsp + 64) {
                        // the run lies inside the cursor window: lane l owns position wsp+l (byte 4 of its
                        // 16), lane 0 also owns wsp-1 (byte 3)
                        const uint32_t q = wsp + lane;
                        if (q >= anchor && q < ip) dst[op + (q - anchor)] = uint8_t(W.d1);
                        if (lane == 0 && anchor + 1 == wsp && lit) dst[op] = uint8_t(W.d0 >> 24);
                    } else copy_bytes(dst + op, src + anchor, lit, lane);
                    op += lit;
                }
                }
                K2PH(pt_ext);
                {   // _next_match (lz4.c:1109-1200)
                    const uint32_t off = ip - cand;
                    if (lane == 0) { dst[op] = uint8_t(off); dst[op + 1] = uint8_t(off >> 8); }
                    op += 2;
                    // forward extension: bytes equal from ip+4 / cand+4, bounded by matchlimit
                    uint32_t mcode = fwd_known;
                    if (!fwd_done) {
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
                    if (limited && op + (1 + kLastLit) + (mcode + 240) / 255 > uint32_t(cap)) return 0;
                    uint32_t tok_add;
                    if (mcode >= 15) { tok_add = 15; op += emit_len(dst + op, mcode - 15, lane); }
                    else tok_add = mcode;
                    if (lane == 0) dst[token_pos] = uint8_t(tok + tok_add);
                    anchor = ip; op = U(op);
                    if (ip >= lim) goto last_literals;
                    // the refill of ip-2 and the immediate re-test of ip (lz4.c:1207-1259) ride on the next batch
                }
            }
            sp = U(ip); retest = true; anchor = U(anchor); op = U(op);
            K2PH(pt_match);
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
    if (lane == 0 && n == (4 << 20)) { uint64_t* c = reinterpret_cast<uint64_t*>(dst + n - 32); c[0] = pt_search; c[1] = pt_ext; c[2] = pt_match; }
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
    const fourmc_block blk = blocks[b];
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
