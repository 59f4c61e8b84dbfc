/*
 * tools/model/seg_decode_model.c - executable model of the segment-parallel LZ4 decoder (4mc_amd/csrc/lz4_seg.hip), lane by lane.
 *
 * Written BEFORE the kernels: it fixes the arithmetic of the three steps (speculative segment walk, resolution of the true token
 * chain, batch execution through a zeroed staging buffer with OR stores) and checks them against the oracle on the corpus, on edge
 * inputs and on damaged streams.  A "wave" is a loop over 64 lanes; every per-lane expression is the one the kernel evaluates.
 * (The kernels' records carry two words since the end of round 4 - position | literal length, offset | match length - so that the
 * executor need not fetch token and offset from the stream; the model keeps the first word and reads the second's contents from the
 * stream: the same values.)
 * Design aid / test infrastructure only (links the oracle); not product.
 *
 *   gcc -O2 -o /tmp/seg_decode_model tools/model/seg_decode_model.c tools/corpus.c -Ioracle oracle/liboracle.so -Wl,-rpath,$PWD/oracle
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "oracle.h"
void corpus_fill(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);
void corpus_fill_logs(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);

#define NL        64            /* lanes = segments */
#define F_CAP     128           /* fix list entries per segment */
#define MARGIN    64            /* tokens whose bytes end beyond csize - MARGIN belong to the tail (exact walker) */
#define OMARGIN   128           /* sequences whose output ends beyond cap - OMARGIN belong to the tail */
#define MINSEG    1024
#define ESC_LL    511u
#define POS_BITS  23
#define POS_MASK  ((1u << POS_BITS) - 1u)
#define CAPB      4032          /* staging bytes a batch may produce */
#define PRO       32            /* prologue: the 32 output bytes in front of the batch */
#define SBYTES    (64 + 16 + CAPB + 64)
#define kRetry    (-1000000003)

typedef struct { uint32_t pos, ll, ml, off, next; int esc, stop; } Tok;

/* one token at p; stop = it (or its bytes) reaches beyond limit = csize - MARGIN: the chain halts AT p */
static Tok decode_tok(const uint8_t* s, int csize, uint32_t p)
{
    Tok t; memset(&t, 0, sizeof t); t.pos = p;
    const uint32_t limit = (uint32_t)(csize - MARGIN);
    if (p >= limit) { t.stop = 1; return t; }
    uint32_t tok = s[p], q = p + 1, ll = tok >> 4, mn = tok & 15;
    if (ll == 15) for (;;) { if (q >= limit) { t.stop = 1; return t; } uint32_t b = s[q++]; ll += b; if (b != 255) break; }
    if (ll > (1u << 23)) { t.stop = 1; return t; }
    uint32_t mo = q + ll;
    if (mo + 2 > limit) { t.stop = 1; return t; }
    t.off = s[mo] | (s[mo + 1] << 8);
    uint32_t q2 = mo + 2, ml = mn + 4, mlx = 0;
    if (mn == 15) for (;;) { if (q2 >= limit) { t.stop = 1; return t; } uint32_t b = s[q2++]; ml += b; mlx++; if (b != 255) break; }
    if (q2 > limit) { t.stop = 1; return t; }
    t.ll = ll; t.ml = ml; t.next = q2; t.esc = (ll >= ESC_LL) || (mlx > 2);
    return t;
}
static uint32_t pack_rec(const Tok* t) { return t->pos | ((t->esc ? ESC_LL : t->ll) << POS_BITS); }

/* ------------------------------------------------------------------------------------------------ step 1 + 2: walk and resolve */
typedef struct {
    uint32_t seglen, nseg;
    uint32_t* F[NL]; uint32_t* L[NL];        /* fix lists, spec lists */
    uint32_t f[NL], k[NL], n[NL], exitp[NL], entry[NL]; int tail[NL];
    uint32_t live[NL], nlive, tail_ip;
    long stat_fix_hops, stat_rewalks, stat_rounds;
} Walk;

static void walk_from(Walk* W, const uint8_t* s, int csize, int j, uint32_t start)
{   /* (re)walk segment j from `start`, recording into L[j] from index 0 */
    const uint32_t seg_end = (j + 1 == (int)W->nseg) ? 0xFFFFFFFFu : (uint32_t)(j + 1) * W->seglen;
    uint32_t p = start, n = 0;
    for (;;) {
        if (p >= seg_end) { W->tail[j] = 0; break; }
        Tok t = decode_tok(s, csize, p);
        if (t.stop) { W->tail[j] = 1; break; }
        W->L[j][n++] = pack_rec(&t); p = t.next;
    }
    W->exitp[j] = p; W->n[j] = n; W->f[j] = 0; W->k[j] = 0; W->entry[j] = start;
}
static void fix_from(Walk* W, const uint8_t* s, int csize, int j, uint32_t e)
{   /* the true chain enters segment j at e: walk it until it falls onto the recorded chain */
    const uint32_t seg_end = (j + 1 == (int)W->nseg) ? 0xFFFFFFFFu : (uint32_t)(j + 1) * W->seglen;
    uint32_t q = e, idx = 0, f = 0; const uint32_t n = W->n[j];
    /* NB the recorded list may itself be the product of an earlier fix: only valid when f == 0 && k == 0 (pure list) */
    for (;;) {
        while (idx < n && (W->L[j][idx] & POS_MASK) < q) idx++;
        if (idx < n && (W->L[j][idx] & POS_MASK) == q) { W->k[j] = idx; break; }                 /* merged */
        if (idx == n && q == W->exitp[j]) { W->k[j] = n; break; }                                    /* merged at the chain's end */
        if (q >= seg_end) { W->k[j] = n; W->exitp[j] = q; W->tail[j] = 0; break; }
        Tok t = decode_tok(s, csize, q);
        if (t.stop) { W->k[j] = n; W->exitp[j] = q; W->tail[j] = 1; break; }
        if (f == F_CAP) { W->stat_rewalks++; walk_from(W, s, csize, j, e); return; }
        W->F[j][f++] = pack_rec(&t); q = t.next; W->stat_fix_hops++;
    }
    W->f[j] = f; W->entry[j] = e;
}
static void walk_block(Walk* W, const uint8_t* s, int csize)
{
    const uint32_t limit = (uint32_t)(csize - MARGIN);
    uint32_t nseg = limit / MINSEG; if (nseg < 1) nseg = 1; if (nseg > NL) nseg = NL;
    W->nseg = nseg; W->seglen = ((limit + nseg - 1) / nseg + 3) & ~3u;
    for (uint32_t j = 0; j < nseg; j++) walk_from(W, s, csize, (int)j, j * W->seglen);          /* phase 1: all lanes at once */
    /* a list that was produced from its segment start is "pure"; fix_from needs a pure list.  A second fix of the same lane
       (entry changed) therefore re-walks: keep it simple, it is rare. */
    uint8_t pure[NL]; memset(pure, 1, sizeof pure);
    uint32_t ex0[NL]; int tl0[NL]; for (uint32_t j = 0; j < nseg; j++) { ex0[j] = W->exitp[j]; tl0[j] = W->tail[j]; }
    for (uint32_t j = 1; j < nseg; j++) {                                                          /* phase 2: optimistic fix-up, all lanes at once */
        const uint32_t pe = ex0[j - 1];
        if (tl0[j - 1]) continue;
        uint32_t sj = pe / W->seglen; if (sj > nseg - 1) sj = nseg - 1;
        if (sj != j || pe == j * W->seglen) continue;
        fix_from(W, s, csize, (int)j, pe); pure[j] = (W->f[j] == 0 && W->k[j] == 0);
    }
    /* phase 3: follow the true path, redo what was assumed wrong */
    uint32_t cur = 0; W->nlive = 0;
    for (;;) {
        W->live[W->nlive++] = cur;
        if (W->tail[cur]) { W->tail_ip = W->exitp[cur]; break; }
        const uint32_t e = W->exitp[cur];
        uint32_t j = e / W->seglen; if (j > nseg - 1) j = nseg - 1;
        if (W->entry[j] != e) {
            W->stat_rounds++;
            if (pure[j]) { fix_from(W, s, csize, (int)j, e); pure[j] = (W->f[j] == 0 && W->k[j] == 0); }
            else walk_from(W, s, csize, (int)j, e);
        }
        cur = j;
    }
}

/* ------------------------------------------------------------------------------------------------ step 3: execute */
typedef struct { uint8_t st[SBYTES + 64]; long batches, passes, esc, iters_lit, sets; } Exec;

/* OR `len` (1..32) string bytes into the staging buffer at byte address pd; R[0..8] holds the string PHASE-ALIGNED to the
   destination: string byte b sits at byte (da + b) of R, da = pd & 3.  Bytes of R outside the string are arbitrary. */
static void or_store(uint8_t* st, uint32_t pd, const uint32_t R[9], uint32_t len)
{
    const uint32_t da = pd & 3, nd = (da + len + 3) >> 2, last = nd - 1, tb = (da + len) & 3;
    const uint32_t hmask = 0xFFFFFFFFu << (8 * da), tmask = tb ? ((1u << (8 * tb)) - 1) : 0xFFFFFFFFu;
    uint32_t* w = (uint32_t*)(st + (pd & ~3u));
    for (uint32_t j = 0; j < 9; j++) {
        uint32_t v = R[j];
        if (j == 0) v &= hmask;
        if (j == last) v &= tmask;
        if (j > last) v = 0;
        w[j] |= v;
    }
}
/* nine phase-aligned dwords of the string that starts at byte address a of `base` (global memory: unaligned loads at a - da) */
static void load_phase_global(const uint8_t* base, long a, uint32_t da, uint32_t R[9], long lo_bound)
{
    long ga = a - (long)da;
    uint8_t tmp[36];
    for (int i = 0; i < 36; i++) tmp[i] = (ga + i >= lo_bound) ? base[ga + i] : 0;   /* kernel: guard branch when ga < lo_bound */
    memcpy(R, tmp, 36);
}
/* the same from the staging buffer: aligned dwords + byte shift (v_perm) */
static void load_phase_lds(const uint8_t* st, uint32_t ps, uint32_t da, uint32_t R[9])
{
    const uint32_t sa = ps & 3; int sigma = (int)sa - (int)da;            /* string byte b is at R'byte (sigma' + b + da) ... */
    uint32_t rb = (ps & ~3u); if (sigma < 0) { rb -= 4; sigma += 4; }
    const uint32_t* w = (const uint32_t*)(st + rb);
    for (int j = 0; j < 9; j++) {
        const uint64_t pair = (uint64_t)w[j] | ((uint64_t)w[j + 1] << 32);
        R[j] = (uint32_t)(pair >> (8 * sigma));
    }
}

static int exec_block(Exec* X, const Walk* W, const uint8_t* s, int csize, uint8_t* dst, int cap, uint32_t* res_ip, uint32_t* res_op)
{
    uint32_t opos = 0;
    memset(X->st, 0, sizeof X->st);
    for (uint32_t li = 0; li < W->nlive; li++) {
        const uint32_t j = W->live[li], f = W->f[j], k = W->k[j], c = f + W->n[j] - k;
        uint32_t t0 = 0;
        while (t0 < c) {
            /* ---- records and fields, one sequence per lane */
            uint32_t pos[NL], ll[NL], ml[NL], off[NL], lsrc[NL], sz[NL], incl[NL]; int valid[NL], esc[NL];
            for (int l = 0; l < NL; l++) {
                const uint32_t t = t0 + l; valid[l] = t < c; esc[l] = 0; sz[l] = 0; ll[l] = ml[l] = off[l] = lsrc[l] = pos[l] = 0;
                if (!valid[l]) continue;
                const uint32_t rec = t < f ? W->F[j][t] : W->L[j][k + t - f];
                pos[l] = rec & POS_MASK; ll[l] = rec >> POS_BITS; esc[l] = ll[l] == ESC_LL;
                if (esc[l]) continue;
                const uint32_t llx = ll[l] < 15 ? 0 : 1 + (ll[l] >= 270);
                lsrc[l] = pos[l] + 1 + llx;
                const uint32_t tok = s[pos[l]], mo = lsrc[l] + ll[l];
                const uint32_t w1 = s[mo] | (s[mo + 1] << 8) | (s[mo + 2] << 16) | ((uint32_t)s[mo + 3] << 24);
                off[l] = w1 & 0xFFFF; ml[l] = (tok & 15) + 4;
                if ((tok & 15) == 15) { const uint32_t e0 = (w1 >> 16) & 255; ml[l] += e0; if (e0 == 255) ml[l] += w1 >> 24; }
                sz[l] = ll[l] + ml[l];
            }
            uint32_t run = 0; for (int l = 0; l < NL; l++) { run += sz[l]; incl[l] = run; }
            /* ---- how many sequences this batch takes: a prefix */
            int cnt = 0, cut_cap = 0;
            for (int l = 0; l < NL; l++) {
                if (!valid[l] || esc[l] || incl[l] > CAPB) break;
                if (opos + incl[l] > (uint32_t)cap - OMARGIN || incl[l] > 0x7FFFFFFFu) { cut_cap = 1; break; }
                cnt++;
            }
            if (cnt == 0) {
                if (cut_cap) { *res_ip = pos[0]; *res_op = opos; return 1; }                 /* the tail starts here (output side) */
                /* lane 0 is an escape: one long sequence, wave-wide, straight in memory */
                Tok t = decode_tok(s, csize, pos[0]);
                if (t.stop) return kRetry;                                                     /* cannot happen: the walk decoded it */
                if ((uint64_t)opos + t.ll + t.ml > (uint64_t)cap - OMARGIN) { *res_ip = pos[0]; *res_op = opos; return 1; }
                /* literal start: recompute like the walk */
                uint32_t q = t.pos + 1; if ((s[t.pos] >> 4) == 15) { for (;;) { uint32_t b = s[q++]; if (b != 255) break; } }
                memcpy(dst + opos, s + q, t.ll);
                const uint32_t m = opos + t.ll;
                if (t.off == 0 || t.off > m) return kRetry;
                for (uint32_t i = 0; i < t.ml; i++) dst[m + i] = dst[m + i - t.off];
                opos = m + t.ml; t0 += 1; X->esc++;
                /* prologue from memory */
                { const uint32_t P0 = 64 + (uint32_t)((uintptr_t)(dst + opos) & 15);
                  memset(X->st, 0, sizeof X->st);
                  for (int i = 0; i < PRO; i++) if (opos >= (uint32_t)(PRO - i)) X->st[P0 - PRO + i] = dst[opos - PRO + i]; }
                continue;
            }
            X->batches++;
            const uint32_t T = incl[cnt - 1];
            const uint32_t P0 = 64 + (uint32_t)((uintptr_t)(dst + opos) & 15);
            uint32_t outl[NL], mrel[NL];
            for (int l = 0; l < cnt; l++) {
                outl[l] = incl[l] - sz[l]; mrel[l] = outl[l] + ll[l];
                if (off[l] == 0 || off[l] > opos + mrel[l]) return kRetry;
            }
            /* ---- literals */
            {
                uint32_t rem[NL], sp[NL], dp[NL]; int any = 0;
                for (int l = 0; l < cnt; l++) { rem[l] = ll[l]; sp[l] = lsrc[l]; dp[l] = outl[l]; any |= rem[l] > 0; }
                while (any) {
                    any = 0; X->iters_lit++;
                    for (int l = 0; l < cnt; l++) if (rem[l]) {
                        const uint32_t len = rem[l] < 32 ? rem[l] : 32, pd = P0 + dp[l]; uint32_t R[9];
                        load_phase_global(s, (long)sp[l], pd & 3, R, 0);
                        or_store(X->st, pd, R, len);
                        sp[l] += len; dp[l] += len; rem[l] -= len; any |= rem[l] > 0;
                    }
                }
            }
            /* ---- matches.  M1: sources that END in front of the batch - 32 bytes per lane and step, from memory (everything below
               opos has been flushed).  M2: everything else (sources in the batch or its prologue, overlapping matches), one sequence at
               a time in lane order, a byte per lane, inside the staging buffer (bytes from before the prologue: memory). */
            {
                uint64_t m2 = 0;
                for (int l = 0; l < cnt; l++) {
                    const long x = (long)mrel[l] - (long)off[l];
                    if (x + (long)ml[l] <= 0) {
                        uint32_t rem = ml[l], dp = mrel[l]; long sa = (long)opos + x;
                        while (rem) {
                            const uint32_t len = rem < 32 ? rem : 32, pd = P0 + dp; uint32_t R[9];
                            load_phase_global(dst, sa, pd & 3, R, 0);
                            or_store(X->st, pd, R, len); X->sets++;
                            sa += len; dp += len; rem -= len;
                        }
                    } else m2 |= 1ull << l;
                }
                while (m2) {
                    const int l = __builtin_ctzll(m2); m2 &= m2 - 1; X->passes++;
                    const long m = mrel[l], o = off[l];
                    for (long i = 0; i < (long)ml[l]; i++) {
                        const long sx = m - o + i;
                        X->st[P0 + m + i] = sx < -(long)PRO ? dst[(long)opos + sx] : X->st[(long)P0 + sx];
                    }
                }
            }
            /* ---- flush + prologue + zero */
            memcpy(dst + opos, X->st + P0, T);
            {
                uint8_t pro[PRO]; memcpy(pro, X->st + P0 + T - PRO, PRO);
                const uint32_t P1 = 64 + (uint32_t)((uintptr_t)(dst + opos + T) & 15);
                memset(X->st, 0, sizeof X->st);
                memcpy(X->st + P1 - PRO, pro, PRO);
            }
            opos += T; t0 += cnt;
        }
    }
    *res_ip = W->tail_ip; *res_op = opos;
    return 1;
}

/* the exact walker from (ip, op): valid streams only (the model's stand-in for the resumed exact kernel) */
static int tail_decode(const uint8_t* s, int csize, uint8_t* dst, int cap, uint32_t ip, uint32_t op)
{
    for (;;) {
        if ((int)ip >= csize) return -1;
        uint32_t tok = s[ip++], ll = tok >> 4, ml = tok & 15;
        if (ll == 15) for (;;) { if ((int)ip >= csize) return -1; uint32_t b = s[ip++]; ll += b; if (b != 255) break; }
        if ((uint64_t)ip + ll > (uint64_t)csize || (uint64_t)op + ll > (uint64_t)cap) return -1;
        memcpy(dst + op, s + ip, ll); ip += ll; op += ll;
        if ((int)ip == csize) return (int)op;
        if ((int)ip + 2 > csize) return -1;
        uint32_t off = s[ip] | (s[ip + 1] << 8); ip += 2;
        if (ml == 15) for (;;) { if ((int)ip >= csize) return -1; uint32_t b = s[ip++]; ml += b; if (b != 255) break; }
        ml += 4;
        if (off == 0 || off > op || (uint64_t)op + ml > (uint64_t)cap) return -1;
        for (uint32_t i = 0; i < ml; i++) dst[op + i] = dst[op + i - off];
        op += ml;
    }
}

static Walk W; static Exec X;
static uint8_t* gbuf;   /* guarded output */
static int model_decode(const uint8_t* s, int csize, uint8_t* dst, int cap, uint32_t* rip, uint32_t* rop)
{
    if (csize < 256 || cap < 256) return kRetry;
    walk_block(&W, s, csize);
    return exec_block(&X, &W, s, csize, dst, cap, rip, rop);
}

static uint64_t rs = 0x1234567;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

int main(int argc, char** argv)
{
    const uint32_t B = 4u << 20;
    int nfuzz = argc > 1 ? atoi(argv[1]) : 200;
    uint8_t* in = malloc(B); uint8_t* c = malloc(B + B / 255 + 64 + 64); uint8_t* ref = malloc(B + 64);
    gbuf = malloc(B + 4096); uint8_t* out = gbuf + 2048;
    for (int j = 0; j < NL; j++) { W.F[j] = malloc(4 * F_CAP); W.L[j] = malloc(4 * (B / 3 + 16)); }
    int bad = 0;
    /* 1. corpus blocks, whole */
    for (int pass = 0; pass < 2; pass++) for (int blk = 0; blk < (pass ? 8 : 24); blk++) {
        (pass ? corpus_fill_logs : corpus_fill)(in, B, 0x4D43, blk);
        int cs = orc_lz4_compress_fast(in, c, B, B - 1);
        if (cs <= 0) continue;
        memset(gbuf, 0xA5, B + 4096); memset(&X, 0, sizeof X); W.stat_fix_hops = W.stat_rewalks = W.stat_rounds = 0;
        uint32_t rip = 0, rop = 0;
        int r = model_decode(c, cs, out, B, &rip, &rop);
        int fin = r == 1 ? tail_decode(c, cs, out, B, rip, rop) : r;
        int ok = fin == (int)B && !memcmp(out, in, B);
        for (int i = 0; i < 2048; i++) if (gbuf[i] != 0xA5 || gbuf[2048 + B + i] != 0xA5) ok = 0;
        printf("%s blk %2d csize %7d: %s  live %u fixhops %ld rewalks %ld rounds %ld | batches %ld M2/batch %.2f esc %ld tail at ip %u (csize-%d) op %u\n",
               pass ? "logs" : "smix", blk, cs, ok ? "OK" : "MISMATCH", W.nlive, W.stat_fix_hops, W.stat_rewalks, W.stat_rounds,
               X.batches, X.batches ? (double)X.passes / X.batches : 0, X.esc, rip, cs - (int)rip, rop);
        if (!ok) { bad++; for (uint32_t i = 0; i < B; i++) if (out[i] != in[i]) { printf("   first diff at %u (decoded %d)\n", i, fin); break; } }
    }
    /* 2. small / odd sizes and damaged streams against the oracle's verdict */
    long accepted = 0, retried = 0, checked = 0;
    for (int it = 0; it < nfuzz; it++) {
        const int blk = rnd() % 48; uint32_t n = 300 + rnd() % (it % 3 == 0 ? 400000 : 20000);
        corpus_fill(in, B, 0x4D43, blk);
        const uint32_t o = rnd() % (B - n);
        if (it % 5 == 0) memset(in + o + n / 3, it & 255, n / 4);                        /* long runs */
        if (it % 7 == 0) for (uint32_t i = n / 2; i < n / 2 + n / 5; i++) in[o + i] = (uint8_t)rnd();   /* long literals */
        int cs = orc_lz4_compress_fast(in + o, c, (int)n, (int)n + 64);
        if (cs <= 0) continue;
        int nm = it % 2 ? 1 + rnd() % 3 : 0;
        for (int m = 0; m < nm; m++) { uint32_t at = rnd() % cs; c[at] = (rnd() & 1) ? (uint8_t)rnd() : (c[at] ^ (1u << (rnd() & 7))); }
        int cap = (int)n - (it % 11 == 0 ? (int)(rnd() % 40) : 0) + (it % 13 == 0 ? 17 : 0);
        memset(ref, 0, n + 64);
        int rr = orc_lz4_decompress_safe(c, ref, cs, cap);
        memset(gbuf, 0xA5, B + 4096); memset(&X, 0, sizeof X);
        uint32_t rip = 0, rop = 0;
        int r = model_decode(c, cs, out, cap, &rip, &rop);
        checked++;
        int ok = 1;
        for (int i = 0; i < 2048; i++) if (gbuf[i] != 0xA5) ok = 0;
        for (int i = 0; i < 2048; i++) if (gbuf[2048 + cap + i] != 0xA5) ok = 0;       /* nothing beyond the capacity */
        if (r == 1) {
            accepted++;
            /* the oracle must not have failed before the hand-over point, and the bytes so far must be its bytes */
            if (rr < 0 && rr != INT32_MIN && (uint32_t)(-rr - 1) < rip) ok = 0;
            if (rr == INT32_MIN) ok = ok && 1;     /* offset 0 somewhere: the oracle's sentinel - must lie in the tail, or we would have retried */
            if (rr != INT32_MIN && memcmp(out, ref, rop)) ok = 0;
            if (rr >= 0 && (int)rop > rr) ok = 0;
            if (rr >= 0) { int fin = tail_decode(c, cs, out, cap, rip, rop); if (nm == 0 && (fin != rr || memcmp(out, ref, rr))) ok = 0; }
        } else retried++;
        if (!ok) { bad++; printf("fuzz %d: n %u cs %d cap %d muts %d: oracle %d model %d rip %u rop %u  BAD\n", it, n, cs, cap, nm, rr, r, rip, rop); }
    }
    printf("fuzz: %ld streams, %ld through the fast path to the tail, %ld handed back; %d bad\n", checked, accepted, retried, bad);
    return bad != 0;
}
