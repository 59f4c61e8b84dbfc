/*
 * tools/model/seg_model.c - statistics behind the segment-parallel LZ4 decoder (lz4_seg.hip): how fast a token chain started at
 * an arbitrary stream byte falls onto the true chain, and how many dependency passes a batch of 64 consecutive sequences needs.
 * Test / design aid only (links the oracle for the streams); not product.
 *   gcc -O2 -o /tmp/seg_model tools/model/seg_model.c tools/corpus.c -Ioracle oracle/liboracle.so -Wl,-rpath,$PWD/oracle
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "oracle.h"
void corpus_fill(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);
void corpus_fill_logs(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);
#define B (4u << 20)
typedef struct { uint32_t sp, ll, lit_src, ml, off, next; } Tok;

static int decode_tok(const uint8_t* s, int csize, uint32_t p, Tok* t)
{   /* returns 0 if the token cannot be decoded inside [0, csize - 64) */
    int limit = csize - 64;
    if ((int)p >= limit) return 0;
    uint32_t tok = s[p], q = p + 1, ll = tok >> 4, ml = tok & 15;
    if (ll == 15) { for (;;) { if ((int)q >= limit) return 0; uint32_t b = s[q++]; ll += b; if (b != 255) break; } }
    t->sp = p; t->ll = ll; t->lit_src = q;
    q += ll; if ((int)q + 2 > limit) return 0;
    t->off = s[q] | (s[q + 1] << 8); q += 2;
    if (ml == 15) { for (;;) { if ((int)q >= limit) return 0; uint32_t b = s[q++]; ml += b; if (b != 255) break; } }
    t->ml = ml + 4; t->next = q;
    return (int)q <= limit;
}

int main(int argc, char** argv)
{
    int logs = argc > 1 && !strcmp(argv[1], "logs");
    int S = argc > 2 ? atoi(argv[2]) : 64;
    int T = argc > 3 ? atoi(argv[3]) : 31;
    uint8_t* in = malloc(B); uint8_t* c = malloc(B + B / 255 + 64);
    Tok* toks = malloc(sizeof(Tok) * (B / 3)); uint32_t* outp = malloc(4 * (B / 3));
    uint8_t* istok = malloc(B + 64);
    printf("blk csize ntok  | merge: mean max nofail | batch: lits>T m>T ovl near | passes wm exact | maxchain\n");
    double tot_b = 0, tot_pw = 0, tot_pe = 0;
    for (int blk = 0; blk < 48; blk++) {
        (logs ? corpus_fill_logs : corpus_fill)(in, B, 0x4D43, blk);
        int cs = orc_lz4_compress_fast(in, c, B, B - 1);
        if (cs <= 0) { printf("%2d stored\n", blk); continue; }
        /* true chain */
        int n = 0; uint32_t p = 0, op = 0; memset(istok, 0, cs + 64);
        Tok t;
        while (decode_tok(c, cs, p, &t)) { toks[n] = t; outp[n] = op; op += t.ll + t.ml; istok[p] = 1; n++; p = t.next; }
        /* spec merge stats */
        int seglen = (cs + S - 1) / S; double msum = 0; int mmax = 0, nofail = 0, cnt = 0;
        for (int j = 1; j < S; j++) {
            uint32_t sj = j * seglen, ej = (j + 1) * seglen; if ((int)sj >= cs - 64) break;
            /* spec chain: hops until it lands on a true token; count TRUE hops from the true entry until the merge point */
            uint32_t q = sj; int merged = 0; uint32_t mp = 0;
            while (q < ej && decode_tok(c, cs, q, &t)) { if (istok[q]) { merged = 1; mp = q; break; } q = t.next; }
            if (!merged) { nofail++; continue; }
            /* true hops from first true token >= sj to mp */
            uint32_t e = sj; while (!istok[e]) e++;
            int hops = 0; while (e < mp) { decode_tok(c, cs, e, &t); e = t.next; hops++; }
            msum += hops; if (hops > mmax) mmax = hops; cnt++;
        }
        /* batch stats */
        long rres = 0, rhops = 0, pr = 0; int rhmax = 0; long nb = 0, litsT = 0, mT = 0, ovl = 0, near = 0, pw = 0, pe = 0; int maxchain = 0;
        for (int b0 = 0; b0 + 64 <= n; b0 += 64) {
            uint32_t B0 = outp[b0]; int donew[64], lvl[64], rlvl[64];
            nb++;
            for (int l = 0; l < 64; l++) {
                Tok* k = &toks[b0 + l]; uint32_t m = outp[b0 + l] + k->ll;
                if (k->ll > (uint32_t)T) litsT++;
                if (k->ml > (uint32_t)T) mT++;
                if (k->off < k->ml) ovl++;
                uint32_t a = m - k->off, e = a + k->ml; if (e > m) e = m;
                if (e > B0) near++;
                /* exact level: 1 + max level of earlier match lanes whose dest range intersects [a,e) */
                int lv = 1;
                for (int j = 0; j < l; j++) { Tok* kj = &toks[b0 + j]; uint32_t mj = outp[b0 + j] + kj->ll;
                    if (mj < e && mj + kj->ml > a) if (lvl[j] + 1 > lv) lv = lvl[j] + 1; }
                lvl[l] = lv;
                /* redirect model: follow the source back through earlier matches of the batch while it lies inside ONE region */
                { uint32_t ra = a, re = e; int hops = 0, res = -1;
                  for (;;) {
                    if (re <= B0) { res = 0; break; }
                    /* find region containing ra */
                    int j; for (j = l - 1; j >= 0; j--) if (outp[b0 + j] <= ra) break;
                    if (j < 0) { res = (re <= B0) ? 0 : 100; break; }     /* starts below B0, reaches into the batch */
                    Tok* kj = &toks[b0 + j]; uint32_t oj = outp[b0 + j], mj = oj + kj->ll, ej = mj + kj->ml;
                    if (ra >= mj) { if (re <= ej) { ra -= kj->off; re -= kj->off; hops++; if (hops > 64) { res = 100; break; } continue; } res = 100; break; }
                    /* starts in literals of j */
                    if (re <= mj) { res = 0; break; }
                    res = 100; break; }
                  if (res == 0) { rlvl[l] = 1; rres++; if (hops > rhmax) rhmax = hops; rhops += hops; }
                  else { int lv2 = 1; for (int j2 = 0; j2 < l; j2++) { Tok* kj = &toks[b0 + j2]; uint32_t mj = outp[b0 + j2] + kj->ll;
                            if (mj < e && mj + kj->ml > a) if (rlvl[j2] + 1 > lv2) lv2 = rlvl[j2] + 1; }
                         rlvl[l] = lv2; } }
                /* watermark level: ready when all lanes j with m_j < e are done, or it is the frontier */
                int lw = 1;
                for (int j = 0; j < l; j++) { Tok* kj = &toks[b0 + j]; uint32_t mj = outp[b0 + j] + kj->ll;
                    if (mj < e) if (donew[j] + 1 > lw) lw = donew[j] + 1; }
                /* frontier rule cannot make it earlier than the max of previous lanes' levels (it is frontier only when all before are done) */
                donew[l] = lw;
            }
            int mw = 0, me = 0; for (int l = 0; l < 64; l++) { if (donew[l] > mw) mw = donew[l]; if (lvl[l] > me) me = lvl[l]; }
            { int mr = 0; for (int l = 0; l < 64; l++) if (rlvl[l] > mr) mr = rlvl[l]; pr += mr; } pw += mw; pe += me; if (me > maxchain) maxchain = me;
        }
        printf("%2d %7d %6d | %5.1f %4d %3d | %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f | %d  (out/seq %.1f)\n", blk, cs, n,
               cnt ? msum / cnt : 0, mmax, nofail,
               (double)litsT / nb, (double)mT / nb, (double)ovl / nb, (double)near / nb, (double)pw / nb, (double)pe / nb, maxchain, (double)op / n); printf("      redirect: resolved %.2f/batch hops/res %.2f max %d passes %.2f\n", (double)rres/nb, rres? (double)rhops/rres:0, rhmax, (double)pr/nb);
        tot_b += nb; tot_pw += pw; tot_pe += pe;
    }
    printf("all: passes watermark %.2f exact %.2f per batch\n", tot_pw / tot_b, tot_pe / tot_b);
    return 0;
}
