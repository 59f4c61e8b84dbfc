/*
 * tools/model/lz4p_model.c - executable model of the ratio-tolerance LZ4 encoder (4mc_amd/csrc/lz4_par_encode.hip), statement
 * by statement what the kernels do: 64 KiB segments of a block compressed independently by one wave each (window of 64
 * positions, every position inserted into an NENT x u16 table in groups of 16 - a group reads before it writes -, one candidate
 * per position - the table's, or the position 1, 2 or 4 back when it holds the same 4 bytes - compared 4 bytes backwards and 12
 * forwards, a position skipped when one of the next three has a longer match,
 * first eligible position at or after the cursor taken, matches that reach the 12 bytes extended), then the segments' sequences
 * stitched into ONE LZ4 block (the literals a segment ends with join the first sequence of the next).
 * Prints the size against the reference parse (oracle) per S-mix block and checks every payload with the oracle's
 * LZ4_decompress_safe restatement.  Test / design aid only; not product.
 *   gcc -O2 -o /tmp/lz4p_model tools/model/lz4p_model.c tools/corpus.c -Ioracle oracle/liboracle.so -Wl,-rpath,$PWD/oracle
 *   gcc -O2 -shared -fPIC -DLZ4P_NO_MAIN -o /tmp/liblz4p_model.so tools/model/lz4p_model.c      (tests: lz4p_model_encode)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define SEG 65536
static int NENT = 4096, GROUP = 16, BACK = 4, LAZY = 3, FWD = 12;
static unsigned NEAR = 0x16;                        /* bit d: a lane also looks d positions back in its row of 16 lanes */
static long n_ext, n_seq_tot, n_prevwin, n_older;

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int HASHV = 1;
static uint32_t slot(const uint8_t* p)
{
    uint32_t x = rd32(p), h;
    if (HASHV == 0) h = (x * 2654435761u) >> 16;
    else if (HASHV == 1) h = ((x & 0xFFFFFF) * 0x9E3779u + (x >> 8) * 0x85EBCAu) >> 16;         /* two 24-bit multiplies */
    else if (HASHV == 2) h = (((x & 0xFFFFFF) * 0x9E3779u) ^ ((x >> 8) * 0x85EBCBu)) >> 16;
    else h = ((x & 0xFFFFFF) * 0x9E3779u + ((x >> 8) & 0xFFFFFF) * 0xC2B2AEu) >> 15;
    return ((h & 0xFFFF) * (uint32_t)NENT) >> 16;
}

static uint8_t* put_len(uint8_t* o, uint32_t r) { while (r >= 255) { *o++ = 255; r -= 255; } *o++ = (uint8_t)r; return o; }
/* one sequence: literals [lit, lit+ll), then a match (ml >= 4) or nothing (ml == 0: the last literals of a block) */
static uint8_t* emit(uint8_t* o, const uint8_t* lit, uint32_t ll, uint32_t off, uint32_t ml)
{
    uint8_t* tok = o++;
    if (ll >= 15) { *tok = 0xF0; o = put_len(o, ll - 15); } else *tok = (uint8_t)(ll << 4);
    memcpy(o, lit, ll); o += ll;
    if (ml) {
        *o++ = (uint8_t)off; *o++ = (uint8_t)(off >> 8);
        if (ml - 4 >= 15) { *tok |= 15; o = put_len(o, ml - 4 - 15); } else *tok |= (uint8_t)(ml - 4);
    }
    return o;
}

/* sequences of segment [s0, s1) of the block in[0, n): complete sequences only, literal lengths counted from s0; returns the
   bytes written and in *tail the literals left after the last match */
static uint32_t encode_segment(const uint8_t* in, uint32_t n, uint32_t s0, uint32_t s1, uint8_t* out, uint32_t* tail)
{
    static uint16_t table[65536];
    uint8_t* o = out;
    memset(table, 0, sizeof table);
    const int pmax = (int)(s1 - 4) < (int)n - 32 ? (int)(s1 - 4) : (int)n - 32;     /* last position that may start a match */
    const int mend = (int)s1 < (int)n - 5 ? (int)s1 : (int)n - 5;                    /* matches end at or before */
    uint32_t sp = s0;
    for (uint32_t wb = s0; wb < s1 && (int)wb <= pmax; wb += 64) {
        uint32_t cand[64]; int mlen[64], bk[64];
        for (int g = 0; g < 64; g += GROUP) {
            for (int l = g; l < g + GROUP; l++) { uint32_t p = wb + l; cand[l] = (int)p <= pmax ? s0 + table[slot(in + p)] : 0xFFFFFFFF; }
            for (int l = g; l < g + GROUP; l++) { uint32_t p = wb + l; if ((int)p <= pmax) table[slot(in + p)] = (uint16_t)(p - s0); }
        }
        for (int l = 0; l < 64 && NEAR; l++) {
            uint32_t p = wb + l;
            if ((int)p > pmax) continue;
            for (int d = 1; d < 16; d++)
                if (((NEAR >> d) & 1) && ((l & 15) >= d || d <= 4) && p >= (uint32_t)d && rd32(in + p) == rd32(in + p - d)) { cand[l] = p - d; break; }
        }
        for (int l = 0; l < 64; l++) {
            uint32_t p = wb + l, c = cand[l]; mlen[l] = 0; bk[l] = 0;
            if (c == 0xFFFFFFFF || c >= p || c < 4) continue;
            int k = 0; while (k < FWD && in[p + k] == in[c + k]) k++;
            int lim = mend - (int)p; if (k > lim) k = lim;
            if (k < 0) k = 0;
            mlen[l] = k;
            int b = 0; while (b < BACK && p >= (uint32_t)b + 1 && in[p - 1 - b] == in[c - 1 - b]) b++;
            bk[l] = b;
        }
        for (int l = 0; l < 64; l++) {
            uint32_t p = wb + l;
            if (p < sp) continue;
            int m = mlen[l];
            if (m < 4) continue;
            int skip = 0;
            for (int d = 1; d <= LAZY; d++) if (l + d < 64 && mlen[l + d] >= m + d) skip = 1;
            if (skip) continue;
            uint32_t c = cand[l]; uint32_t ml = m;
            if (m == FWD) { while ((int)(p + ml) < mend && in[p + ml] == in[c + ml]) ml++; n_ext++; }
            uint32_t b = bk[l]; if (b > p - sp) b = p - sp;
            uint32_t start = p - b;
            if (sp < wb) { if (sp + 64 >= wb) n_prevwin++; else n_older++; }
            o = emit(o, in + sp, start - sp, p - c, ml + b); n_seq_tot++;
            sp = p + ml;
        }
    }
    *tail = s1 - sp;
    return (uint32_t)(o - out);
}

/* the whole block; cap: bytes available in out (0 returned when the block does not fit, like LZ4_compress_default) */
int lz4p_model_encode(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap)
{
    uint32_t nseg = (n + SEG - 1) / SEG; if (nseg > 64) nseg = 64;
    uint8_t* scratch = malloc(SEG + SEG / 255 + 64);
    uint8_t* tmp = malloc((size_t)n + n / 255 + 64 + 4096);
    uint8_t* o = tmp;
    uint32_t carry = 0;
    for (uint32_t k = 0; k < nseg; k++) {
        uint32_t s0 = k * SEG, s1 = s0 + SEG < n ? s0 + SEG : n, tail;
        uint32_t len = encode_segment(in, n, s0, s1, scratch, &tail);
        if (len) {
            /* first sequence: its literal length grows by the carry; the carried literals come from the input */
            uint32_t tok = scratch[0], ll0 = tok >> 4, q = 1;
            if (ll0 == 15) { uint32_t bb; do { bb = scratch[q++]; ll0 += bb; } while (bb == 255); }
            uint32_t ll = ll0 + carry;
            uint8_t* t = o++;
            if (ll >= 15) { *t = (uint8_t)(0xF0 | (tok & 15)); o = put_len(o, ll - 15); } else *t = (uint8_t)((ll << 4) | (tok & 15));
            memcpy(o, in + s0 - carry, carry); o += carry;
            memcpy(o, scratch + q, len - q); o += len - q;
            carry = tail;
        } else carry += tail;
    }
    if (nseg * (uint32_t)SEG < n) carry += n - nseg * SEG;          /* what lies beyond 64 segments goes out as literals */
    o = emit(o, in + n - carry, carry, 0, 0);
    size_t total = (size_t)(o - tmp);
    int r = 0;
    if (total <= cap) { memcpy(out, tmp, total); r = (int)total; }
    free(scratch); free(tmp);
    return r;
}
void lz4p_model_set(int nent, int group, int back, int lazy, int fwd, unsigned near) { NENT = nent; GROUP = group; BACK = back; LAZY = lazy; FWD = fwd; NEAR = near; }

#ifndef LZ4P_NO_MAIN
#include "oracle.h"
void corpus_fill(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);
void corpus_fill_logs(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);
#define B (4u << 20)
int main(int argc, char** argv)
{
    int logs = 0, verbose = 0; uint32_t bsize = B;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "logs")) logs = 1;
        else if (!strcmp(argv[i], "-v")) verbose = 1;
        else if (!strncmp(argv[i], "nent=", 5)) NENT = atoi(argv[i] + 5);
        else if (!strncmp(argv[i], "group=", 6)) GROUP = atoi(argv[i] + 6);
        else if (!strncmp(argv[i], "back=", 5)) BACK = atoi(argv[i] + 5);
        else if (!strncmp(argv[i], "lazy=", 5)) LAZY = atoi(argv[i] + 5);
        else if (!strncmp(argv[i], "fwd=", 4)) FWD = atoi(argv[i] + 4);
        else if (!strncmp(argv[i], "hash=", 5)) HASHV = atoi(argv[i] + 5);
        else if (!strncmp(argv[i], "near=", 5)) NEAR = strtoul(argv[i] + 5, 0, 0);
        else if (!strncmp(argv[i], "bsize=", 6)) bsize = atoi(argv[i] + 6);
    }
    uint8_t* in = malloc(B + 64); uint8_t* c = malloc(B + B / 255 + 64); uint8_t* c2 = malloc(B + B / 255 + 4096); uint8_t* back = malloc(B);
    long tot_ref = 0, tot_par = 0, bad = 0;
    for (int blk = 0; blk < 48; blk++) {
        (logs ? corpus_fill_logs : corpus_fill)(in, B, 0x4D43, blk);
        int cs = orc_lz4_compress_fast(in, c, bsize, bsize + bsize / 255 + 64);
        int cp = lz4p_model_encode(in, bsize, c2, bsize + bsize / 255 + 4096);
        int r = orc_lz4_decompress_safe(c2, back, cp, bsize);
        if (r != (int)bsize || memcmp(back, in, bsize)) { bad++; printf("blk %d: BAD r=%d\n", blk, r); }
        if (cs <= 0 || cs >= (int)bsize) cs = bsize;
        if (cp >= (int)bsize) cp = bsize;
        if (verbose) printf("blk %2d ref %8d par %8d  %+6.2f %%\n", blk, cs, cp, 100.0 * (cp - cs) / cs);
        tot_ref += cs; tot_par += cp;
    }
    printf("ref %ld par %ld  ratio ref %.4f par %.4f (%+.2f %%)  seqs/block %ld ext %.1f %% lit from prev window %.1f %% older %.2f %% bad %ld\n", tot_ref, tot_par,
           48.0 * bsize / tot_ref, 48.0 * bsize / tot_par, 100.0 * ((double)tot_ref / tot_par - 1), n_seq_tot / 48,
           100.0 * n_ext / n_seq_tot, 100.0 * n_prevwin / n_seq_tot, 100.0 * n_older / n_seq_tot, bad);
    return bad != 0;
}
#endif
