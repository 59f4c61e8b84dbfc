/*
 * tools/model/tile_decode_model.c - executable model of the tile LZ4 decoder (4mc_amd/csrc/lz4_tile.hip), thread by thread.
 *
 * Written BEFORE the kernels.  Two steps:
 *   WALK  (one wave per block, one lane per stream segment): as the segment-parallel walk, but what it leaves is ONE BIT PER STREAM
 *         BYTE - "a token of the true chain starts here" - instead of a record per sequence: a lane writes the 32-bit words of its
 *         own segment (seglen is a multiple of 32), a chain that enters the segment somewhere else rewrites the words from the
 *         segment's start until it falls onto a bit of the chain already there.
 *   EXEC  (one workgroup of 512 threads per block, the 64 KiB LZ4 window resident in LDS as a ring): per chunk of 2 KiB of stream
 *         the tokens are compacted out of the bitmap, decoded one per thread from the staged stream, placed by a prefix sum; the
 *         chunk's output is produced in tiles of <= 4096 bytes, ONE THREAD PER OUTPUT BYTE: every sequence marks where its literal
 *         part and its match part begin, a max-scan gives every byte its (sequence, part), literal bytes come from the staged
 *         stream, match bytes whose source lies in front of the tile from the ring (all reads of old ring contents happen before
 *         the tile's first write), match bytes whose source lies inside the tile keep a 16-bit POINTER to it and are resolved by
 *         chasing pointers until a byte that is final - no order between the threads is needed for that, and a byte that has been
 *         resolved is final for everyone behind it.  Sequences of any length are simply clipped to the tile (no escapes).
 * (The kernels kept the steps and changed their shape on the way: the walk runs inside the executor's workgroup with one segment per
 *  thread, a chunk is 1536 stream bytes and at most 384 sequences, a tile at most 4088 bytes, a thread owns eight consecutive bytes,
 *  a final byte points to itself and the chase writes shortened pointers back - DESIGN.md section 3, "K1t".)
 * Checked against the oracle on the corpus, on edge inputs and on damaged streams.  A "workgroup" is a loop over 512 threads per
 * phase (a phase ends where the kernel has a barrier); the pointer chase runs its threads in a shuffled order.
 * Design aid / test infrastructure only (links the oracle); not product.
 *
 *   gcc -O2 -o /tmp/tile_decode_model tools/model/tile_decode_model.c tools/corpus.c -Ioracle oracle/liboracle.so -Wl,-rpath,$PWD/oracle
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "oracle.h"
void corpus_fill(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);
void corpus_fill_logs(uint8_t* dst, size_t len, uint64_t seed, uint64_t first_block);

#define NL        64            /* lanes of the walk = segments */
#define MARGIN    64            /* tokens whose bytes end beyond csize - MARGIN belong to the tail (exact walker) */
#define OMARGIN   128           /* sequences whose output ends beyond cap - OMARGIN belong to the tail */
#define MINSEG    1024
#define kRetry    (-1000000003)

#define NT        512           /* threads of the executor workgroup */
#define TCAP      4096          /* output bytes per tile */
#define SC        2048          /* stream bytes whose tokens one chunk takes (64 bitmap words) */
#define STG       (SC + 32 + 272)   /* staged stream bytes per chunk */
#define RING      65536
#define FINAL     0xFFFFu
#define SZCLAMP   ((4u << 20) + 1u)

typedef struct { uint32_t pos, ll, ml, off, next, lsrc; int stop; } Tok;

/* one token at p; stop = it (or its bytes) reaches beyond limit = csize - MARGIN: the chain halts AT p */
static Tok decode_tok(const uint8_t* s, int csize, uint32_t p)
{
    Tok t; memset(&t, 0, sizeof t); t.pos = p;
    const uint32_t limit = (uint32_t)(csize - MARGIN);
    if (p >= limit) { t.stop = 1; return t; }
    uint32_t tok = s[p], q = p + 1, ll = tok >> 4, mn = tok & 15;
    if (ll == 15) for (;;) { if (q >= limit) { t.stop = 1; return t; } uint32_t b = s[q++]; ll += b; if (b != 255) break; if (ll > (1u << 23)) { t.stop = 1; return t; } }
    uint32_t mo = q + ll;
    if (mo + 2 > limit) { t.stop = 1; return t; }
    t.off = s[mo] | (s[mo + 1] << 8);
    uint32_t q2 = mo + 2, ml = mn + 4;
    if (mn == 15) for (;;) { if (q2 >= limit) { t.stop = 1; return t; } uint32_t b = s[q2++]; ml += b; if (b != 255) break; }
    if (q2 > limit) { t.stop = 1; return t; }
    t.ll = ll; t.ml = ml; t.next = q2; t.lsrc = q;
    return t;
}

/* ------------------------------------------------------------------------------------------------ walk: token bitmap */
typedef struct {
    uint32_t seglen, nseg;
    uint32_t* bm; uint32_t nwords;
    uint32_t exitp[NL], entry[NL]; int tail[NL];
    uint32_t tail_ip;
    long stat_fix_hops, stat_rounds;
} Walk;

/* lane j (re)writes the words of its segment for the chain that enters at e.  first: nothing is there yet (phase 1, e = segment
   start): the chain is walked to the segment's end.  Otherwise the walk ends where the chain falls onto a bit that is already there. */
static void lane_chain(Walk* W, const uint8_t* s, int csize, uint32_t j, uint32_t e, int first)
{
    const uint32_t sj = j * W->seglen;
    const uint32_t seg_end = (j + 1 == W->nseg) ? 0xFFFFFFFFu : (j + 1) * W->seglen;
    const uint32_t wend = (j + 1 == W->nseg) ? W->nwords : (seg_end >> 5);      /* the lane's words: [sj >> 5, wend) */
    uint32_t q = e, curw = sj >> 5, acc = 0, old = first ? 0 : W->bm[curw];
    for (;;) {
        if (q >= seg_end) { W->exitp[j] = q; W->tail[j] = 0; break; }
        const uint32_t w = q >> 5;
        if (w != curw) {
            W->bm[curw] = acc; for (uint32_t x = curw + 1; x < w; x++) W->bm[x] = 0;
            curw = w; acc = 0; old = first ? 0 : W->bm[w];
        }
        if ((old >> (q & 31)) & 1) {                                   /* merged: the bits at and above q stay */
            W->bm[curw] = acc | (old & ~((1u << (q & 31)) - 1u));
            W->entry[j] = e;
            return;
        }
        Tok t = decode_tok(s, csize, q);
        if (t.stop) { W->exitp[j] = q; W->tail[j] = 1; break; }
        acc |= 1u << (q & 31);
        q = t.next; if (!first) W->stat_fix_hops++;
    }
    W->bm[curw] = acc; for (uint32_t x = curw + 1; x < wend; x++) W->bm[x] = 0;
    W->entry[j] = e;
}
static void walk_block(Walk* W, const uint8_t* s, int csize)
{
    const uint32_t limit = (uint32_t)(csize - MARGIN);
    uint32_t nseg = limit / MINSEG; if (nseg < 1) nseg = 1; if (nseg > NL) nseg = NL;
    W->nseg = nseg; W->seglen = ((limit + nseg - 1) / nseg + 31) & ~31u;
    W->nwords = ((uint32_t)csize + 31) >> 5;
    /* (segments that begin at or beyond the limit walk nothing) */
    for (uint32_t j = 0; j < nseg; j++) lane_chain(W, s, csize, j, j * W->seglen, 1);            /* phase 1: all lanes at once */
    uint32_t ex0[NL]; int tl0[NL]; for (uint32_t j = 0; j < nseg; j++) { ex0[j] = W->exitp[j]; tl0[j] = W->tail[j]; }
    for (uint32_t j = 1; j < nseg; j++) {                                                          /* phase 2: optimistic, all lanes at once */
        const uint32_t pe = ex0[j - 1];
        if (tl0[j - 1]) continue;
        uint32_t sj = pe / W->seglen; if (sj > nseg - 1) sj = nseg - 1;
        if (sj != j || pe == j * W->seglen) continue;
        lane_chain(W, s, csize, j, pe, 0);
    }
    uint32_t cur = 0;                                                                              /* phase 3: the true path */
    for (;;) {
        if (W->tail[cur]) { W->tail_ip = W->exitp[cur]; break; }
        const uint32_t e = W->exitp[cur];
        uint32_t j = e / W->seglen; if (j > nseg - 1) j = nseg - 1;
        for (uint32_t d = cur + 1; d < j; d++)                                                     /* segments the chain jumps over: no tokens */
            for (uint32_t x = (d * W->seglen) >> 5; x < (((d + 1) * W->seglen) >> 5) && x < W->nwords; x++) W->bm[x] = 0;
        if (W->entry[j] != e) { W->stat_rounds++; lane_chain(W, s, csize, j, e, 0); }
        cur = j;
    }
}

/* ------------------------------------------------------------------------------------------------ exec: tiles */
typedef struct {
    uint8_t  ring[RING];
    uint16_t code[TCAP];                /* per tile byte: (sequence + 1) << 2 | overlap << 1 | part; then: pointer / FINAL */
    uint8_t  stage[STG];
    uint32_t DL[NT]; uint16_t OFF[NT], MSC[NT];
    uint16_t toks[NT];                  /* (kernel: the same LDS as MSC) */
    long chunks, tiles, near_bytes, hops, maxhops, lit_global;
    long depth_hist[65], depth_sum, near_seq, seqs;
} Exec;

static uint32_t perm[NT];

static int exec_block(Exec* X, const Walk* W, const uint8_t* s, int csize, uint8_t* dst, int cap, uint32_t* res_ip, uint32_t* res_op)
{
    const uint32_t tail_ip = W->tail_ip, olimit = (uint32_t)cap - OMARGIN;
    const uint32_t A = (uint32_t)((uintptr_t)dst & 0xFFFFu);          /* ring index of output position P: (A + P) & 0xFFFF: congruent to the address mod 16 */
    uint32_t ip = 0, opos = 0, flushed = 0;
    int cut = 0;
    *res_ip = tail_ip;
    while (ip < tail_ip && !cut) {
        X->chunks++;
        /* ---- stage + bitmap words of [ip, cend) */
        const uint32_t sbase = ip & ~15u;
        for (uint32_t k = 0; k < STG; k++) X->stage[k] = sbase + k < (uint32_t)csize ? s[sbase + k] : 0;
        const uint32_t w0 = ip >> 5;
        uint32_t cend = (w0 << 5) + SC; if (cend > tail_ip) cend = tail_ip;
        uint32_t word[64], cnt[64];
        for (uint32_t l = 0; l < 64; l++) {
            uint32_t w = (w0 + l) < W->nwords ? W->bm[w0 + l] : 0;
            const uint32_t lo = (w0 + l) << 5;
            if (lo < ip) w &= ~((1u << (ip - lo)) - 1u);                              /* ip - lo < 32: only the first word */
            if (lo >= cend) w = 0; else if (cend - lo < 32) w &= (1u << (cend - lo)) - 1u;
            word[l] = w; cnt[l] = (uint32_t)__builtin_popcount(w);
        }
        uint32_t ntok = 0, next_ip = cend;
        for (uint32_t l = 0; l < 64; l++) {                                          /* wave 0: lane l, exclusive scan of cnt */
            uint32_t idx = ntok, w = word[l];
            while (w) {
                const uint32_t bit = (uint32_t)__builtin_ctz(w); w &= w - 1;
                const uint32_t p = ((w0 + l) << 5) + bit;
                if (idx < NT) X->toks[idx] = (uint16_t)(p - sbase);
                else if (idx == NT) next_ip = p;
                idx++;
            }
            ntok += cnt[l];
        }
        uint32_t n = ntok < NT ? ntok : NT;
        if (n == 0) { ip = next_ip; continue; }
        /* ---- one sequence per thread */
        uint32_t ll[NT], ml[NT], off[NT], lsrc[NT], sz[NT], incl[NT], tpos[NT];
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t p = sbase + X->toks[i];
            /* (kernel: bytes from the stage, from memory beyond it) */
            const Tok t = decode_tok(s, csize, p);
            if (t.stop) { printf("model: token at %u in the bitmap stops\n", p); return kRetry; }
            tpos[i] = p; ll[i] = t.ll; ml[i] = t.ml; off[i] = t.off; lsrc[i] = t.lsrc;
            const uint64_t z = (uint64_t)t.ll + t.ml; sz[i] = z > SZCLAMP ? SZCLAMP : (uint32_t)z;
        }
        { uint32_t run = 0; for (uint32_t i = 0; i < n; i++) { run += sz[i]; incl[i] = run; } }
        uint32_t nfit = n, nbad = n;
        for (uint32_t i = 0; i < n; i++) {
            if (nfit == n && !(opos + incl[i] <= olimit && sz[i] < SZCLAMP)) nfit = i;
            const uint32_t mabs = opos + (incl[i] - sz[i]) + ll[i];
            if (nbad == n && (off[i] == 0 || off[i] > mabs)) nbad = i;
        }
        if (nbad < nfit) return kRetry;
        if (nfit < n) { cut = 1; *res_ip = tpos[nfit]; n = nfit; }
        const uint32_t total = n ? incl[n - 1] : 0;
        for (uint32_t i = 0; i < n; i++) { X->DL[i] = lsrc[i] - (incl[i] - sz[i]); X->OFF[i] = (uint16_t)off[i]; }
        /* ---- tiles */
        for (uint32_t R0 = 0; R0 < total; R0 += TCAP) {
            const uint32_t T = total - R0 < TCAP ? total - R0 : TCAP;
            X->tiles++;
            memset(X->code, 0, sizeof X->code);
            for (uint32_t i = 0; i < n; i++) {                                       /* marks */
                const uint32_t outl = incl[i] - sz[i], ms = outl + ll[i], me = outl + sz[i];
                if (ll[i] && outl < R0 + T && ms > R0) X->code[(outl > R0 ? outl : R0) - R0] = (uint16_t)(((i + 1) << 2) | 0);
                if (ms < R0 + T && me > R0) {
                    const uint32_t m = (ms > R0 ? ms : R0) - R0;
                    X->code[m] = (uint16_t)(((i + 1) << 2) | 1 | (off[i] < ml[i] ? 2 : 0));
                    X->MSC[i] = (uint16_t)m;
                }
            }
            { uint16_t run = 0; for (uint32_t r = 0; r < T; r++) { if (X->code[r] > run) run = X->code[r]; X->code[r] = run; } }   /* max-scan */
            /* pass 1, reads: sources of everything that is final */
            uint8_t val[TCAP]; uint16_t ptr[TCAP];
            for (uint32_t r = 0; r < T; r++) {
                const uint32_t c = X->code[r], i = (c >> 2) - 1;
                if (c == 0) { printf("model: byte %u of a tile without a sequence\n", r); return kRetry; }
                if (!(c & 1)) {
                    const uint32_t sp = X->DL[i] + R0 + r;
                    if (sp - sbase < STG) val[r] = X->stage[sp - sbase]; else { val[r] = s[sp]; X->lit_global++; }
                    ptr[r] = FINAL;
                } else {
                    const uint32_t o = X->OFF[i];
                    int32_t src;
                    if (c & 2) { const uint32_t m = X->MSC[i], k = r - m; src = (int32_t)m - (int32_t)o + (int32_t)(k % o); }
                    else src = (int32_t)r - (int32_t)o;
                    if (src < 0) { val[r] = X->ring[(A + opos + (uint32_t)src) & 0xFFFFu]; ptr[r] = FINAL; }
                    else ptr[r] = (uint16_t)src;
                }
            }
            /* pass 1, writes */
            for (uint32_t r = 0; r < T; r++) {
                if (ptr[r] == FINAL) X->ring[(A + opos + r) & 0xFFFFu] = val[r];
                X->code[r] = ptr[r];
            }
            {   /* statistics: dependency depth of the tile (bytes in ascending order: a source is always below) */
                static uint16_t dep[TCAP]; uint32_t mx = 0;
                for (uint32_t r = 0; r < T; r++) { dep[r] = ptr[r] == FINAL ? 0 : (uint16_t)(dep[ptr[r]] + 1); if (dep[r] > mx) mx = dep[r]; }
                X->depth_hist[mx > 64 ? 64 : mx]++; X->depth_sum += mx;
            }
            /* pass 2: bytes with a source inside the tile, in any order */
            {
                static uint32_t order[TCAP];
                for (uint32_t r = 0; r < T; r++) order[r] = r;
                if (X->tiles & 1) { for (uint32_t r = 0; r + 1 < T; r++) { const uint32_t y = r + (uint32_t)(rand() % (T - r)); const uint32_t t = order[r]; order[r] = order[y]; order[y] = t; } }
                else for (uint32_t b0 = 0; b0 < T; b0 += NT) { const uint32_t m = T - b0 < NT ? T - b0 : NT; for (uint32_t r = 0; r + 1 < m; r++) { const uint32_t y = r + (uint32_t)(rand() % (m - r)); const uint32_t t = order[b0 + r]; order[b0 + r] = order[b0 + y]; order[b0 + y] = t; } }
                for (uint32_t x = 0; x < T; x++) {
                    const uint32_t r = order[x];
                    if (ptr[r] == FINAL) continue;
                    uint32_t p = ptr[r]; long h = 1;
                    for (;;) { const uint32_t q = X->code[p]; if (q == FINAL) break; p = q; h++; }
                    X->near_bytes++; X->hops += h; if (h > X->maxhops) X->maxhops = h;
                    X->ring[(A + opos + r) & 0xFFFFu] = X->ring[(A + opos + p) & 0xFFFFu];
                    X->code[r] = FINAL;
                }
            }
            /* flush: whole 16-byte pieces by ADDRESS; the piece the tile ends in waits for the next tile */
            const uint32_t E = opos + T;
            if (((uintptr_t)dst + flushed) & 15) {                                       /* the block's first bytes up to an aligned address */
                uint32_t h = flushed + 16 - (uint32_t)(((uintptr_t)dst + flushed) & 15); if (h > E) h = E;
                for (uint32_t k = flushed; k < h; k++) dst[k] = X->ring[(A + k) & 0xFFFFu];
                flushed = h;
            }
            while ((((uintptr_t)dst + flushed) & 15) == 0 && flushed + 16 <= E) {
                for (int k = 0; k < 16; k++) dst[flushed + k] = X->ring[(A + flushed + k) & 0xFFFFu];
                flushed += 16;
            }
            opos = E;
        }
        ip = next_ip;
    }
    for (uint32_t k = flushed; k < opos; k++) dst[k] = X->ring[(A + k) & 0xFFFFu];        /* what the last tile left */
    *res_op = opos;
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
    /* the bitmap must be exactly the true chain below tail_ip */
    {
        uint32_t p = 0, nb = 0;
        for (;;) { Tok t = decode_tok(s, csize, p); if (t.stop) break; if (!((W.bm[p >> 5] >> (p & 31)) & 1)) { printf("model: true token %u missing\n", p); return -7; } nb++; p = t.next; }
        if (p != W.tail_ip) { printf("model: tail_ip %u, true chain stops at %u\n", W.tail_ip, p); return -7; }
        uint32_t have = 0;
        for (uint32_t q = 0; q < W.tail_ip; q++) have += (W.bm[q >> 5] >> (q & 31)) & 1;
        if (have != nb) { printf("model: %u bits below tail_ip, %u true tokens\n", have, nb); return -7; }
    }
    return exec_block(&X, &W, s, csize, dst, cap, rip, rop);
}

static uint64_t rs = 0x1234567;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

int main(int argc, char** argv)
{
    const uint32_t B = 4u << 20;
    int nfuzz = argc > 1 ? atoi(argv[1]) : 200;
    uint8_t* in = malloc(B); uint8_t* c = malloc(B + B / 255 + 64 + 64); uint8_t* ref = malloc(B + 64);
    gbuf = malloc(B + 4096 + 64);
    W.bm = malloc(4 * ((B + B / 255 + 128) / 32 + 8));
    for (uint32_t i = 0; i < NT; i++) perm[i] = i;
    int bad = 0;
    /* 1. corpus blocks, whole */
    for (int pass = 0; pass < 2; pass++) for (int blk = 0; blk < (pass ? 8 : 24); blk++) {
        (pass ? corpus_fill_logs : corpus_fill)(in, B, 0x4D43, blk);
        int cs = orc_lz4_compress_fast(in, c, B, B - 1);
        if (cs <= 0) continue;
        uint8_t* out = gbuf + 2048 + (blk % 16);
        memset(gbuf, 0xA5, B + 4096 + 64); memset(&X, 0, sizeof X); W.stat_fix_hops = W.stat_rounds = 0;
        memset(W.bm, 0xFF, 4 * ((B + B / 255 + 128) / 32 + 8));
        uint32_t rip = 0, rop = 0;
        int r = model_decode(c, cs, out, B, &rip, &rop);
        int fin = r == 1 ? tail_decode(c, cs, out, B, rip, rop) : r;
        int ok = fin == (int)B && !memcmp(out, in, B);
        for (int i = 0; i < 2048; i++) if (gbuf[i] != 0xA5 || out[B + i] != 0xA5) ok = 0;
        printf("%s blk %2d csize %7d: %s  fixhops %ld rounds %ld | chunks %ld tiles %ld near %.1f%% hops/near %.2f max %ld lit beyond stage %ld tail at ip %u (csize-%d) op %u\n",
               pass ? "logs" : "smix", blk, cs, ok ? "OK" : "MISMATCH", W.stat_fix_hops, W.stat_rounds,
               X.chunks, X.tiles, 100.0 * X.near_bytes / B, X.near_bytes ? (double)X.hops / X.near_bytes : 0, X.maxhops, X.lit_global, rip, cs - (int)rip, rop);
        if (!ok) { bad++; for (uint32_t i = 0; i < B; i++) if (out[i] != in[i]) { printf("   first diff at %u (decoded %d)\n", i, fin); break; } }
        { printf("      tile depth: mean %.1f; tiles with max depth 0..7: ", (double)X.depth_sum / X.tiles); for (int d = 0; d < 8; d++) printf("%ld ", X.depth_hist[d]); long r8 = 0, r16 = 0, r32 = 0; for (int d = 8; d < 65; d++) { if (d < 16) r8 += X.depth_hist[d]; else if (d < 32) r16 += X.depth_hist[d]; else r32 += X.depth_hist[d]; } printf("| 8-15: %ld 16-31: %ld 32+: %ld\n", r8, r16, r32); }
    }
    /* 2. small / odd sizes and damaged streams against the oracle's verdict */
    long accepted = 0, retried = 0, checked = 0;
    for (int it = 0; it < nfuzz; it++) {
        const int blk = rnd() % 48; uint32_t n = 300 + rnd() % (it % 3 == 0 ? 400000 : 20000);
        corpus_fill(in, B, 0x4D43, blk);
        const uint32_t o = rnd() % (B - n);
        if (it % 5 == 0) memset(in + o + n / 3, it & 255, n / 4);                        /* long runs */
        if (it % 7 == 0) for (uint32_t i = n / 2; i < n / 2 + n / 5; i++) in[o + i] = (uint8_t)rnd();   /* long literals */
        if (it % 9 == 0) for (uint32_t i = n / 8; i < n / 2; i++) in[o + i] = in[o + i - 1 - (it % 7)];   /* short periods */
        int cs = orc_lz4_compress_fast(in + o, c, (int)n, (int)n + 64);
        if (cs <= 0) continue;
        int nm = it % 2 ? 1 + rnd() % 3 : 0;
        for (int m = 0; m < nm; m++) { uint32_t at = rnd() % cs; c[at] = (rnd() & 1) ? (uint8_t)rnd() : (c[at] ^ (1u << (rnd() & 7))); }
        int cap = (int)n - (it % 11 == 0 ? (int)(rnd() % 40) : 0) + (it % 13 == 0 ? 17 : 0);
        memset(ref, 0, n + 64);
        int rr = orc_lz4_decompress_safe(c, ref, cs, cap);
        uint8_t* out = gbuf + 2048 + (it % 16);
        memset(gbuf, 0xA5, B + 4096 + 64); memset(&X, 0, sizeof X);
        memset(W.bm, 0xFF, 4 * ((B + B / 255 + 128) / 32 + 8));
        uint32_t rip = 0, rop = 0;
        int r = model_decode(c, cs, out, cap, &rip, &rop);
        checked++;
        int ok = r != -7;
        for (int i = 0; i < 2048; i++) if (gbuf[i] != 0xA5) ok = 0;
        for (int i = 0; i < 2048; i++) if (out[cap + i] != 0xA5) ok = 0;               /* nothing beyond the capacity */
        if (r == 1) {
            accepted++;
            /* the oracle must not have failed before the hand-over point, and the bytes so far must be its bytes */
            if (rr < 0 && rr != INT32_MIN && (uint32_t)(-rr - 1) < rip) ok = 0;
            if (rr != INT32_MIN && memcmp(out, ref, rop)) ok = 0;
            if (rr >= 0 && (int)rop > rr) ok = 0;
            if (rr >= 0) { int fin = tail_decode(c, cs, out, cap, rip, rop); if (nm == 0 && (fin != rr || memcmp(out, ref, rr))) ok = 0; }
        } else retried++;
        if (!ok) { bad++; printf("fuzz %d: n %u cs %d cap %d muts %d: oracle %d model %d rip %u rop %u  BAD\n", it, n, cs, cap, nm, rr, r, rip, rop); }
    }
    printf("fuzz: %ld streams, %ld through the fast path to the tail, %ld handed back; %d bad\n", checked, accepted, retried, bad);
    return bad != 0;
}
