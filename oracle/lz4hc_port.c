/*
 * oracle/lz4hc_port.c — scalar restatement of LZ4_compress_HC for the hash-chain levels 4mc uses
 * (4mc High = level 4, 4mc Ultra = level 8).  TEST INFRASTRUCTURE, NOT PRODUCT (see oracle.h).
 *
 *   entry        LZ4_compress_HC            native/lz4/lz4hc.c:958-973 -> :939-949 -> :800-861
 *   parse        LZ4HC_compress_hashChain   native/lz4/lz4hc.c:553-788   (lazy 3-match arbitration)
 *   search       LZ4HC_InsertAndGetWiderMatch :239-447 with patternAnalysis = 0 (nbSearches <= 128,
 *                :565) and chainSwap = 0 (:461,:603,:648), no dictionary
 *   tables       LZ4HC_Insert :120-141, hash :82, table init :98-117 (indices start at 64 KiB)
 *   emit         LZ4HC_encodeSequence :467-548, last literals :735-762
 *
 * Positions are plain offsets into the block; a table index is position + 65536 exactly as in the
 * reference (LZ4HC_init_internal), so an all-zero hash table means "no candidate".
 * Parity: pinned — byte-identical to oracle/_ref (LZ4_compress_HC, levels 4 and 8, capacities
 * bound / n-1) on the corpus and edge inputs (tests/test_oracle_golden.py) and to the per-block
 * manifests of the reference CLI at `4mc -3` / `-4` (tests/golden/corpus_manifest.json).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define HASH_LOG   15
#define MAXD       65536
#define MAX_DIST   65535
#define MINMATCH   4
#define MFLIMIT    12
#define LASTLIT    5
#define OPTIMAL_ML 18
#define IDX0       65536u                 /* index of position 0 */

typedef struct {
    uint32_t hash[1 << HASH_LOG];
    uint16_t chain[MAXD];
    uint32_t next_to_update;              /* index */
    const uint8_t* src;
} hc_t;

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static uint32_t hc_hash(const uint8_t* p) { return (rd32(p) * 2654435761u) >> (32 - HASH_LOG); }

static void hc_insert(hc_t* c, uint32_t ip /* position */)
{
    const uint32_t target = ip + IDX0;
    uint32_t idx = c->next_to_update;
    while (idx < target) {
        const uint32_t h = hc_hash(c->src + (idx - IDX0));
        uint32_t delta = idx - c->hash[h];
        if (delta > MAX_DIST) delta = MAX_DIST;
        c->chain[idx & 0xFFFF] = (uint16_t)delta;
        c->hash[h] = idx;
        idx++;
    }
    c->next_to_update = target;
}

static int count_fwd(const uint8_t* s, uint32_t a, uint32_t b, uint32_t lim)   /* a > b, stops at a == lim */
{
    const uint32_t a0 = a;
    while (a < lim && s[a] == s[b]) { a++; b++; }
    return (int)(a - a0);
}

/* best match for `ip`, allowed to start as early as `low` (lookback) and end by `high`; only
 * matches longer than `longest` count.  Returns the new longest; *mpos / *spos = match / start. */
static int hc_wider(hc_t* c, uint32_t ip, uint32_t low, uint32_t high, int longest,
                    uint32_t* mpos, uint32_t* spos, int attempts)
{
    const uint8_t* s = c->src;
    const uint32_t ip_idx = ip + IDX0;
    const uint32_t lowest = (IDX0 + MAXD > ip_idx) ? IDX0 : ip_idx - MAX_DIST;
    const int lookback = (int)(ip - low);
    const uint32_t pattern = rd32(s + ip);
    uint32_t mi;
    hc_insert(c, ip);
    mi = c->hash[hc_hash(s + ip)];
    while (mi >= lowest && attempts > 0) {
        const uint32_t m = mi - IDX0;
        attempts--;
        if (rd16(s + low + longest - 1) == rd16(s + m - lookback + longest - 1) && rd32(s + m) == pattern) {
            int back = 0, ml;
            if (lookback) {
                int min = (int)low - (int)ip;                      /* max(iMin - ip, mMin - match) */
                if (-(int)m > min) min = -(int)m;
                while (back > min && s[ip + back - 1] == s[m + back - 1]) back--;
            }
            ml = MINMATCH + count_fwd(s, ip + MINMATCH, m + MINMATCH, high) - back;
            if (ml > longest) { longest = ml; *mpos = m + back; *spos = ip + back; }
        }
        mi -= c->chain[mi & 0xFFFF];
    }
    return longest;
}

/* token + literals + offset + match length; returns 1 when `limited` and the output would overflow */
static int hc_emit(const uint8_t* src, uint32_t* ip, uint8_t** op, uint32_t* anchor, int ml, uint32_t match,
                   int limited, uint8_t* oend)
{
    size_t len = *ip - *anchor;
    uint8_t* token = (*op)++;
    if (limited && (*op + len / 255 + len + (2 + 1 + LASTLIT) > oend)) return 1;
    if (len >= 15) {
        size_t l = len - 15;
        *token = 0xF0;
        for (; l >= 255; l -= 255) *(*op)++ = 255;
        *(*op)++ = (uint8_t)l;
    } else *token = (uint8_t)(len << 4);
    memcpy(*op, src + *anchor, len); *op += len;
    (*op)[0] = (uint8_t)(*ip - match); (*op)[1] = (uint8_t)((*ip - match) >> 8); *op += 2;
    len = (size_t)ml - MINMATCH;
    if (limited && (*op + len / 255 + (1 + LASTLIT) > oend)) return 1;
    if (len >= 15) {
        *token += 15; len -= 15;
        for (; len >= 255; len -= 255) *(*op)++ = 255;
        *(*op)++ = (uint8_t)len;
    } else *token += (uint8_t)len;
    *ip += (uint32_t)ml;
    *anchor = *ip;
    return 0;
}

int orc_lz4hc_compress(const uint8_t* src, uint8_t* dst, int n, int cap, int level)
{
    static const int searches[10] = {2, 2, 2, 4, 8, 16, 32, 64, 128, 256};
    hc_t* c;
    int attempts, limited, result = 0;
    uint32_t ip = 0, anchor = 0;
    uint8_t* op = dst;
    uint8_t* const oend = dst + cap;
    int ml, ml2, ml3, ml0;
    uint32_t ref = 0, start2 = 0, ref2 = 0, start3 = 0, ref3 = 0, start0, ref0;

    if (level < 1 || level > 8) return -2;              /* levels 9+ need pattern analysis / the optimal parser */
    if ((unsigned)n > 0x7E000000u) return 0;
    attempts = searches[level];
    limited = cap < orc_lz4_compress_bound(n);
    c = (hc_t*)malloc(sizeof *c);
    memset(c->hash, 0, sizeof c->hash);
    memset(c->chain, 0xFF, sizeof c->chain);
    c->next_to_update = IDX0; c->src = src;

    if (n >= MFLIMIT + 1) {
        const uint32_t mflimit = (uint32_t)n - MFLIMIT, matchlimit = (uint32_t)n - LASTLIT;
        while (ip <= mflimit) {
            ml = hc_wider(c, ip, ip, matchlimit, MINMATCH - 1, &ref, &start0 /*unused*/, attempts);
            if (ml < MINMATCH) { ip++; continue; }
            start0 = ip; ref0 = ref; ml0 = ml;
        search2:
            if (ip + ml <= mflimit)
                ml2 = hc_wider(c, ip + ml - 2, ip, matchlimit, ml, &ref2, &start2, attempts);
            else ml2 = ml;
            if (ml2 == ml) {                                            /* no better match: encode ML1 */
                if (hc_emit(src, &ip, &op, &anchor, ml, ref, limited, oend)) goto overflow;
                continue;
            }
            if (start0 < ip && start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
            if (start2 - ip < 3) {                                      /* first match too small: dropped */
                ml = ml2; ip = start2; ref = ref2;
                goto search2;
            }
        search3:
            if (start2 - ip < OPTIMAL_ML) {
                int new_ml = ml, correction;
                if (new_ml > OPTIMAL_ML) new_ml = OPTIMAL_ML;
                if (ip + new_ml > start2 + ml2 - MINMATCH) new_ml = (int)(start2 - ip) + ml2 - MINMATCH;
                correction = new_ml - (int)(start2 - ip);
                if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
            }
            if (start2 + ml2 <= mflimit)
                ml3 = hc_wider(c, start2 + ml2 - 3, start2, matchlimit, ml2, &ref3, &start3, attempts);
            else ml3 = ml2;
            if (ml3 == ml2) {                                           /* encode ML1 and ML2 */
                if (start2 < ip + ml) ml = (int)(start2 - ip);
                if (hc_emit(src, &ip, &op, &anchor, ml, ref, limited, oend)) goto overflow;
                ip = start2;
                if (hc_emit(src, &ip, &op, &anchor, ml2, ref2, limited, oend)) goto overflow;
                continue;
            }
            if (start3 < ip + ml + 3) {                                 /* not enough room for match 2 */
                if (start3 >= ip + ml) {                                /* Seq1 can go out now; Seq3 becomes Seq1 */
                    if (start2 < ip + ml) {
                        const int correction = (int)(ip + ml - start2);
                        start2 += correction; ref2 += correction; ml2 -= correction;
                        if (ml2 < MINMATCH) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                    }
                    if (hc_emit(src, &ip, &op, &anchor, ml, ref, limited, oend)) goto overflow;
                    ip = start3; ref = ref3; ml = ml3;
                    start0 = start2; ref0 = ref2; ml0 = ml2;
                    goto search2;
                }
                start2 = start3; ref2 = ref3; ml2 = ml3;
                goto search3;
            }
            /* three ascending matches: write the first */
            if (start2 < ip + ml) {
                if (start2 - ip < OPTIMAL_ML) {
                    int correction;
                    if (ml > OPTIMAL_ML) ml = OPTIMAL_ML;
                    if (ip + ml > start2 + ml2 - MINMATCH) ml = (int)(start2 - ip) + ml2 - MINMATCH;
                    correction = ml - (int)(start2 - ip);
                    if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
                } else ml = (int)(start2 - ip);
            }
            if (hc_emit(src, &ip, &op, &anchor, ml, ref, limited, oend)) goto overflow;
            ip = start2; ref = ref2; ml = ml2;
            start2 = start3; ref2 = ref3; ml2 = ml3;
            goto search3;
        }
    }
    {   /* last literals (lz4hc.c:735-762) */
        const size_t run = (size_t)n - anchor, add = (run + 255 - 15) / 255;
        if (limited && op + 1 + add + run > oend) goto overflow;
        if (run >= 15) {
            size_t acc = run - 15;
            *op++ = 0xF0;
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else *op++ = (uint8_t)(run << 4);
        memcpy(op, src + anchor, run); op += run;
        result = (int)(op - dst);
    }
overflow:
    free(c);
    return result;
}

int orc_codec_lz4hc4(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; return orc_lz4hc_compress(src, dst, n, cap, 4); }
int orc_codec_lz4hc8(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; return orc_lz4hc_compress(src, dst, n, cap, 8); }
