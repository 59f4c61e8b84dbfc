/*
 * oracle/lz4_port.c — scalar restatement of the LZ4 block codec as 4mc uses it.
 * TEST INFRASTRUCTURE, NOT PRODUCT (see oracle.h).  Written for this repository from the block
 * format and the observable behaviour of the reference:
 *
 *   encoder : LZ4_compress_default, acceleration 1, 64-bit little-endian build
 *             native/lz4/lz4.c:1435 -> :1416 -> :1346-1367 -> :910-1302
 *             (hash fns :758-780, table put/get :800-860, LZ4_count :658-682)
 *   decoder : LZ4_decompress_safe (noDict, decode_full_block)
 *             native/lz4/lz4.c:2345-2350 -> :1936-2339 (fast loop :1995-2110 is compiled in on
 *             x86-64, :457-459; its acceptance set is wider than the safe loop's and is
 *             reproduced here so that accept/reject and the negative return codes agree).
 *
 * Parity: pinned — byte-identical to oracle/_ref on the seeded corpus and on fuzzed/corrupted
 * streams (tests/test_oracle_golden.py), and to the golden vectors in tests/golden/.
 * Documented deviation: a match with offset 0 reads not-yet-written output in the reference
 * (undefined result); the port returns ORC_LZ4_ERR_OFFSET0 for it.
 */
#include <limits.h>
#include <string.h>
#include "oracle.h"

#define ORC_LZ4_ERR_OFFSET0 INT_MIN

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

int orc_lz4_compress_bound(int n)
{
    return ((unsigned)n > 0x7E000000u) ? 0 : n + n / 255 + 16;      /* lz4.h:211-212 */
}

/* ------------------------------------------------------------------------------------------ */
/* Encoder                                                                                    */
/* ------------------------------------------------------------------------------------------ */
enum { TAB_U16 = 0, TAB_U32 = 1 };
#define HASHLOG      12                 /* lz4.h:654 (LZ4_MEMORY_USAGE 14 -> 4096 u32 slots)  */
#define SMALL_LIMIT  (65536 + 11)       /* lz4.c:689  LZ4_64Klimit                            */
#define MAX_DIST     65535u             /* lz4.h:633                                           */
#define MFLIMIT      12
#define LASTLITERALS 5
#define MINLEN       (MFLIMIT + 1)      /* lz4.c:247                                           */

static uint32_t hash_pos(const uint8_t* p, int tab)
{
    if (tab == TAB_U32)                 /* 5-byte hash on 64-bit LE: lz4.c:764-769,776-780     */
        return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - HASHLOG));
    return (rd32(p) * 2654435761u) >> (32 - (HASHLOG + 1));         /* lz4.c:758-759           */
}

/* number of equal bytes of a[] and b[] before a reaches lim  (lz4.c:658-682) */
static uint32_t common_len(const uint8_t* a, const uint8_t* b, const uint8_t* lim)
{
    const uint8_t* const a0 = a;
    while (a < lim && *a == *b) { a++; b++; }
    return (uint32_t)(a - a0);
}

int orc_lz4_compress_fast(const uint8_t* src, uint8_t* dst, int n, int cap)
{
    uint32_t tab32[1 << HASHLOG];            /* viewed as 8192 u16 slots in TAB_U16 mode        */
    uint16_t* const tab16 = (uint16_t*)tab32;
    const int limited = cap < orc_lz4_compress_bound(n);             /* lz4.c:1352            */
    const int tab = (n < SMALL_LIMIT) ? TAB_U16 : TAB_U32;            /* lz4.c:1353-1357       */
    uint8_t* op = dst;
    uint32_t anchor = 0, ip, fwd_hash;

    if ((unsigned)n > 0x7E000000u) return 0;                          /* lz4.c:1324            */
    if (n == 0) {                                                     /* lz4.c:1325-1335       */
        if (limited && cap <= 0) return 0;
        dst[0] = 0; return 1;
    }
    memset(tab32, 0, sizeof tab32);                                   /* LZ4_initStream :1348  */
#define TAB_GET(h)    ((tab == TAB_U32) ? tab32[h] : (uint32_t)tab16[h])
#define TAB_PUT(h, v) do { if (tab == TAB_U32) tab32[h] = (v); else tab16[h] = (uint16_t)(v); } while (0)

    if (n >= MINLEN) {
        const uint32_t mflimit_p1 = (uint32_t)n - MFLIMIT + 1;        /* first pos a match may NOT start at */
        const uint8_t* const matchlimit = src + n - LASTLITERALS;
        TAB_PUT(hash_pos(src, tab), 0);
        ip = 1; fwd_hash = hash_pos(src + 1, tab);
        for (;;) {
            uint32_t cand, fwd = ip, step = 1, tries = 1u << 6;      /* lz4.c:1014-1016        */
            uint8_t* token;
            /* probe forward until a 4-byte match inside the 64 KiB window turns up */
            for (;;) {
                const uint32_t h = fwd_hash, cur = fwd;
                cand = TAB_GET(h);
                ip = fwd; fwd += step; step = tries++ >> 6;           /* lz4.c:1021-1023        */
                if (fwd > mflimit_p1) goto last_literals;
                fwd_hash = hash_pos(src + fwd, tab);
                TAB_PUT(h, cur);
                if (tab == TAB_U32 && cand + MAX_DIST < cur) continue;   /* lz4.c:1064-1067     */
                if (rd32(src + cand) == rd32(src + ip)) break;
            }
            /* extend backwards over equal bytes (lz4.c:1080) */
            while (ip > anchor && cand > 0 && src[ip - 1] == src[cand - 1]) { ip--; cand--; }

            {   const uint32_t lit = ip - anchor;
                token = op++;
                if (limited && (size_t)(op - dst) + lit + (2 + 1 + LASTLITERALS) + lit / 255 > (size_t)cap) return 0;   /* lz4.c:1085 */
                if (lit >= 15) {
                    uint32_t rest = lit - 15;
                    *token = 0xF0;
                    for (; rest >= 255; rest -= 255) *op++ = 255;
                    *op++ = (uint8_t)rest;
                } else *token = (uint8_t)(lit << 4);
                memcpy(op, src + anchor, lit); op += lit;
            }
        next_match:
            {   const uint32_t off = ip - cand;
                uint32_t mcode;
                op[0] = (uint8_t)off; op[1] = (uint8_t)(off >> 8); op += 2;
                mcode = common_len(src + ip + 4, src + cand + 4, matchlimit);
                ip += mcode + 4;
                if (limited && (size_t)(op - dst) + (1 + LASTLITERALS) + (mcode + 240) / 255 > (size_t)cap) return 0;   /* lz4.c:1158 */
                if (mcode >= 15) {
                    *token += 15; mcode -= 15;
                    for (; mcode >= 255; mcode -= 255) *op++ = 255;
                    *op++ = (uint8_t)mcode;
                } else *token += (uint8_t)mcode;
            }
            anchor = ip;
            if (ip >= mflimit_p1) break;
            TAB_PUT(hash_pos(src + ip - 2, tab), ip - 2);             /* lz4.c:1207             */
            {   const uint32_t h = hash_pos(src + ip, tab);           /* lz4.c:1218-1259        */
                cand = TAB_GET(h);
                TAB_PUT(h, ip);
                if ((tab == TAB_U16 || cand + MAX_DIST >= ip) && rd32(src + cand) == rd32(src + ip)) {
                    token = op++; *token = 0;
                    goto next_match;
                }
            }
            fwd_hash = hash_pos(src + (++ip), tab);
        }
    }
last_literals:
    {   const uint32_t run = (uint32_t)n - anchor;
        if (limited && (size_t)(op - dst) + run + 1 + (run + 255 - 15) / 255 > (size_t)cap) return 0;   /* lz4.c:1269 */
        if (run >= 15) {
            uint32_t rest = run - 15;
            *op++ = 0xF0;
            for (; rest >= 255; rest -= 255) *op++ = 255;
            *op++ = (uint8_t)rest;
        } else *op++ = (uint8_t)(run << 4);
        memcpy(op, src + anchor, run); op += run;
    }
    return (int)(op - dst);
#undef TAB_GET
#undef TAB_PUT
}

/* ------------------------------------------------------------------------------------------ */
/* Decoder                                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* Length continuation bytes (lz4.c:1903-1928): adds bytes until one != 255.  `lim` is the
 * position the cursor may not pass; returns 0 on success. */
static int more_len(const uint8_t* src, int64_t* ip, int64_t lim, int check_first, int64_t* len)
{
    uint32_t b;
    if (check_first && *ip >= lim) return -1;
    do {
        b = src[*ip]; (*ip)++;
        *len += b;
        if (*ip > lim) return -1;
    } while (b == 255);
    return 0;
}

static void seq_copy(uint8_t* dst, int64_t op, int64_t from, int64_t len)
{
    /* LZ4 match semantics: byte-serial forward copy (overlap replicates the pattern) */
    int64_t i;
    for (i = 0; i < len; i++) dst[op + i] = dst[from + i];
}

int orc_lz4_decompress_safe(const uint8_t* src, uint8_t* dst, int csize, int cap)
{
    const int64_t iend = csize, oend = cap;
    int64_t ip = 0, op = 0, lit, mlen, off, from;
    unsigned token;
    int fast;

    if (src == NULL || cap < 0) return -1;
    if (cap == 0) return (csize == 1 && src[0] == 0) ? 0 : -1;        /* lz4.c:1977-1981        */
    if (csize == 0) return -1;

    fast = (oend - op) >= 64;                                         /* lz4.c:1990             */
    for (;;) {
        token = src[ip++];
        lit = token >> 4;
        if (fast) {
            /* ---- fast loop (lz4.c:1995-2110): invariant oend-op >= 64 --------------------- */
            if (lit == 15) {
                if (more_len(src, &ip, iend - 15, 1, &lit)) goto err;
                if (op + lit > oend - 32 || ip + lit > iend - 32) goto safe_literals;
            } else {
                if (ip > iend - 17) goto safe_literals;
            }
            memcpy(dst + op, src + ip, (size_t)lit); ip += lit; op += lit;
            off = src[ip] | (src[ip + 1] << 8); ip += 2;
            from = op - off;
            mlen = token & 15;
            if (mlen == 15) {
                if (more_len(src, &ip, iend - 4, 0, &mlen)) goto err;
                mlen += 4;
                if (from < 0) goto err;
                if (op + mlen >= oend - 64) goto safe_match;
            } else {
                mlen += 4;
                if (op + mlen >= oend - 64) goto safe_match;
                if (from < 0) goto err;
            }
            if (off == 0) return ORC_LZ4_ERR_OFFSET0;
            seq_copy(dst, op, from, mlen); op += mlen;
            continue;
        }
        /* ---- safe loop (lz4.c:2114-2325) -------------------------------------------------- */
        if (lit != 15 && ip < iend - 16 && op <= oend - 32) {         /* shortcut :2128-2162    */
            memcpy(dst + op, src + ip, (size_t)lit); ip += lit; op += lit;
            mlen = token & 15;
            off = src[ip] | (src[ip + 1] << 8); ip += 2;
            from = op - off;
            if (mlen != 15 && off >= 8 && from >= 0) {
                seq_copy(dst, op, from, mlen + 4); op += mlen + 4;
                continue;
            }
            goto copy_match;
        }
        if (lit == 15) {
            if (more_len(src, &ip, iend - 15, 1, &lit)) goto err;
        }
    safe_literals:
        fast = 0;
        if (op + lit > oend - MFLIMIT || ip + lit > iend - (2 + 1 + LASTLITERALS)) {
            /* must be the terminating literal run (lz4.c:2175-2225) */
            if (ip + lit != iend || op + lit > oend) goto err;
            memmove(dst + op, src + ip, (size_t)lit);
            op += lit;
            return (int)op;
        }
        memcpy(dst + op, src + ip, (size_t)lit); ip += lit; op += lit;
        off = src[ip] | (src[ip + 1] << 8); ip += 2;
        from = op - off;
        mlen = token & 15;
    copy_match:
        if (mlen == 15) {
            if (more_len(src, &ip, iend - 4, 0, &mlen)) goto err;
        }
        mlen += 4;
    safe_match:
        fast = 0;
        if (from < 0) goto err;                                       /* lz4.c:2250             */
        if (op + mlen > oend - 12 && op + mlen > oend - LASTLITERALS) goto err;   /* :2315-2317 */
        if (off == 0) return ORC_LZ4_ERR_OFFSET0;
        seq_copy(dst, op, from, mlen); op += mlen;
    }
err:
    return (int)(-ip) - 1;                                            /* lz4.c:2336-2337        */
}

/* Sequence list of a stream under the rules of the SAFE loop alone (safe_literals / copy_match above, lz4.c:2114-2325):
 * the subset of streams every path of the decoder accepts with one meaning.  tokpos[i] / outpos[i] = stream position of
 * the token and output position of sequence i; the last sequence is the literal-only tail.  Returns the number of
 * sequences and *total = decoded size (fields, if given: literal start, literal length, match length, offset per sequence),
 * or -1 if a rule of the safe loop is broken (such streams are left to
 * orc_lz4_decompress_safe).  Checker of the GPU parser kernel (tests/test_gpu_lz4par.py). */
int orc_lz4_sequences_ex(const uint8_t* src, int csize, int cap, uint32_t* tokpos, uint32_t* outpos, uint32_t* fields, int maxseq, int* total)
{
    const int64_t iend = csize, oend = cap;
    int64_t ip = 0, op = 0, lit, mlen, off;
    int n = 0;
    if (src == NULL || cap < 64 || csize < 1) return -1;
    for (;;) {
        unsigned token;
        if (ip >= iend || n >= maxseq) return -1;
        tokpos[n] = (uint32_t)ip; outpos[n] = (uint32_t)op; n++;
        token = src[ip++];
        lit = token >> 4;
        if (lit == 15) { if (more_len(src, &ip, iend - 15, 1, &lit)) return -1; }
        if (fields) { fields[4 * (n - 1)] = (uint32_t)ip; fields[4 * (n - 1) + 1] = (uint32_t)lit; fields[4 * (n - 1) + 2] = 0; fields[4 * (n - 1) + 3] = 0; }
        if (op + lit > oend - MFLIMIT || ip + lit > iend - (2 + 1 + LASTLITERALS)) {
            if (ip + lit != iend || op + lit > oend) return -1;
            *total = (int)(op + lit);
            return n;
        }
        ip += lit; op += lit;
        off = src[ip] | (src[ip + 1] << 8); ip += 2;
        mlen = token & 15;
        if (mlen == 15) { if (more_len(src, &ip, iend - 4, 0, &mlen)) return -1; }
        mlen += 4;
        if (off == 0 || op - off < 0) return -1;
        if (op + mlen > oend - LASTLITERALS) return -1;
        if (fields) { fields[4 * (n - 1) + 2] = (uint32_t)mlen; fields[4 * (n - 1) + 3] = (uint32_t)off; }
        op += mlen;
    }
}

int orc_lz4_sequences(const uint8_t* src, int csize, int cap, uint32_t* tokpos, uint32_t* outpos, int maxseq, int* total)
{ return orc_lz4_sequences_ex(src, csize, cap, tokpos, outpos, NULL, maxseq, total); }

int orc_codec_lz4_fast(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; return orc_lz4_compress_fast(src, dst, n, cap); }

int orc_codec_lz4_decode(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; return orc_lz4_decompress_safe(src, dst, n, cap); }
