/*
 * oracle/lz4mc_port.c — scalar restatement of 4mc's own "Medium" LZ4 encoder (LZ4_compressMC /
 * LZ4_compressMC_limitedOutput), 64-bit little-endian build.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 *   entry     LZ4_compressMC :582-591, _limitedOutput :593-606 -> LZ4MC_compress_generic :518-579
 *   search    LZ4MC_InsertAndFindBestMatch :435-462 (4 attempts = 1 << LZ4MC_DEFAULT_COMPRESSIONLEVEL, :42,:537)
 *   tables    LZ4MC_Insert :387-404 (only the first pending and the last skipped position enter),
 *             init :351-359 (positions stored unbiased: an empty bucket aliases position 0)
 *   emit      LZ4MC_encodeSequence :466-505 (limit checks use length>>8, not length/255)
 * (all in native/lz4/lz4mc.c)
 * Parity: pinned — byte-identical to oracle/_ref on the corpus + edge inputs for both entry points
 * and to the reference CLI's `4mc -2` manifest (tests/golden/corpus_manifest.json).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define HASH_LOG 15
#define MAXD     65536
#define MAXDIST  65535
#define MFLIMIT  12
#define LASTLIT  5

typedef struct { uint32_t hash[1 << HASH_LOG]; uint16_t chain[MAXD]; } mc_t;

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t mc_hash(const uint8_t* p) { return (rd32(p) * 2654435761u) >> (32 - HASH_LOG); }

static uint8_t* put_len(uint8_t* op, uint32_t rest)      /* continuation bytes of a length >= 15 */
{
    for (; rest >= 255; rest -= 255) *op++ = 255;
    *op++ = (uint8_t)rest;
    return op;
}

/* cap < 0: no output limit (LZ4_compressMC); else LZ4_compressMC_limitedOutput(.., cap) */
int orc_lz4mc_compress(const uint8_t* src, uint8_t* dst, int n, int cap)
{
    mc_t* c = (mc_t*)malloc(sizeof *c);
    const int limited = cap >= 0;
    const int64_t mflimit = (int64_t)n - MFLIMIT, matchlimit = (int64_t)n - LASTLIT;
    int64_t ip = 1, anchor = 0, ntu = 1;              /* nextToUpdate = base + 1 (:355) */
    uint8_t* op = dst;
    uint8_t* const oend = dst + (limited ? cap : 0);
    uint32_t tries = 64, step = 1;
    int result = 0;
    memset(c->hash, 0, sizeof c->hash);
    memset(c->chain, 0xFF, sizeof c->chain);

    while (ip < mflimit) {
        int64_t ref, best = 0;
        size_t ml = 0;
        int attempts = 4;
        /* insert: the first pending position, then jump to ip-1 (:391-403) */
        while (ntu < ip) {
            const uint32_t h = mc_hash(src + ntu);
            int64_t delta = ntu - (int64_t)c->hash[h];
            if (delta > MAXDIST) delta = MAXDIST;
            c->chain[ntu & (MAXD - 1)] = (uint16_t)delta;
            c->hash[h] = (uint32_t)ntu;
            ntu++;
            if (ntu < ip) ntu = ip - 1;
        }
        ref = (int64_t)c->hash[mc_hash(src + ip)];
        while ((uint32_t)(ip - ref) <= MAXDIST && attempts) {
            attempts--;
            if (src[ref + ml] == src[ip + ml] && rd32(src + ref) == rd32(src + ip)) {
                size_t k = 4;
                while (ip + (int64_t)k < matchlimit && src[ip + k] == src[ref + k]) k++;
                if (k > ml) { ml = k; best = ref; }
            }
            ref -= c->chain[ref & (MAXD - 1)];
        }
        if (!ml) { ip += step; step = tries++ >> 6; continue; }
        {   /* encode (:466-505) */
            uint32_t len = (uint32_t)(ip - anchor);
            uint8_t* token = op++;
            if (limited && op + len + (2 + 1 + LASTLIT) + (len >> 8) > oend) goto out;
            if (len >= 15) { *token = 0xF0; op = put_len(op, len - 15); } else *token = (uint8_t)(len << 4);
            memcpy(op, src + anchor, len); op += len;
            op[0] = (uint8_t)(ip - best); op[1] = (uint8_t)((ip - best) >> 8); op += 2;
            len = (uint32_t)ml - 4;
            if (limited && op + (1 + LASTLIT) + (len >> 8) > oend) goto out;
            if (len >= 15) { *token += 15; op = put_len(op, len - 15); } else *token += (uint8_t)len;
            ip += (int64_t)ml; anchor = ip;
        }
        step = 1; tries = 64;
    }
    {   /* last literals (:565-573) */
        const uint32_t run = (uint32_t)(n - anchor);
        if (limited && (uint32_t)(op - dst) + run + 1 + (run + 255 - 15) / 255 > (uint32_t)cap) goto out;
        if (run >= 15) { *op++ = 0xF0; op = put_len(op, run - 15); } else *op++ = (uint8_t)(run << 4);
        memcpy(op, src + anchor, run); op += run;
        result = (int)(op - dst);
    }
out:
    free(c);
    return result;
}

int orc_codec_lz4mc(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; return orc_lz4mc_compress(src, dst, n, cap); }
