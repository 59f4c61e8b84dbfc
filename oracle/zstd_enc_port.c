/*
 * oracle/zstd_enc_port.c — scalar restatement of ZSTD_compress() (zstd 1.5.3 as vendored under
 * native/zstd) for the one-shot, no-dictionary, known-size call 4mz makes per block
 * (native/4mc.c:467: ZSTD_compress(out+12, n-1, in, n, level)).  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Covered: ZSTD_fast (4mz "fast" = zstd level 1), ZSTD_dfast ("medium" = level 3), ZSTD_lazy / lazy2 with the row-hash
 * and hash-chain match finders ("high" = level 6: every size class; "ultra" = level 12: lazy2 for inputs > 256 KiB) and
 * ZSTD_btlazy2 (the binary-tree finder of level 12 between 16 KiB + 1 and 256 KiB) and ZSTD_btopt (the optimal parser of
 * level 12 at 16 KiB and below: compress/zstd_opt.c, optLevel 0, first-block statistics), and with them every other level
 * from 1 to 12 (their rows of clevels.h add ZSTD_greedy and other table sizes, nothing else; the JNI name
 * compressBytesDirectHC(level) passes any level through).  Levels 13 and above return ORC_ZSTD_UNSUPPORTED.
 *
 *   parameters   ZSTD_getCParams_internal        compress/zstd_compress.c:6465-6488, clevels.h:25-130,
 *                ZSTD_adjustCParams_internal     compress/zstd_compress.c:1335-1399
 *   frame        ZSTD_writeFrameHeader :4065-4116, ZSTD_compress_frameChunk :3983-4062,
 *                ZSTD_writeEpilogue :4661-4698, block size :1883-1884
 *   block        ZSTD_compressBlock_internal :3812-3877, ZSTD_buildSeqStore :2859-2940,
 *                ZSTD_entropyCompressSeqStore(_internal) :2632-2775, ZSTD_buildSequencesStatistics :2489-2615
 *   match finder ZSTD_compressBlock_fast_noDict_generic   compress/zstd_fast.c:95-365
 *                ZSTD_compressBlock_doubleFast_noDict_generic  compress/zstd_double_fast.c:98-330
 *   literals     ZSTD_compressLiterals           compress/zstd_compress_literals.c:100-196
 *   Huffman      HUF_compress_internal           compress/huf_compress.c:1250-1360 (+ sort :604, tree :665,
 *                setMaxHeight :360, writeCTable :230, compressWeights :147, 1X/4X streams :1029-1194)
 *   FSE          fse_compress.c: buildCTable :68-200, writeNCount :224-330, optimalTableLog :355-368,
 *                normalizeCount :373-520, compress_usingCTable :560-620
 *   sequences    zstd_compress_sequences.c: selectEncodingType :153-239, buildCTable :246-300,
 *                encodeSequences :302-400;   codes: zstd_compress_internal.h:480-509
 *
 * Positions are offsets into the input; a hash-table index is position + 2 as in the reference
 * (ZSTD_WINDOW_START_INDEX), so 0 means "empty".
 * Parity: pinned — byte-identical to oracle/_ref (ZSTD_compress, levels 1 .. 12) on the corpus blocks, edge
 * inputs and tails, with capacity n-1 and ZSTD_compressBound(n) (tests/test_oracle_golden.py), and
 * to the per-block manifest of the reference CLI at `4mc -z -1` (tests/golden/corpus_manifest.json).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define ERR_GENERIC   (-1)
#define ERR_TOOSMALL  (-70)               /* ZSTD_error_dstSize_tooSmall */
#define BLOCK_MAX     (128 * 1024)

typedef struct { uint32_t wlog, clog, hlog, slog, mml, tlen, strat; } zparams;

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static int hibit(uint32_t v) { return 31 - __builtin_clz(v); }

/* ------------------------------------------------------------------------------------------------
 * bit writer with the overflow rule of BIT_CStream_t / HUF_CStream_t: the stream "does not fit"
 * when floor(total_bits / 8) >= cap - 8 (bitstream.h:153-163,:219-240, huf_compress.c:830-956).
 */
typedef struct { uint8_t* p; size_t cap, pos; uint64_t acc; unsigned nb; uint64_t total; } bitw;

static int bw_init(bitw* w, uint8_t* p, size_t cap)
{ w->p = p; w->cap = cap; w->pos = 0; w->acc = 0; w->nb = 0; w->total = 0; return cap > 8; }

static void bw_put(bitw* w, uint64_t v, unsigned n)          /* n <= 32 */
{
    w->acc |= (v & ((1ull << n) - 1)) << w->nb;
    w->nb += n; w->total += n;
    while (w->nb >= 8) {
        if (w->pos < w->cap) w->p[w->pos] = (uint8_t)w->acc;
        w->pos++; w->acc >>= 8; w->nb -= 8;
    }
}

static size_t bw_close(bitw* w)                              /* end mark + size, 0 = did not fit */
{
    bw_put(w, 1, 1);
    if ((w->total >> 3) + 8 >= w->cap) return 0;
    if (w->nb) { w->p[w->pos] = (uint8_t)w->acc; return w->pos + 1; }
    return w->pos;
}

/* ------------------------------------------------------------------------------------------------ FSE */
typedef struct {
    uint16_t next[1 << 9];       /* state table (max table log 9 on this path) */
    uint32_t dbits[64];          /* deltaNbBits   */
    int32_t  dfind[64];          /* deltaFindState */
    uint32_t log;
    uint32_t maxsym;             /* ZSTD_getFSEMaxSymbolValue */
} fse_ct;

static unsigned hist(uint32_t* count, unsigned* max_sym, const uint8_t* s, size_t n)   /* hist.c:31-60 */
{
    unsigned m = *max_sym, largest = 0;
    memset(count, 0, (m + 1) * sizeof *count);
    if (!n) { *max_sym = 0; return 0; }
    for (size_t i = 0; i < n; i++) count[s[i]]++;
    while (!count[m]) m--;
    *max_sym = m;
    for (unsigned i = 0; i <= m; i++) if (count[i] > largest) largest = count[i];
    return largest;
}

static uint32_t fse_min_log(size_t n, unsigned max_sym)
{
    const uint32_t a = (uint32_t)hibit((uint32_t)n) + 1, b = (uint32_t)hibit(max_sym) + 2;
    return a < b ? a : b;
}

static uint32_t fse_optimal_log(uint32_t max_log, size_t n, unsigned max_sym, unsigned minus)
{
    const uint32_t src_bits = (uint32_t)hibit((uint32_t)(n - 1)) - minus, min_bits = fse_min_log(n, max_sym);
    uint32_t log = max_log ? max_log : 11;
    if (src_bits < log) log = src_bits;
    if (min_bits > log) log = min_bits;
    if (log < 5) log = 5;
    if (log > 12) log = 12;
    return log;
}

static int fse_normalize_m2(int16_t* norm, uint32_t log, const uint32_t* count, size_t total, unsigned max_sym, int16_t low)
{
    uint32_t distributed = 0, todo, s;
    const uint32_t low_thr = (uint32_t)(total >> log);
    uint32_t low_one = (uint32_t)((total * 3) >> (log + 1));
    for (s = 0; s <= max_sym; s++) {
        if (!count[s]) { norm[s] = 0; continue; }
        if (count[s] <= low_thr) { norm[s] = low; distributed++; total -= count[s]; continue; }
        if (count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = -2;
    }
    todo = (1u << log) - distributed;
    if (!todo) return 0;
    if (total / todo > low_one) {
        low_one = (uint32_t)((total * 3) / (todo * 2));
        for (s = 0; s <= max_sym; s++)
            if (norm[s] == -2 && count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; }
        todo = (1u << log) - distributed;
    }
    if (distributed == max_sym + 1) {
        uint32_t best = 0, bc = 0;
        for (s = 0; s <= max_sym; s++) if (count[s] > bc) { best = s; bc = count[s]; }
        norm[best] += (int16_t)todo;
        return 0;
    }
    if (!total) {
        for (s = 0; todo > 0; s = (s + 1) % (max_sym + 1)) if (norm[s] > 0) { todo--; norm[s]++; }
        return 0;
    }
    {
        const uint64_t vlog = 62 - log, mid = (1ull << (vlog - 1)) - 1;
        const uint64_t rstep = (((1ull << vlog) * todo) + mid) / (uint32_t)total;
        uint64_t acc = mid;
        for (s = 0; s <= max_sym; s++) if (norm[s] == -2) {
            const uint64_t end = acc + count[s] * rstep;
            const uint32_t w = (uint32_t)(end >> vlog) - (uint32_t)(acc >> vlog);
            if (w < 1) return ERR_GENERIC;
            norm[s] = (int16_t)w; acc = end;
        }
    }
    return 0;
}

static int fse_normalize(int16_t* norm, uint32_t log, const uint32_t* count, size_t total, unsigned max_sym, int use_low)
{
    static const uint32_t rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    const int16_t low = use_low ? -1 : 1;
    const uint64_t scale = 62 - log, step = (1ull << 62) / (uint32_t)total, vstep = 1ull << (scale - 20);
    const uint32_t low_thr = (uint32_t)(total >> log);
    int remaining = 1 << log;
    unsigned s, largest = 0;
    int16_t largest_p = 0;
    if (log < fse_min_log(total, max_sym)) return ERR_GENERIC;
    for (s = 0; s <= max_sym; s++) {
        if (count[s] == total) return 0;
        if (!count[s]) { norm[s] = 0; continue; }
        if (count[s] <= low_thr) { norm[s] = low; remaining--; continue; }
        {
            int16_t p = (int16_t)((count[s] * step) >> scale);
            if (p < 8) p += (count[s] * step) - ((uint64_t)p << scale) > vstep * rtb[p];
            if (p > largest_p) { largest_p = p; largest = s; }
            norm[s] = p; remaining -= p;
        }
    }
    if (-remaining >= (norm[largest] >> 1)) return fse_normalize_m2(norm, log, count, total, max_sym, low);
    norm[largest] += (int16_t)remaining;
    return 0;
}

/* returns bytes written or ERR_* */
static int fse_write_ncount(uint8_t* out, size_t cap, const int16_t* norm, unsigned max_sym, uint32_t log)
{
    const size_t bound = max_sym ? (((max_sym + 1) * log + 4 + 2) / 8) + 1 + 2 : 512;
    const int safe = cap >= bound;
    const unsigned alphabet = max_sym + 1;
    size_t o = 0;
    int nbits = (int)log + 1, remaining = (1 << log) + 1, threshold = 1 << log, bc = 4, prev0 = 0;
    uint32_t bs = log - 5;
    unsigned sym = 0;
#define NC_FLUSH() do { if (!safe && o + 2 > cap) return ERR_TOOSMALL; \
        out[o] = (uint8_t)bs; out[o + 1] = (uint8_t)(bs >> 8); o += 2; bs >>= 16; } while (0)
    while (sym < alphabet && remaining > 1) {
        if (prev0) {
            unsigned start = sym;
            while (sym < alphabet && !norm[sym]) sym++;
            if (sym == alphabet) break;
            while (sym >= start + 24) { start += 24; bs += 0xFFFFu << bc; NC_FLUSH(); }
            while (sym >= start + 3) { start += 3; bs += 3u << bc; bc += 2; }
            bs += (sym - start) << bc; bc += 2;
            if (bc > 16) { NC_FLUSH(); bc -= 16; }
        }
        {
            int c = norm[sym++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= c < 0 ? -c : c;
            c++;
            if (c >= threshold) c += max;
            bs += (uint32_t)c << bc;
            bc += nbits; bc -= (c < max);
            prev0 = (c == 1);
            if (remaining < 1) return ERR_GENERIC;
            while (remaining < threshold) { nbits--; threshold >>= 1; }
        }
        if (bc > 16) { NC_FLUSH(); bc -= 16; }
    }
    if (remaining != 1) return ERR_GENERIC;
    if (!safe && o + 2 > cap) return ERR_TOOSMALL;
    out[o] = (uint8_t)bs; out[o + 1] = (uint8_t)(bs >> 8);
    o += (size_t)(bc + 7) / 8;
#undef NC_FLUSH
    return (int)o;
}

static void fse_build(fse_ct* ct, const int16_t* norm, unsigned max_sym, uint32_t log)
{
    const uint32_t size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint16_t cumul[66];
    uint8_t symbol_at[1 << 9];
    uint32_t high = size - 1, pos = 0, total = 0;
    ct->log = log; ct->maxsym = max_sym;
    cumul[0] = 0;
    for (unsigned s = 0; s <= max_sym; s++) {
        if (norm[s] == -1) { cumul[s + 1] = cumul[s] + 1; symbol_at[high--] = (uint8_t)s; }
        else cumul[s + 1] = cumul[s] + (uint16_t)norm[s];
    }
    for (unsigned s = 0; s <= max_sym; s++)
        for (int i = 0; i < norm[s]; i++) {
            symbol_at[pos] = (uint8_t)s;
            do pos = (pos + step) & mask; while (pos > high);
        }
    for (uint32_t u = 0; u < size; u++) ct->next[cumul[symbol_at[u]]++] = (uint16_t)(size + u);
    for (unsigned s = 0; s <= max_sym; s++) {
        const int n = norm[s];
        if (n == 0) { ct->dbits[s] = ((log + 1) << 16) - size; ct->dfind[s] = 0; }
        else if (n == -1 || n == 1) { ct->dbits[s] = (log << 16) - size; ct->dfind[s] = (int32_t)total - 1; total++; }
        else {
            const uint32_t max_out = log - (uint32_t)hibit((uint32_t)n - 1);
            ct->dbits[s] = (max_out << 16) - ((uint32_t)n << max_out);
            ct->dfind[s] = (int32_t)total - n;
            total += (uint32_t)n;
        }
    }
}

static void fse_build_rle(fse_ct* ct, unsigned sym)
{ ct->log = 0; ct->maxsym = sym; ct->next[0] = ct->next[1] = 0; ct->dbits[sym] = 0; ct->dfind[sym] = 0; }

static uint32_t fse_first_state(const fse_ct* ct, unsigned sym)              /* FSE_initCState2 */
{
    const uint32_t nb = (ct->dbits[sym] + (1u << 15)) >> 16;
    const uint32_t v = (nb << 16) - ct->dbits[sym];
    return ct->next[(int32_t)(v >> nb) + ct->dfind[sym]];
}

static uint32_t fse_encode(bitw* w, const fse_ct* ct, uint32_t state, unsigned sym)
{
    const uint32_t nb = (state + ct->dbits[sym]) >> 16;
    bw_put(w, state, nb);
    return ct->next[(int32_t)(state >> nb) + ct->dfind[sym]];
}

/* FSE_compress_usingCTable with two interleaved states (used for Huffman weights only) */
static size_t fse_compress2(uint8_t* dst, size_t cap, const uint8_t* src, size_t n, const fse_ct* ct)
{
    bitw w;
    size_t i = n;
    uint32_t s1, s2;
    if (n <= 2 || !bw_init(&w, dst, cap)) return 0;
    if (n & 1) { s1 = fse_first_state(ct, src[--i]); s2 = fse_first_state(ct, src[--i]); s1 = fse_encode(&w, ct, s1, src[--i]); }
    else       { s2 = fse_first_state(ct, src[--i]); s1 = fse_first_state(ct, src[--i]); }
    while (i > 0) { s2 = fse_encode(&w, ct, s2, src[--i]); s1 = fse_encode(&w, ct, s1, src[--i]); }
    bw_put(&w, s2, ct->log); bw_put(&w, s1, ct->log);
    return bw_close(&w);
}

/* ------------------------------------------------------------------------------------------------ Huffman */
typedef struct { uint8_t nbits[256]; uint16_t code[256]; uint32_t log; } huf_ct;
typedef struct { uint32_t count; uint16_t parent; uint8_t byte, nbits; } hnode;

static uint32_t huf_bucket(uint32_t c) { return c < 165 ? c : (uint32_t)hibit(c) + 158; }   /* huf_compress.c:497-517 */

static void huf_isort(hnode* a, int n)
{
    for (int i = 1; i < n; i++) {
        const hnode key = a[i];
        int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}

static void huf_qsort(hnode* a, int lo, int hi)              /* descending; pivot = rightmost (:555-600) */
{
    if (hi - lo < 8) { huf_isort(a + lo, hi - lo + 1); return; }
    while (lo < hi) {
        const uint32_t pivot = a[hi].count;
        int i = lo - 1, idx;
        hnode t;
        for (int j = lo; j < hi; j++) if (a[j].count > pivot) { i++; t = a[i]; a[i] = a[j]; a[j] = t; }
        t = a[i + 1]; a[i + 1] = a[hi]; a[hi] = t;
        idx = i + 1;
        if (idx - lo < hi - idx) { huf_qsort(a, lo, idx - 1); lo = idx + 1; }
        else                     { huf_qsort(a, idx + 1, hi); hi = idx - 1; }
    }
}

static void huf_sort(hnode* node, const uint32_t* count, unsigned max_sym)
{
    struct { uint16_t base, curr; } rank[192];
    memset(rank, 0, sizeof rank);
    for (unsigned s = 0; s <= max_sym; s++) rank[huf_bucket(count[s])].base++;
    for (int r = 191; r > 0; r--) { rank[r - 1].base += rank[r].base; rank[r - 1].curr = rank[r - 1].base; }
    for (unsigned s = 0; s <= max_sym; s++) {
        const uint32_t pos = rank[huf_bucket(count[s]) + 1].curr++;
        node[pos].count = count[s]; node[pos].byte = (uint8_t)s;
    }
    for (int r = 165; r < 191; r++) {
        const int len = rank[r].curr - rank[r].base;
        if (len > 1) huf_qsort(node + rank[r].base, 0, len - 1);
    }
}

static uint32_t huf_limit_height(hnode* node, uint32_t last, uint32_t target)   /* HUF_setMaxHeight :360-470 */
{
    const uint32_t largest = node[last].nbits;
    int cost = 0, n = (int)last;
    uint32_t rank_last[14];
    if (largest <= target) return largest;
    {
        const int base = 1 << (largest - target);
        while (node[n].nbits > target) { cost += base - (1 << (largest - node[n].nbits)); node[n].nbits = (uint8_t)target; n--; }
        while (node[n].nbits == target) n--;
        cost >>= (largest - target);
    }
    for (int i = 0; i < 14; i++) rank_last[i] = 0xF0F0F0F0u;
    {
        uint32_t cur = target;
        for (int pos = n; pos >= 0; pos--) {
            if (node[pos].nbits >= cur) continue;
            cur = node[pos].nbits;
            rank_last[target - cur] = (uint32_t)pos;
        }
    }
    while (cost > 0) {
        uint32_t dec = (uint32_t)hibit((uint32_t)cost) + 1;
        for (; dec > 1; dec--) {
            const uint32_t hp = rank_last[dec], lp = rank_last[dec - 1];
            if (hp == 0xF0F0F0F0u) continue;
            if (lp == 0xF0F0F0F0u) break;
            if (node[hp].count <= 2 * node[lp].count) break;
        }
        while (dec <= 12 && rank_last[dec] == 0xF0F0F0F0u) dec++;
        cost -= 1 << (dec - 1);
        node[rank_last[dec]].nbits++;
        if (rank_last[dec - 1] == 0xF0F0F0F0u) rank_last[dec - 1] = rank_last[dec];
        if (rank_last[dec] == 0) rank_last[dec] = 0xF0F0F0F0u;
        else {
            rank_last[dec]--;
            if (node[rank_last[dec]].nbits != target - dec) rank_last[dec] = 0xF0F0F0F0u;
        }
    }
    while (cost < 0) {
        if (rank_last[1] == 0xF0F0F0F0u) {
            while (node[n].nbits == target) n--;
            node[n + 1].nbits--;
            rank_last[1] = (uint32_t)(n + 1);
            cost++;
            continue;
        }
        node[rank_last[1] + 1].nbits--;
        rank_last[1]++;
        cost++;
    }
    return target;
}

/* HUF_buildCTable_wksp: returns the table log */
static uint32_t huf_build(huf_ct* ct, const uint32_t* count, unsigned max_sym, uint32_t max_bits)
{
    hnode store[513];
    hnode* const node = store + 1;               /* node[-1] is the barrier entry */
    int last, low_s, low_n, nb = 256, root;
    memset(store, 0, sizeof store);
    huf_sort(node, count, max_sym);
    last = (int)max_sym;
    while (!node[last].count) last--;
    low_s = last; root = nb + low_s - 1; low_n = nb;
    node[nb].count = node[low_s].count + node[low_s - 1].count;
    node[low_s].parent = node[low_s - 1].parent = (uint16_t)nb;
    nb++; low_s -= 2;
    for (int n = nb; n <= root; n++) node[n].count = 1u << 30;
    node[-1].count = 1u << 31;
    while (nb <= root) {
        const int n1 = node[low_s].count < node[low_n].count ? low_s-- : low_n++;
        const int n2 = node[low_s].count < node[low_n].count ? low_s-- : low_n++;
        node[nb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (uint16_t)nb;
        nb++;
    }
    node[root].nbits = 0;
    for (int n = root - 1; n >= 256; n--) node[n].nbits = (uint8_t)(node[node[n].parent].nbits + 1);
    for (int n = 0; n <= last; n++) node[n].nbits = (uint8_t)(node[node[n].parent].nbits + 1);
    max_bits = huf_limit_height(node, (uint32_t)last, max_bits);
    {   /* canonical codes: values ascend within a rank in symbol order (:714-735) */
        uint16_t per_rank[13] = {0}, val[13] = {0}, min = 0;
        for (int n = 0; n <= last; n++) per_rank[node[n].nbits]++;
        for (int r = (int)max_bits; r > 0; r--) { val[r] = min; min = (uint16_t)((min + per_rank[r]) >> 1); }
        memset(ct, 0, sizeof *ct);
        for (unsigned n = 0; n <= max_sym; n++) ct->nbits[node[n].byte] = node[n].nbits;
        for (unsigned s = 0; s <= max_sym; s++) ct->code[s] = ct->nbits[s] ? val[ct->nbits[s]]++ : 0;
        ct->log = max_bits;
    }
    return max_bits;
}

/* HUF_writeCTable_wksp: bytes written or ERR_* */
static int huf_write_table(uint8_t* dst, size_t cap, const huf_ct* ct, unsigned max_sym, uint32_t log)
{
    uint8_t weight[256];
    for (unsigned s = 0; s < max_sym; s++) weight[s] = ct->nbits[s] ? (uint8_t)(log + 1 - ct->nbits[s]) : 0;
    if (cap < 1) return ERR_TOOSMALL;
    {   /* HUF_compressWeights (:147-186) */
        uint8_t* const out = dst + 1;
        const size_t ocap = cap - 1;
        int h = 0;
        if (max_sym > 1) {
            uint32_t count[13];
            int16_t norm[13];
            unsigned wmax = 12;
            const unsigned top = hist(count, &wmax, weight, max_sym);
            if (top == max_sym) h = 1;
            else if (top == 1) h = 0;
            else {
                fse_ct wt;
                const uint32_t wlog = fse_optimal_log(6, max_sym, wmax, 2);
                int nc, r = fse_normalize(norm, wlog, count, max_sym, wmax, 0);
                size_t c;
                if (r < 0) return r;
                nc = fse_write_ncount(out, ocap, norm, wmax, wlog);
                if (nc < 0) return nc;
                fse_build(&wt, norm, wmax, wlog);
                c = fse_compress2(out + nc, ocap - (size_t)nc, weight, max_sym, &wt);
                h = c ? nc + (int)c : 0;
            }
        }
        if (h > 1 && (unsigned)h < max_sym / 2) { dst[0] = (uint8_t)h; return h + 1; }
    }
    if (max_sym > 128) return ERR_GENERIC;
    if (((max_sym + 1) / 2) + 1 > cap) return ERR_TOOSMALL;
    dst[0] = (uint8_t)(128 + (max_sym - 1));
    weight[max_sym] = 0;
    for (unsigned s = 0; s < max_sym; s += 2) dst[s / 2 + 1] = (uint8_t)((weight[s] << 4) + weight[s + 1]);
    return (int)((max_sym + 1) / 2) + 1;
}

static size_t huf_stream(uint8_t* dst, size_t cap, const uint8_t* src, size_t n, const huf_ct* ct)   /* 1X (:1029-1094) */
{
    bitw w;
    if (cap < 8 || !bw_init(&w, dst, cap)) return 0;
    for (size_t i = n; i-- > 0;) bw_put(&w, ct->code[src[i]], ct->nbits[src[i]]);
    return bw_close(&w);
}

/* HUF_compressCTable_internal: `hdr` bytes of table description already sit in front of `dst` */
static size_t huf_encode(uint8_t* dst, size_t cap, size_t hdr, const uint8_t* src, size_t n, int four, const huf_ct* ct)
{
    size_t c;
    if (!four) c = huf_stream(dst, cap, src, n, ct);
    else {
        const size_t seg = (n + 3) / 4;
        size_t o = 6;
        if (cap < 6 + 1 + 1 + 1 + 8 || n < 12) return 0;
        for (int k = 0; k < 4; k++) {
            const size_t len = k < 3 ? seg : n - 3 * seg;
            const size_t s = huf_stream(dst + o, cap - o, src + (size_t)k * seg, len, ct);
            if (s == 0 || s > 65535) return 0;
            if (k < 3) { dst[2 * k] = (uint8_t)s; dst[2 * k + 1] = (uint8_t)(s >> 8); }
            o += s;
        }
        c = o;
    }
    if (!c) return 0;
    if (hdr + c >= n - 1) return 0;
    return hdr + c;
}

enum { REP_NONE = 0, REP_CHECK = 1, REP_VALID = 2 };

/* HUF_compress_internal (1X/4X _repeat).  `table` holds the previous block's table on entry and the
 * table in force on exit.  Returns compressed size, 0 (not compressible) or ERR_*. */
static int64_t huf_compress(uint8_t* dst, size_t cap, const uint8_t* src, size_t n, int four,
                            huf_ct* table, int* repeat, int prefer_repeat, int suspect)
{
    uint32_t count[256];
    unsigned max_sym = 255;
    huf_ct fresh;
    uint32_t log;
    int h;
    if (!n || !cap) return 0;
    if (prefer_repeat && *repeat == REP_VALID) return (int64_t)huf_encode(dst, cap, 0, src, n, four, table);
    if (suspect && n >= 4096 * 10) {
        unsigned m1 = 255, m2 = 255;
        const unsigned a = hist(count, &m1, src, 4096), b = hist(count, &m2, src + n - 4096, 4096);
        if (a + b <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {
        const unsigned largest = hist(count, &max_sym, src, n);
        if (largest == n) { dst[0] = src[0]; return 1; }
        if (largest <= (n >> 7) + 4) return 0;
    }
    if (*repeat == REP_CHECK) {
        int bad = 0;
        for (unsigned s = 0; s <= max_sym; s++) bad |= (count[s] != 0) & (table->nbits[s] == 0);
        if (bad) *repeat = REP_NONE;
    }
    if (prefer_repeat && *repeat != REP_NONE) return (int64_t)huf_encode(dst, cap, 0, src, n, four, table);
    log = fse_optimal_log(11, n, max_sym, 1);
    log = huf_build(&fresh, count, max_sym, log);
    h = huf_write_table(dst, cap, &fresh, max_sym, log);
    if (h < 0) return h;
    if (*repeat != REP_NONE) {
        size_t old_bits = 0, new_bits = 0;
        for (unsigned s = 0; s <= max_sym; s++) { old_bits += (size_t)table->nbits[s] * count[s]; new_bits += (size_t)fresh.nbits[s] * count[s]; }
        if ((old_bits >> 3) <= (size_t)h + (new_bits >> 3) || (size_t)h + 12 >= n)
            return (int64_t)huf_encode(dst, cap, 0, src, n, four, table);
    }
    if ((size_t)h + 12 >= n) return 0;
    *repeat = REP_NONE;
    *table = fresh;
    return (int64_t)huf_encode(dst + h, cap - (size_t)h, (size_t)h, src, n, four, &fresh);
}

/* ------------------------------------------------------------------------------------------------ literals */
typedef struct { huf_ct huf; int huf_repeat; uint32_t rep[3]; fse_ct fse[3]; int fse_repeat[3]; /* ll, of, ml */ } zentropy;

static int64_t raw_literals(uint8_t* dst, size_t cap, const uint8_t* src, size_t n, int rle)
{
    const unsigned fl = 1 + (n > 31) + (n > 4095);
    const uint32_t type = rle ? 1u : 0u;
    if (!rle && n + fl > cap) return ERR_TOOSMALL;
    if (fl == 1) dst[0] = (uint8_t)(type + (n << 3));
    else if (fl == 2) { const uint32_t v = type + (1u << 2) + ((uint32_t)n << 4); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); }
    else { const uint32_t v = type + (3u << 2) + ((uint32_t)n << 4); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); }
    if (rle) { dst[fl] = src[0]; return fl + 1; }
    memcpy(dst + fl, src, n);
    return (int64_t)(n + fl);
}

static int64_t compress_literals(const zentropy* prev, zentropy* next, uint8_t* dst, size_t cap,
                                 const uint8_t* src, size_t n, int suspect, uint32_t strat)
{
    const size_t min_gain = (n >> 6) + 2, lh = 3 + (n >= 1024) + (n >= 16384);
    int single = n < 256, repeat = prev->huf_repeat, type = 2;
    int64_t c;
    next->huf = prev->huf; next->huf_repeat = prev->huf_repeat;
    if (n <= (size_t)(prev->huf_repeat == REP_VALID ? 6 : 63)) return raw_literals(dst, cap, src, n, 0);
    if (cap < lh + 1) return ERR_TOOSMALL;
    if (repeat == REP_VALID && lh == 3) single = 1;
    c = huf_compress(dst + lh, cap - lh, src, n, !single, &next->huf, &repeat, strat < 4 /* ZSTD_lazy */ && n <= 1024, suspect);
    if (repeat != REP_NONE) type = 3;
    if (c <= 0 || (size_t)c >= n - min_gain) { next->huf = prev->huf; return raw_literals(dst, cap, src, n, 0); }
    if (c == 1) { next->huf = prev->huf; return raw_literals(dst, cap, src, n, 1); }
    if (type == 2) next->huf_repeat = REP_CHECK;
    if (lh == 3) { const uint32_t v = (uint32_t)type + ((uint32_t)!single << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 14); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); }
    else if (lh == 4) { const uint32_t v = (uint32_t)type + (2u << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 18); memcpy(dst, &v, 4); }
    else { const uint32_t v = (uint32_t)type + (3u << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 22); memcpy(dst, &v, 4); dst[4] = (uint8_t)(c >> 10); }
    return (int64_t)lh + c;
}

/* ------------------------------------------------------------------------------------------------ sequences */
typedef struct { uint32_t ll, ml /* match length - 3 */, off /* offBase */; } zseq;

static const uint8_t  kLLBits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint8_t  kMLBits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
static const int16_t  kLLNorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int16_t  kMLNorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static const int16_t  kOFNorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

static unsigned ll_code(uint32_t v)
{
    static const uint8_t t[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
        22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
    return v > 63 ? (unsigned)hibit(v) + 19 : t[v];
}

static unsigned ml_code(uint32_t v)
{
    if (v > 127) return (unsigned)hibit(v) + 36;
    if (v < 32) return v;
    if (v < 40) return 32 + ((v - 32) >> 1);
    if (v < 48) return 36 + ((v - 40) >> 2);
    if (v < 64) return 38 + ((v - 48) >> 3);
    if (v < 96) return 40 + ((v - 64) >> 4);
    return 42;
}

/* floor(-log2(i / 256) * 256): kInverseProbabilityLog256 of zstd_compress_sequences.c:19-42 */
static unsigned inv_prob_log256(unsigned i)
{
    static unsigned tab[256];
    if (!tab[1]) {
        /* fixed point log2 by repeated squaring, exact enough for 11 fractional bits over 1..255 */
        for (unsigned v = 1; v < 256; v++) {
            uint64_t x = (uint64_t)v << 56;                 /* v / 256 in 0.64 fixed point */
            unsigned ip = 0, r = 0;
            while (!(x >> 63)) { x <<= 1; ip++; }           /* v/256 = 2^-(ip+1) * m, m in [1,2) held as 1.63 fixed point */
            for (int k = 0; k < 24; k++) {                  /* fractional bits of log2(m) */
                const __uint128_t sq = (__uint128_t)x * x;
                x = (uint64_t)(sq >> 63);
                r <<= 1;
                if ((sq >> 127) & 1) { r |= 1; x = (uint64_t)(sq >> 64); }   /* m^2 >= 2: bit set, halve */
            }
            /* -log2(v/256) = ip + 1 - frac, frac = r / 2^24 */
            {
                const uint64_t neg = ((uint64_t)(ip + 1) << 24) - r;     /* 24 fractional bits */
                tab[v] = (unsigned)(neg >> 16);                    /* * 256, floor */
            }
        }
    }
    return tab[i];
}

#define COST_ERR ((size_t)-1)

/* ZSTD_fseBitCost (:100-128) with FSE_bitCost (common/fse.h:570-586) */
static size_t fse_bit_cost(const fse_ct* ct, const uint32_t* count, unsigned max)
{
    size_t cost = 0;
    if (ct->maxsym < max) return COST_ERR;
    for (unsigned s = 0; s <= max; s++) {
        const uint32_t log = ct->log, bad = (log + 1) << 8;
        const uint32_t min_bits = ct->dbits[s] >> 16, threshold = (min_bits + 1) << 16;
        const uint32_t delta = threshold - (ct->dbits[s] + (1u << log));
        const uint32_t bits = (min_bits + 1) * 256 - ((delta << 8) >> log);
        if (!count[s]) continue;
        if (bits >= bad) return COST_ERR;
        cost += (size_t)count[s] * bits;
    }
    return cost >> 8;
}

/* ZSTD_selectEncodingType (zstd_compress_sequences.c:153-239), no dictionary: 0 basic, 1 rle, 2 compressed, 3 repeat */
static int select_type(int* repeat, const uint32_t* count, unsigned max, size_t most, size_t nseq, uint32_t fse_log,
                       const fse_ct* prev, const int16_t* def_norm, uint32_t def_log, int def_ok, uint32_t strat)
{
    if (most == nseq) { *repeat = REP_NONE; return (def_ok && nseq <= 2) ? 0 : 1; }
    if (strat < 4) {                                               /* < ZSTD_lazy */
        if (def_ok) {
            const size_t dyn_min = (((size_t)1 << def_log) * (10 - strat)) >> 3;
            if (*repeat == REP_VALID && nseq < 1000) return 3;
            if (nseq < dyn_min || most < (nseq >> (def_log - 1))) { *repeat = REP_NONE; return 0; }
        }
    } else {
        size_t basic = COST_ERR, rep = COST_ERR, nc, comp = 0;
        if (def_ok) {                                              /* ZSTD_crossEntropyCost */
            basic = 0;
            for (unsigned s = 0; s <= max; s++) basic += (size_t)count[s] * inv_prob_log256((unsigned)(def_norm[s] != -1 ? def_norm[s] : 1) << (8 - def_log));
            basic >>= 8;
        }
        if (*repeat != REP_NONE) rep = fse_bit_cost(prev, count, max);
        {                                                          /* ZSTD_NCountCost + ZSTD_entropyCost */
            uint8_t tmp[512]; int16_t norm[53];
            const uint32_t log = fse_optimal_log(fse_log, nseq, max, 2);
            fse_normalize(norm, log, count, nseq, max, nseq >= 2048);
            nc = (size_t)fse_write_ncount(tmp, sizeof tmp, norm, max, log);
            for (unsigned s = 0; s <= max; s++) {
                unsigned q = (unsigned)((256 * (uint64_t)count[s]) / nseq);
                if (count[s] && !q) q = 1;
                comp += (size_t)count[s] * inv_prob_log256(q);
            }
            comp = (nc << 3) + (comp >> 8);
        }
        if (basic <= rep && basic <= comp) { *repeat = REP_NONE; return 0; }
        if (rep <= comp) return 3;
    }
    *repeat = REP_CHECK;
    return 2;
}

/* ZSTD_buildCTable: bytes of table description written, or ERR_* */
static int build_seq_table(uint8_t* dst, size_t cap, fse_ct* ct, const fse_ct* prev, uint32_t fse_log, int type, uint32_t* count, unsigned max,
                           const uint8_t* codes, size_t nseq, const int16_t* def_norm, uint32_t def_log, unsigned def_max)
{
    if (type == 1) { fse_build_rle(ct, max); if (!cap) return ERR_TOOSMALL; dst[0] = codes[0]; return 1; }
    if (type == 0) { fse_build(ct, def_norm, def_max, def_log); return 0; }
    if (type == 3) { *ct = *prev; return 0; }
    {
        int16_t norm[53];
        size_t n1 = nseq;
        const uint32_t log = fse_optimal_log(fse_log, nseq, max, 2);
        int r;
        if (count[codes[nseq - 1]] > 1) { count[codes[nseq - 1]]--; n1--; }
        r = fse_normalize(norm, log, count, n1, max, n1 >= 2048);
        if (r < 0) return r;
        r = fse_write_ncount(dst, cap, norm, max, log);
        if (r < 0) return r;
        fse_build(ct, norm, max, log);
        return r;
    }
}

/* sequences section after the literals; returns bytes or 0 / ERR_* (ZSTD_entropyCompressSeqStore_internal tail) */
static int64_t encode_sequences(uint8_t* dst, size_t cap, const zseq* seq, size_t nseq, uint8_t* llc, uint8_t* ofc, uint8_t* mlc, uint32_t strat,
                                const zentropy* pe, zentropy* ne)
{
    size_t o = 0, last_count = 0;
    fse_ct* const ll = &ne->fse[0]; fse_ct* const of = &ne->fse[1]; fse_ct* const ml = &ne->fse[2];
    uint32_t count[64];
    if (cap < 4) return ERR_TOOSMALL;
    if (nseq < 128) dst[o++] = (uint8_t)nseq;
    else if (nseq < 0x7F00) { dst[o++] = (uint8_t)((nseq >> 8) + 0x80); dst[o++] = (uint8_t)nseq; }
    else { dst[o++] = 0xFF; dst[o++] = (uint8_t)(nseq - 0x7F00); dst[o++] = (uint8_t)((nseq - 0x7F00) >> 8); }
    if (!nseq) { memcpy(ne->fse, pe->fse, sizeof ne->fse); memcpy(ne->fse_repeat, pe->fse_repeat, sizeof ne->fse_repeat); return (int64_t)o; }
    for (size_t i = 0; i < nseq; i++) { llc[i] = (uint8_t)ll_code(seq[i].ll); ofc[i] = (uint8_t)hibit(seq[i].off); mlc[i] = (uint8_t)ml_code(seq[i].ml); }
    {
        uint8_t* const head = dst + o++;
        int tll, tof, tml, r;
        unsigned max;
        size_t most;
        max = 35; most = hist(count, &max, llc, nseq);
        ne->fse_repeat[0] = pe->fse_repeat[0];
        tll = select_type(&ne->fse_repeat[0], count, max, most, nseq, 9, &pe->fse[0], kLLNorm, 6, 1, strat);
        r = build_seq_table(dst + o, cap - o, ll, &pe->fse[0], 9, tll, count, max, llc, nseq, kLLNorm, 6, 35);
        if (r < 0) return r;
        if (tll == 2) last_count = (size_t)r;
        o += (size_t)r;
        max = 31; most = hist(count, &max, ofc, nseq);
        ne->fse_repeat[1] = pe->fse_repeat[1];
        tof = select_type(&ne->fse_repeat[1], count, max, most, nseq, 8, &pe->fse[1], kOFNorm, 5, max <= 28, strat);
        r = build_seq_table(dst + o, cap - o, of, &pe->fse[1], 8, tof, count, max, ofc, nseq, kOFNorm, 5, 28);
        if (r < 0) return r;
        if (tof == 2) last_count = (size_t)r;
        o += (size_t)r;
        max = 52; most = hist(count, &max, mlc, nseq);
        ne->fse_repeat[2] = pe->fse_repeat[2];
        tml = select_type(&ne->fse_repeat[2], count, max, most, nseq, 9, &pe->fse[2], kMLNorm, 6, 1, strat);
        r = build_seq_table(dst + o, cap - o, ml, &pe->fse[2], 9, tml, count, max, mlc, nseq, kMLNorm, 6, 52);
        if (r < 0) return r;
        if (tml == 2) last_count = (size_t)r;
        o += (size_t)r;
        *head = (uint8_t)((tll << 6) + (tof << 4) + (tml << 2));      /* set_basic 0, set_rle 1, set_compressed 2, set_repeat 3 */
    }
    {   /* ZSTD_encodeSequences_body: last sequence first */
        bitw w;
        size_t i = nseq - 1, bytes;
        uint32_t sml, sof, sll;
        if (!bw_init(&w, dst + o, cap - o)) return ERR_TOOSMALL;
        sml = fse_first_state(ml, mlc[i]); sof = fse_first_state(of, ofc[i]); sll = fse_first_state(ll, llc[i]);
        bw_put(&w, seq[i].ll, kLLBits[llc[i]]);
        bw_put(&w, seq[i].ml, kMLBits[mlc[i]]);
        bw_put(&w, seq[i].off, ofc[i]);
        while (i-- > 0) {
            sof = fse_encode(&w, of, sof, ofc[i]);
            sml = fse_encode(&w, ml, sml, mlc[i]);
            sll = fse_encode(&w, ll, sll, llc[i]);
            bw_put(&w, seq[i].ll, kLLBits[llc[i]]);
            bw_put(&w, seq[i].ml, kMLBits[mlc[i]]);
            bw_put(&w, seq[i].off, ofc[i]);
        }
        bw_put(&w, sml, ml->log); bw_put(&w, sof, of->log); bw_put(&w, sll, ll->log);
        bytes = bw_close(&w);
        if (!bytes) return ERR_TOOSMALL;
        o += bytes;
        if (last_count && last_count + bytes < 4) return 0;      /* zstd <= 1.3.4 decoder workaround (:2737-2744) */
    }
    return (int64_t)o;
}

/* ------------------------------------------------------------------------------------------------ match finder */
typedef struct {
    uint32_t* table;              /* fast: the hash table; dfast: the long (8-byte) hash table */
    uint32_t* small;              /* dfast: the short hash table; lazy with hash chains: the chain table */
    uint8_t*  tags;               /* row match finder: tag table, 2 bytes per hash-table entry (head byte + tags at +16) */
    uint32_t  hash_cache[8], next_to_update, low_limit, dict_limit;   /* window.lowLimit / dictLimit as indices */
    zparams   p;
    zseq*     seq;  size_t nseq;
    uint8_t*  lit;  size_t nlit;
} zmatch;

static uint32_t zhash(const uint8_t* p, uint32_t hlog, uint32_t mls)
{
    switch (mls) {
    case 5:  return (uint32_t)(((rd64(p) << 24) * 889523592379ull) >> (64 - hlog));
    case 6:  return (uint32_t)(((rd64(p) << 16) * 227718039650203ull) >> (64 - hlog));
    case 7:  return (uint32_t)(((rd64(p) << 8) * 58295818150454627ull) >> (64 - hlog));
    default: return (rd32(p) * 2654435761u) >> (32 - hlog);
    }
}

static size_t count_eq(const uint8_t* s, size_t a, size_t b, size_t end)
{
    const size_t a0 = a;
    while (a < end && s[a] == s[b]) { a++; b++; }
    return a - a0;
}

static void store_seq(zmatch* m, const uint8_t* s, size_t anchor, size_t ll, uint32_t off_base, size_t ml)
{
    memcpy(m->lit + m->nlit, s + anchor, ll); m->nlit += ll;
    m->seq[m->nseq].ll = (uint32_t)ll; m->seq[m->nseq].ml = (uint32_t)(ml - 3); m->seq[m->nseq].off = off_base;
    m->nseq++;
}

/* ZSTD_compressBlock_fast_noDict_generic over s[start, end); returns the last-literals length */
static size_t fast_block(zmatch* m, uint32_t rep[3], const uint8_t* s, size_t start, size_t end)
{
    uint32_t* const tab = m->table;
    const uint32_t hlog = m->p.hlog, wsize = 1u << m->p.wlog;
    const uint32_t mls = (m->p.mml >= 5 && m->p.mml <= 7) ? m->p.mml : 4;
    const size_t step0 = m->p.tlen > 1 ? (size_t)m->p.tlen + 1 : 2;
    const uint32_t end_idx = (uint32_t)end + 2;
    const uint32_t prefix_idx = end_idx - 2 > wsize ? end_idx - wsize : 2;     /* dictLimit after enforceMaxDist */
    const size_t prefix = prefix_idx - 2;
    const int64_t ilimit = (int64_t)end - 8;       /* may be negative for a 7-byte input: compared as signed */
    size_t anchor = start, ip0 = start, ip1, ip2, ip3, step, next_step, match0, mlen;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0, cur0 = 0, idx, off_base;
    uint32_t h0, h1;

    ip0 += (ip0 == prefix);
    {
        const uint32_t cur = (uint32_t)ip0 + 2;
        const uint32_t low = cur - prefix_idx > wsize ? cur - wsize : prefix_idx;
        const uint32_t max_rep = cur - low;
        if (rep2 > max_rep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > max_rep) { saved1 = rep1; rep1 = 0; }
    }
    for (;;) {                                     /* _start */
        step = step0; next_step = ip0 + 128;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if ((int64_t)ip3 >= ilimit) break;
        h0 = zhash(s + ip0, hlog, mls); h1 = zhash(s + ip1, hlog, mls);
        idx = tab[h0];
        for (;;) {
            const uint32_t rval = rd32(s + ip2 - rep1);
            cur0 = (uint32_t)ip0 + 2;
            tab[h0] = cur0;
            if ((rd32(s + ip2) == rval) & (rep1 > 0)) {
                ip0 = ip2; match0 = ip0 - rep1;
                mlen = s[ip0 - 1] == s[match0 - 1];
                ip0 -= mlen; match0 -= mlen;
                off_base = 1; mlen += 4;
                tab[h1] = (uint32_t)ip1 + 2;
                goto match;
            }
            if (idx >= prefix_idx && rd32(s + idx - 2) == rd32(s + ip0)) { tab[h1] = (uint32_t)ip1 + 2; goto offset; }
            idx = tab[h1]; h0 = h1; h1 = zhash(s + ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            cur0 = (uint32_t)ip0 + 2;
            tab[h0] = cur0;
            if (idx >= prefix_idx && rd32(s + idx - 2) == rd32(s + ip0)) { if (step <= 4) tab[h1] = (uint32_t)ip1 + 2; goto offset; }
            idx = tab[h1]; h0 = h1; h1 = zhash(s + ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            if (ip2 >= next_step) { step++; next_step += 128; }
            if ((int64_t)ip3 >= ilimit) goto cleanup;
        }
    offset:
        match0 = idx - 2;
        rep2 = rep1; rep1 = (uint32_t)(ip0 - match0);
        off_base = rep1 + 3;
        mlen = 4;
        while (ip0 > anchor && match0 > prefix && s[ip0 - 1] == s[match0 - 1]) { ip0--; match0--; mlen++; }
    match:
        mlen += count_eq(s, ip0 + mlen, match0 + mlen, end);
        store_seq(m, s, anchor, ip0 - anchor, off_base, mlen);
        ip0 += mlen; anchor = ip0;
        if ((int64_t)ip0 <= ilimit) {
            tab[zhash(s + cur0, hlog, mls)] = cur0 + 2;                  /* position cur0 - 2 + 2 */
            tab[zhash(s + ip0 - 2, hlog, mls)] = (uint32_t)ip0;          /* index of ip0 - 2 */
            if (rep2 > 0)
                while ((int64_t)ip0 <= ilimit && rd32(s + ip0) == rd32(s + ip0 - rep2)) {
                    const size_t rlen = count_eq(s, ip0 + 4, ip0 + 4 - rep2, end) + 4;
                    const uint32_t t = rep2; rep2 = rep1; rep1 = t;
                    tab[zhash(s + ip0, hlog, mls)] = (uint32_t)ip0 + 2;
                    ip0 += rlen;
                    store_seq(m, s, anchor, 0, 1, rlen);
                    anchor = ip0;
                }
        }
    }
cleanup:
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return end - anchor;
}


/* ZSTD_compressBlock_doubleFast_noDict_generic (compress/zstd_double_fast.c:98-330) over s[start, end) */
static size_t dfast_block(zmatch* m, uint32_t rep[3], const uint8_t* s, size_t start, size_t end)
{
    uint32_t* const tl = m->table; uint32_t* const ts = m->small;
    const uint32_t hl_log = m->p.hlog, hs_log = m->p.clog, wsize = 1u << m->p.wlog;
    const uint32_t mls = (m->p.mml >= 5 && m->p.mml <= 7) ? m->p.mml : 4;
    const uint32_t end_idx = (uint32_t)end + 2;
    const uint32_t prefix_idx = end_idx - 2 > wsize ? end_idx - wsize : 2;
    const size_t prefix = prefix_idx - 2;
    const int64_t ilimit = (int64_t)end - 8;
    size_t anchor = start, ip = start, ip1, step, next_step, mlen, match;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0, cur = 0, offset, idxl0, idxl1, hl0, hl1;
#define HL(p_) ((uint32_t)((rd64(s + (p_)) * 0xCF1BBCDCB7A56463ull) >> (64 - hl_log)))
    ip += (ip == prefix);
    {
        const uint32_t c = (uint32_t)ip + 2;
        const uint32_t low = c - prefix_idx > wsize ? c - wsize : prefix_idx;
        const uint32_t max_rep = c - low;
        if (rep2 > max_rep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > max_rep) { saved1 = rep1; rep1 = 0; }
    }
    for (;;) {
        step = 1; next_step = ip + 256; ip1 = ip + step;
        if ((int64_t)ip1 > ilimit) goto cleanup;
        hl0 = HL(ip); idxl0 = tl[hl0];
        for (;;) {
            const uint32_t hs0 = zhash(s + ip, hs_log, mls);
            const uint32_t idxs0 = ts[hs0];
            cur = (uint32_t)ip + 2;
            tl[hl0] = ts[hs0] = cur;
            if ((rep1 > 0) & (rd32(s + ip + 1 - rep1) == rd32(s + ip + 1))) {
                mlen = count_eq(s, ip + 1 + 4, ip + 1 + 4 - rep1, end) + 4;
                ip++;
                store_seq(m, s, anchor, ip - anchor, 1, mlen);
                goto stored;
            }
            hl1 = HL(ip1);
            if (idxl0 > prefix_idx && rd64(s + idxl0 - 2) == rd64(s + ip)) {
                match = idxl0 - 2;
                mlen = count_eq(s, ip + 8, match + 8, end) + 8;
                goto found;
            }
            idxl1 = tl[hl1];
            if (idxs0 > prefix_idx && rd32(s + idxs0 - 2) == rd32(s + ip)) {
                if (idxl1 > prefix_idx && rd64(s + idxl1 - 2) == rd64(s + ip1)) {      /* _search_next_long */
                    ip = ip1; match = idxl1 - 2;
                    mlen = count_eq(s, ip + 8, match + 8, end) + 8;
                } else {
                    match = idxs0 - 2;
                    mlen = count_eq(s, ip + 4, match + 4, end) + 4;
                }
                goto found;
            }
            if (ip1 >= next_step) { step++; next_step += 256; }
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1;
            if ((int64_t)ip1 > ilimit) goto cleanup;
        }
    found:
        offset = (uint32_t)(ip - match);
        while (ip > anchor && match > prefix && s[ip - 1] == s[match - 1]) { ip--; match--; mlen++; }
        rep2 = rep1; rep1 = offset;
        if (step < 4) tl[hl1] = (uint32_t)ip1 + 2;
        store_seq(m, s, anchor, ip - anchor, offset + 3, mlen);
    stored:
        ip += mlen; anchor = ip;
        if ((int64_t)ip <= ilimit) {
            const uint32_t ins = cur + 2;                               /* index; its position is ins - 2 = cur */
            tl[HL(cur)] = ins;
            tl[HL(ip - 2)] = (uint32_t)ip;
            ts[zhash(s + cur, hs_log, mls)] = ins;
            ts[zhash(s + ip - 1, hs_log, mls)] = (uint32_t)ip + 1;
            while ((int64_t)ip <= ilimit && ((rep2 > 0) & (rd32(s + ip) == rd32(s + ip - rep2)))) {
                const size_t rlen = count_eq(s, ip + 4, ip + 4 - rep2, end) + 4;
                const uint32_t t = rep2; rep2 = rep1; rep1 = t;
                ts[zhash(s + ip, hs_log, mls)] = (uint32_t)ip + 2;
                tl[HL(ip)] = (uint32_t)ip + 2;
                store_seq(m, s, anchor, 0, 1, rlen);
                ip += rlen; anchor = ip;
            }
        }
    }
cleanup:
#undef HL
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return end - anchor;
}

/* ------------------------------------------------------------------------------------------------ lazy parsers
 * ZSTD_compressBlock_lazy_generic (compress/zstd_lazy.c:1486-1737, noDict) with the row-hash match finder
 * (ZSTD_RowFindBestMatch :1139-1250, row update :885-946, hash cache :842-883) or the hash-chain one
 * (ZSTD_HcFindBestMatch :649-730).  Which one: ZSTD_resolveRowMatchFinderMode (zstd_compress.c:232-248). */
static uint32_t lz_mls(const zmatch* m) { return m->p.mml < 4 ? 4 : m->p.mml > 6 ? 6 : m->p.mml; }
static uint32_t lz_mls3(const zmatch* m) { return m->p.mml < 3 ? 3 : m->p.mml > 6 ? 6 : m->p.mml; }   /* the bt of the optimal parser: 3..6 (zstd_opt.c:862) */
static uint32_t lz_rowlog(const zmatch* m) { return m->p.slog < 4 ? 4 : m->p.slog > 6 ? 6 : m->p.slog; }
static uint32_t row_hash(const zmatch* m, const uint8_t* s, uint32_t idx)
{ return zhash(s + idx - 2, m->p.hlog - lz_rowlog(m) + 8, lz_mls(m)); }

static void row_fill_cache(zmatch* m, const uint8_t* s, uint32_t idx, int64_t limit_pos)
{
    const int64_t at = (int64_t)idx - 2;
    const uint32_t n = at > limit_pos ? 0 : (uint32_t)(limit_pos - at + 1);
    const uint32_t lim = idx + (n < 8 ? n : 8);
    for (; idx < lim; idx++) m->hash_cache[idx & 7] = row_hash(m, s, idx);
}

static uint32_t row_next_hash(zmatch* m, const uint8_t* s, uint32_t idx)
{
    const uint32_t fresh = row_hash(m, s, idx + 8), h = m->hash_cache[idx & 7];
    m->hash_cache[idx & 7] = fresh;
    return h;
}

static void row_insert(zmatch* m, uint32_t hash, uint32_t idx)
{
    const uint32_t rowlog = lz_rowlog(m), mask = (1u << rowlog) - 1, rel = (hash >> 8) << rowlog;
    uint8_t* const tag_row = m->tags + 2 * (size_t)rel;
    const uint32_t pos = (tag_row[0] - 1u) & mask;
    tag_row[0] = (uint8_t)pos;
    tag_row[16 + pos] = (uint8_t)hash;
    m->table[rel + pos] = idx;
}

static void row_update(zmatch* m, const uint8_t* s, uint32_t target)         /* ZSTD_row_update_internal, useCache = 1 */
{
    uint32_t idx = m->next_to_update;
    if (target - idx > 384) {
        for (const uint32_t bound = idx + 96; idx < bound; idx++) row_insert(m, row_next_hash(m, s, idx), idx);
        idx = target - 32;
        row_fill_cache(m, s, idx, (int64_t)target - 2 + 1);
    }
    for (; idx < target; idx++) row_insert(m, row_next_hash(m, s, idx), idx);
    m->next_to_update = target;
}

static uint32_t lz_low_limit(const zmatch* m, uint32_t curr)
{
    const uint32_t max_dist = 1u << m->p.wlog;
    return curr - m->low_limit > max_dist ? curr - max_dist : m->low_limit;
}

static size_t row_search(zmatch* m, const uint8_t* s, size_t ip, size_t end, uint32_t* ofb)
{
    const uint32_t curr = (uint32_t)ip + 2, low = lz_low_limit(m, curr);
    const uint32_t rowlog = lz_rowlog(m), entries = 1u << rowlog, mask = entries - 1;
    uint32_t attempts = 1u << (m->p.slog < rowlog ? m->p.slog : rowlog);
    uint32_t cand[64], ncand = 0;
    size_t ml = 3;
    row_update(m, s, curr);
    {
        const uint32_t hash = row_next_hash(m, s, curr), rel = (hash >> 8) << rowlog, tag = hash & 255;
        uint8_t* const tag_row = m->tags + 2 * (size_t)rel;
        const uint32_t head = tag_row[0] & mask;
        for (uint32_t i = 0; i < entries && attempts > 0; i++) {        /* newest first: the rotated SSE match mask */
            const uint32_t pos = (head + i) & mask;
            if (tag_row[16 + pos] != tag) continue;
            if (m->table[rel + pos] < low) break;
            cand[ncand++] = m->table[rel + pos];
            attempts--;
        }
        {   /* the current position goes in as well (:1229-1234) */
            const uint32_t pos = (tag_row[0] - 1u) & mask;
            tag_row[0] = (uint8_t)pos; tag_row[16 + pos] = (uint8_t)tag;
            m->table[rel + pos] = m->next_to_update++;
        }
    }
    for (uint32_t k = 0; k < ncand; k++) {
        const size_t match = cand[k] - 2;
        size_t cur = 0;
        if (s[match + ml] == s[ip + ml]) cur = count_eq(s, ip, match, end);
        if (cur > ml) { ml = cur; *ofb = curr - cand[k] + 3; if (ip + cur == end) break; }
    }
    return ml;
}

static size_t hc_search(zmatch* m, const uint8_t* s, size_t ip, size_t end, uint32_t* ofb)
{
    const uint32_t curr = (uint32_t)ip + 2, low = lz_low_limit(m, curr), mls = lz_mls(m);
    const uint32_t chain_size = 1u << m->p.clog, cmask = chain_size - 1;
    const uint32_t min_chain = curr > chain_size ? curr - chain_size : 0;
    uint32_t attempts = 1u << m->p.slog, idx = m->next_to_update, mi;
    size_t ml = 3;
    for (; idx < curr; idx++) {                                       /* ZSTD_insertAndFindFirstIndex_internal */
        const uint32_t h = zhash(s + idx - 2, m->p.hlog, mls);
        m->small[idx & cmask] = m->table[h];
        m->table[h] = idx;
    }
    m->next_to_update = curr;
    mi = m->table[zhash(s + ip, m->p.hlog, mls)];
    for (; (mi >= low) & (attempts > 0); attempts--) {
        const size_t match = mi - 2;
        size_t cur = 0;
        if (s[match + ml] == s[ip + ml]) cur = count_eq(s, ip, match, end);
        if (cur > ml) { ml = cur; *ofb = curr - mi + 3; if (ip + cur == end) break; }
        if (mi <= min_chain) break;
        mi = m->small[mi & cmask];
    }
    return ml;
}

/* Binary-tree match finder of btlazy2 (compress/zstd_lazy.c:20-58 ZSTD_updateDUBT, :64-150 ZSTD_insertDUBT1, :231-379
 * ZSTD_DUBT_findBestMatch, :383-392 ZSTD_BtFindBestMatch; noDict).  m->small is the tree: two links per index. */
#define DUBT_UNSORTED 1u
static void bt_insert1(zmatch* m, const uint8_t* s, uint32_t curr, size_t end, uint32_t nb_compares, uint32_t bt_low)
{
    uint32_t* const bt = m->small;
    const uint32_t bt_mask = (1u << (m->p.clog - 1)) - 1, max_dist = 1u << m->p.wlog;
    const uint32_t window_low = curr - m->low_limit > max_dist ? curr - max_dist : m->low_limit;
    const size_t ip = curr - 2;
    size_t common_smaller = 0, common_larger = 0;
    uint32_t dummy, *smaller = bt + 2 * (curr & bt_mask), *larger = smaller + 1, mi = *smaller;
    for (; nb_compares && mi > window_low; --nb_compares) {
        uint32_t* const next = bt + 2 * (mi & bt_mask);
        const size_t match = mi - 2;
        size_t ml = common_smaller < common_larger ? common_smaller : common_larger;
        ml += count_eq(s, ip + ml, match + ml, end);
        if (ip + ml == end) break;                                    /* equal: no way to know if smaller or larger */
        if (s[match + ml] < s[ip + ml]) {
            *smaller = mi; common_smaller = ml;
            if (mi <= bt_low) { smaller = &dummy; break; }
            smaller = next + 1; mi = next[1];
        } else {
            *larger = mi; common_larger = ml;
            if (mi <= bt_low) { larger = &dummy; break; }
            larger = next; mi = next[0];
        }
    }
    *smaller = *larger = 0;
}

static size_t bt_search(zmatch* m, const uint8_t* s, size_t ip, size_t end, uint32_t* ofb)
{
    uint32_t* const bt = m->small;
    const uint32_t curr = (uint32_t)ip + 2, mls = lz_mls(m), bt_mask = (1u << (m->p.clog - 1)) - 1;
    uint32_t idx, h, mi, nb_compares, nb_candidates, previous = 0, *next_cand, *unsorted;
    uint32_t window_low, bt_low, unsort_limit;
    if (curr < m->next_to_update) return 0;                           /* skipped area */
    for (idx = m->next_to_update; idx < curr; idx++) {                /* ZSTD_updateDUBT: chain the new positions in, unsorted */
        const uint32_t hh = zhash(s + idx - 2, m->p.hlog, mls);
        bt[2 * (idx & bt_mask)] = m->table[hh];
        bt[2 * (idx & bt_mask) + 1] = DUBT_UNSORTED;
        m->table[hh] = idx;
    }
    m->next_to_update = curr;
    h = zhash(s + ip, m->p.hlog, mls);
    mi = m->table[h];
    window_low = lz_low_limit(m, curr);
    bt_low = bt_mask >= curr ? 0 : curr - bt_mask;
    unsort_limit = bt_low > window_low ? bt_low : window_low;
    next_cand = bt + 2 * (mi & bt_mask); unsorted = next_cand + 1;
    nb_compares = 1u << m->p.slog; nb_candidates = nb_compares;
    while (mi > unsort_limit && *unsorted == DUBT_UNSORTED && nb_candidates > 1) {      /* reach the end of the unsorted candidates */
        *unsorted = previous; previous = mi;
        mi = *next_cand;
        next_cand = bt + 2 * (mi & bt_mask); unsorted = next_cand + 1;
        nb_candidates--;
    }
    if (mi > unsort_limit && *unsorted == DUBT_UNSORTED) *next_cand = *unsorted = 0;   /* nullify the last one if still unsorted */
    mi = previous;
    while (mi) {                                                      /* batch sort the stacked candidates */
        uint32_t* const nip = bt + 2 * (mi & bt_mask) + 1;
        const uint32_t nxt = *nip;
        bt_insert1(m, s, mi, end, nb_candidates, unsort_limit);
        mi = nxt; nb_candidates++;
    }
    {   /* find the longest match, inserting curr into the tree */
        size_t common_smaller = 0, common_larger = 0, best = 0;
        uint32_t dummy, *smaller = bt + 2 * (curr & bt_mask), *larger = smaller + 1, match_end_idx = curr + 8 + 1;
        mi = m->table[h];
        m->table[h] = curr;
        for (; nb_compares && mi > window_low; --nb_compares) {
            uint32_t* const next = bt + 2 * (mi & bt_mask);
            const size_t match = mi - 2;
            size_t ml = common_smaller < common_larger ? common_smaller : common_larger;
            ml += count_eq(s, ip + ml, match + ml, end);
            if (ml > best) {
                if (ml > match_end_idx - mi) match_end_idx = mi + (uint32_t)ml;
                if (4 * (int)(ml - best) > (int)(hibit(curr - mi + 1) - hibit(*ofb))) { best = ml; *ofb = curr - mi + 3; }
                if (ip + ml == end) break;                            /* equal: drop, to keep the tree consistent */
            }
            if (s[match + ml] < s[ip + ml]) {
                *smaller = mi; common_smaller = ml;
                if (mi <= bt_low) { smaller = &dummy; break; }
                smaller = next + 1; mi = next[1];
            } else {
                *larger = mi; common_larger = ml;
                if (mi <= bt_low) { larger = &dummy; break; }
                larger = next; mi = next[0];
            }
        }
        *smaller = *larger = 0;
        m->next_to_update = match_end_idx - 8;                        /* skip repetitive patterns */
        return best;
    }
}

static size_t lazy_block(zmatch* m, uint32_t rep[3], const uint8_t* s, size_t start, size_t end, int depth, int method /* 0 hash chains, 1 rows, 2 binary tree */)
{
    const int use_row = method == 1;
    const int64_t ilimit = (int64_t)end - 8 - (use_row ? 8 : 0);
    const uint32_t prefix_idx = m->dict_limit;
    const size_t prefix = prefix_idx - 2;
    size_t ip = start, anchor = start;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
#define SEARCH(pos_, ofb_) (method == 1 ? row_search(m, s, (pos_), end, (ofb_)) : method == 2 ? bt_search(m, s, (pos_), end, (ofb_)) : hc_search(m, s, (pos_), end, (ofb_)))
    ip += (ip == prefix);
    {
        const uint32_t c = (uint32_t)ip + 2, max_dist = 1u << m->p.wlog;
        const uint32_t low = c - m->dict_limit > max_dist ? c - max_dist : m->dict_limit;
        const uint32_t max_rep = c - low;
        if (rep2 > max_rep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > max_rep) { saved1 = rep1; rep1 = 0; }
    }
    if (use_row) row_fill_cache(m, s, m->next_to_update, ilimit);
    while ((int64_t)ip < ilimit) {
        size_t ml = 0, at = ip + 1;
        uint32_t ofb = 1;
        if ((rep1 > 0) & (rd32(s + ip + 1 - rep1) == rd32(s + ip + 1))) {
            ml = count_eq(s, ip + 1 + 4, ip + 1 + 4 - rep1, end) + 4;
            if (depth == 0) goto store;
        }
        {
            uint32_t found = 999999999;
            const size_t ml2 = SEARCH(ip, &found);
            if (ml2 > ml) { ml = ml2; at = ip; ofb = found; }
        }
        if (ml < 4) { ip += ((ip - anchor) >> 8) + 1; continue; }
        if (depth >= 1)
            while ((int64_t)ip < ilimit) {
                ip++;
                if ((rep1 > 0) & (rd32(s + ip) == rd32(s + ip - rep1))) {
                    const size_t mr = count_eq(s, ip + 4, ip + 4 - rep1, end) + 4;
                    const int g2 = (int)(mr * 3), g1 = (int)(ml * 3 - (size_t)hibit(ofb) + 1);
                    if (mr >= 4 && g2 > g1) { ml = mr; ofb = 1; at = ip; }
                }
                {
                    uint32_t cand = 999999999;
                    const size_t ml2 = SEARCH(ip, &cand);
                    const int g2 = (int)(ml2 * 4 - (size_t)hibit(cand)), g1 = (int)(ml * 4 - (size_t)hibit(ofb) + 4);
                    if (ml2 >= 4 && g2 > g1) { ml = ml2; ofb = cand; at = ip; continue; }
                }
                if (depth == 2 && (int64_t)ip < ilimit) {
                    ip++;
                    if ((rep1 > 0) & (rd32(s + ip) == rd32(s + ip - rep1))) {
                        const size_t mr = count_eq(s, ip + 4, ip + 4 - rep1, end) + 4;
                        const int g2 = (int)(mr * 4), g1 = (int)(ml * 4 - (size_t)hibit(ofb) + 1);
                        if (mr >= 4 && g2 > g1) { ml = mr; ofb = 1; at = ip; }
                    }
                    {
                        uint32_t cand = 999999999;
                        const size_t ml2 = SEARCH(ip, &cand);
                        const int g2 = (int)(ml2 * 4 - (size_t)hibit(cand)), g1 = (int)(ml * 4 - (size_t)hibit(ofb) + 7);
                        if (ml2 >= 4 && g2 > g1) { ml = ml2; ofb = cand; at = ip; continue; }
                    }
                }
                break;
            }
        if (ofb > 3) {
            const uint32_t off = ofb - 3;
            while (at > anchor && at - off > prefix && s[at - 1] == s[at - off - 1]) { at--; ml++; }
            rep2 = rep1; rep1 = off;
        }
    store:
        store_seq(m, s, anchor, at - anchor, ofb, ml);
        anchor = ip = at + ml;
        while (((int64_t)ip <= ilimit) & (rep2 > 0) && rd32(s + ip) == rd32(s + ip - rep2)) {
            const uint32_t t = rep2;
            ml = count_eq(s, ip + 4, ip + 4 - rep2, end) + 4;
            rep2 = rep1; rep1 = t;
            store_seq(m, s, anchor, 0, 1, ml);
            ip += ml; anchor = ip;
        }
    }
#undef SEARCH
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return end - anchor;
}

/* ------------------------------------------------------------------------------------------------ optimal parser */
/* ZSTD_btopt (compress/zstd_opt.c, optLevel 0): zstd level 12 for inputs of 16 KiB and less (clevels.h:118), which are
 * one block, so only the first-block statistics (zstd_opt.c:191-226) are restated.  Prices are in 1/256 bit. */
#define OPT_NUM   4096
#define OPT_MAXP  (1 << 30)
#define OPT_BIT   256
typedef struct { int price; uint32_t off, mlen, litlen, rep[3]; } optnode;
typedef struct { uint32_t off, len; } optmatch;
typedef struct {
    uint32_t lit_freq[256], ll_freq[36], ml_freq[53], of_freq[32];
    uint32_t lit_sum, ll_sum, ml_sum, of_sum, lit_base, ll_base, ml_base, of_base;
    int predef;
    optnode* node; optmatch* match;
    uint32_t* hash3; uint32_t hlog3, next3;
} optstate;

static uint32_t opt_weight(uint32_t stat) { return (uint32_t)hibit(stat + 1) * OPT_BIT; }       /* ZSTD_bitWeight :40-43 */
static void opt_base_prices(optstate* o)                                                         /* :71-78 */
{ o->lit_base = opt_weight(o->lit_sum); o->ll_base = opt_weight(o->ll_sum); o->ml_base = opt_weight(o->ml_sum); o->of_base = opt_weight(o->of_sum); }

static void opt_first_block_stats(optstate* o, const uint8_t* src, size_t n)                     /* ZSTD_rescaleFreqs :123-240, no dictionary */
{
    static const uint8_t of0[11] = {6, 2, 1, 1, 2, 3, 4, 4, 4, 3, 2};
    unsigned i;
    o->predef = n <= 1024;
    memset(o->lit_freq, 0, sizeof o->lit_freq);
    for (size_t k = 0; k < n; k++) o->lit_freq[src[k]]++;
    o->lit_sum = 0;
    for (i = 0; i < 256; i++) { o->lit_freq[i] = 1 + (o->lit_freq[i] >> 8); o->lit_sum += o->lit_freq[i]; }
    o->ll_sum = 0;
    for (i = 0; i < 36; i++) { o->ll_freq[i] = i == 0 ? 4 : i == 1 ? 2 : 1; o->ll_sum += o->ll_freq[i]; }
    for (i = 0; i < 53; i++) o->ml_freq[i] = 1;
    o->ml_sum = 53;
    o->of_sum = 0;
    for (i = 0; i < 32; i++) { o->of_freq[i] = i < 11 ? of0[i] : 1; o->of_sum += o->of_freq[i]; }
    opt_base_prices(o);
}

static uint32_t opt_literal_price(const optstate* o, uint8_t b)                                  /* ZSTD_rawLiteralsCost :245-269, one literal */
{
    uint32_t w;
    if (o->predef) return 6 * OPT_BIT;
    w = opt_weight(o->lit_freq[b]);
    if (w > o->lit_base - OPT_BIT) w = o->lit_base - OPT_BIT;
    return o->lit_base - w;
}
static uint32_t opt_ll_price(const optstate* o, uint32_t ll)                                     /* ZSTD_litLengthPrice :273-292 */
{
    if (o->predef) return opt_weight(ll);
    if (ll == BLOCK_MAX) return OPT_BIT + opt_ll_price(o, BLOCK_MAX - 1);
    { const unsigned c = ll_code(ll); return kLLBits[c] * OPT_BIT + o->ll_base - opt_weight(o->ll_freq[c]); }
}
static uint32_t opt_match_price(const optstate* o, uint32_t off_base, uint32_t mlen)             /* ZSTD_getMatchPrice :300-328 */
{
    const uint32_t ofc = (uint32_t)hibit(off_base), mlb = mlen - 3;
    uint32_t price;
    if (o->predef) return opt_weight(mlb) + (16 + ofc) * OPT_BIT;
    price = ofc * OPT_BIT + (o->of_base - opt_weight(o->of_freq[ofc]));
    if (ofc >= 20) price += (ofc - 19) * 2 * OPT_BIT;
    { const unsigned c = ml_code(mlb); price += kMLBits[c] * OPT_BIT + (o->ml_base - opt_weight(o->ml_freq[c])); }
    return price + OPT_BIT / 5;
}
static void opt_count_sequence(optstate* o, uint32_t ll, const uint8_t* lits, uint32_t off_base, uint32_t mlen)   /* ZSTD_updateStats :332-363 */
{
    for (uint32_t u = 0; u < ll; u++) o->lit_freq[lits[u]] += 2;
    o->lit_sum += 2 * ll;
    o->ll_freq[ll_code(ll)]++; o->ll_sum++;
    o->of_freq[hibit(off_base)]++; o->of_sum++;
    o->ml_freq[ml_code(mlen - 3)]++; o->ml_sum++;
}
static void opt_new_rep(uint32_t out[3], const uint32_t rep[3], uint32_t off_base, uint32_t ll0)                  /* ZSTD_newRep, zstd_compress_internal.h:676-706 */
{
    uint32_t r0 = rep[0], r1 = rep[1], r2 = rep[2];
    if (off_base > 3) { r2 = r1; r1 = r0; r0 = off_base - 3; }
    else {
        const uint32_t code = off_base - 1 + ll0;
        if (code > 0) { const uint32_t v = code == 3 ? r0 - 1 : code == 1 ? r1 : r2; if (code >= 2) r2 = r1; r1 = r0; r0 = v; }
    }
    out[0] = r0; out[1] = r1; out[2] = r2;
}
static uint32_t opt_hash3(const uint8_t* p, uint32_t hlog) { return ((rd32(p) << 8) * 506832829u) >> (32 - hlog); }  /* ZSTD_hash3Ptr */
static uint32_t opt_rd3(const uint8_t* p, uint32_t minmatch) { return minmatch == 3 ? rd32(p) << 8 : rd32(p); }        /* ZSTD_readMINMATCH :369-380 */

/* ZSTD_insertBt1 (:414-530): index curr enters the tree; returns how many positions to move on */
static uint32_t opt_tree_insert(zmatch* m, const uint8_t* s, uint32_t curr, size_t end, uint32_t target)
{
    uint32_t* const bt = m->small;
    const uint32_t bt_mask = (1u << (m->p.clog - 1)) - 1, bt_low = bt_mask >= curr ? 0 : curr - bt_mask;
    const uint32_t window_low = lz_low_limit(m, target);
    const size_t ip = curr - 2;
    const uint32_t h = zhash(s + ip, m->p.hlog, lz_mls3(m));
    uint32_t mi = m->table[h], smaller = 2 * (curr & bt_mask), larger = smaller + 1, end_idx = curr + 8 + 1, left = 1u << m->p.slog;
    size_t common_s = 0, common_l = 0, best = 8;
    m->table[h] = curr;
    for (; left && mi >= window_low; --left) {
        const uint32_t next = 2 * (mi & bt_mask);
        const size_t match = mi - 2;
        size_t ml = common_s < common_l ? common_s : common_l;
        ml += count_eq(s, ip + ml, match + ml, end);
        if (ml > best) { best = ml; if (ml > end_idx - mi) end_idx = mi + (uint32_t)ml; }
        if (ip + ml == end) break;
        if (s[match + ml] < s[ip + ml]) {
            if (smaller != ~0u) bt[smaller] = mi;
            common_s = ml;
            if (mi <= bt_low) { smaller = ~0u; break; }
            smaller = next + 1; mi = bt[next + 1];
        } else {
            if (larger != ~0u) bt[larger] = mi;
            common_l = ml;
            if (mi <= bt_low) { larger = ~0u; break; }
            larger = next; mi = bt[next];
        }
    }
    if (smaller != ~0u) bt[smaller] = 0;
    if (larger != ~0u) bt[larger] = 0;
    {   uint32_t positions = 0;
        if (best > 384) positions = best - 384 < 192 ? (uint32_t)(best - 384) : 192;
        return positions > end_idx - (curr + 8) ? positions : end_idx - (curr + 8);
    }
}

/* ZSTD_btGetAllMatches (:798-816) = ZSTD_updateTree_internal (:533-552) + ZSTD_insertBtAndGetAllMatches (:559-786), noDict:
 * every match at ip longer than the ones before it, shortest first; ip enters the tree */
static uint32_t opt_matches(zmatch* m, optstate* o, const uint8_t* s, size_t ip, size_t end, const uint32_t rep[3], uint32_t ll0, uint32_t to_beat)
{
    const uint32_t curr = (uint32_t)ip + 2, mls = lz_mls3(m), minmatch = mls == 3 ? 3 : 4;
    const uint32_t sufficient = m->p.tlen < OPT_NUM - 1 ? m->p.tlen : OPT_NUM - 1;
    uint32_t* const bt = m->small;
    optmatch* const out = o->match;
    uint32_t n = 0;
    size_t best = to_beat - 1;
    if (curr < m->next_to_update) return 0;                                      /* skipped area */
    for (uint32_t idx = m->next_to_update; idx < curr; ) idx += opt_tree_insert(m, s, idx, end, curr);
    m->next_to_update = curr;
    {
        const uint32_t bt_mask = (1u << (m->p.clog - 1)) - 1, bt_low = bt_mask >= curr ? 0 : curr - bt_mask;
        const uint32_t window_low = lz_low_limit(m, curr), match_low = window_low ? window_low : 1;
        const uint32_t h = zhash(s + ip, m->p.hlog, mls);
        uint32_t mi = m->table[h], smaller = 2 * (curr & bt_mask), larger = smaller + 1, end_idx = curr + 8 + 1, left = 1u << m->p.slog;
        size_t common_s = 0, common_l = 0;
        /* repeat offsets */
        for (uint32_t code = ll0; code < 3 + ll0; code++) {
            const uint32_t off = code == 3 ? rep[0] - 1 : rep[code];
            uint32_t len = 0;
            if (off - 1 < curr - m->dict_limit) {                                /* 1 <= off <= distance to the prefix start */
                if (curr - off >= window_low && opt_rd3(s + ip, minmatch) == opt_rd3(s + ip - off, minmatch))
                    len = (uint32_t)count_eq(s, ip + minmatch, ip + minmatch - off, end) + minmatch;
            }
            if (len > best) {
                best = len;
                out[n].off = code - ll0 + 1; out[n].len = len; n++;
                if (len > sufficient || ip + len == end) return n;
            }
        }
        /* 3-byte matches through their own hash table (:385-404, :659-688) */
        if (mls == 3 && best < mls) {
            uint32_t i3;
            for (uint32_t idx = o->next3; idx < curr; idx++) o->hash3[opt_hash3(s + idx - 2, o->hlog3)] = idx;
            o->next3 = curr;
            i3 = o->hash3[opt_hash3(s + ip, o->hlog3)];
            if (i3 >= match_low && curr - i3 < (1u << 18)) {
                const size_t len = count_eq(s, ip, i3 - 2, end);
                if (len >= mls) {
                    best = len;
                    out[0].off = curr - i3 + 3; out[0].len = (uint32_t)len; n = 1;
                    if (len > sufficient || ip + len == end) { m->next_to_update = curr + 1; return 1; }
                }
            }
        }
        m->table[h] = curr;
        for (; left && mi >= match_low; --left) {
            const uint32_t next = 2 * (mi & bt_mask);
            const size_t match = mi - 2;
            size_t ml = common_s < common_l ? common_s : common_l;
            ml += count_eq(s, ip + ml, match + ml, end);
            if (ml > best) {
                if (ml > end_idx - mi) end_idx = mi + (uint32_t)ml;
                best = ml;
                out[n].off = curr - mi + 3; out[n].len = (uint32_t)ml; n++;
                if (ml > OPT_NUM || ip + ml == end) break;                       /* equal to the end: no order, keep the tree consistent */
            }
            if (s[match + ml] < s[ip + ml]) {
                if (smaller != ~0u) bt[smaller] = mi;
                common_s = ml;
                if (mi <= bt_low) { smaller = ~0u; break; }
                smaller = next + 1; mi = bt[next + 1];
            } else {
                if (larger != ~0u) bt[larger] = mi;
                common_l = ml;
                if (mi <= bt_low) { larger = ~0u; break; }
                larger = next; mi = bt[next];
            }
        }
        if (smaller != ~0u) bt[smaller] = 0;
        if (larger != ~0u) bt[larger] = 0;
        m->next_to_update = end_idx - 8;                                         /* skip repetitive patterns */
    }
    return n;
}

/* ZSTD_compressBlock_opt_generic (:1039-1325), optLevel 0, no dictionary, no long-distance matches */
static size_t opt_block(zmatch* m, optstate* o, uint32_t rep[3], const uint8_t* s, size_t start, size_t end)
{
    optnode* const node = o->node;
    const optmatch* const match = o->match;
    const int64_t ilimit = (int64_t)end - 8;
    const uint32_t sufficient = m->p.tlen < OPT_NUM - 1 ? m->p.tlen : OPT_NUM - 1, minmatch = m->p.mml == 3 ? 3 : 4;
    size_t ip = start, anchor = start;
    o->next3 = m->next_to_update;
    opt_first_block_stats(o, s + start, end - start);
    ip += (uint32_t)ip + 2 == m->dict_limit;
    while ((int64_t)ip < ilimit) {
        uint32_t cur, last_pos = 0;
        optnode last;
        {   /* the matches at ip open a series */
            const uint32_t litlen = (uint32_t)(ip - anchor), ll0 = !litlen;
            const uint32_t nb = opt_matches(m, o, s, ip, end, rep, ll0, minmatch);
            uint32_t pos, lits_price;
            if (!nb) { ip++; continue; }
            memcpy(node[0].rep, rep, sizeof node[0].rep);
            node[0].mlen = 0; node[0].litlen = litlen; node[0].price = (int)opt_ll_price(o, litlen);
            if (match[nb - 1].len > sufficient) {                                /* long match: taken at once */
                last.litlen = litlen; last.mlen = match[nb - 1].len; last.off = match[nb - 1].off; last.price = 0;
                cur = 0;
                goto shortest_path;
            }
            lits_price = (uint32_t)node[0].price + opt_ll_price(o, 0);
            for (pos = 1; pos < minmatch; pos++) node[pos].price = OPT_MAXP;
            for (uint32_t k = 0; k < nb; k++)
                for (; pos <= match[k].len; pos++) {
                    node[pos].mlen = pos; node[pos].off = match[k].off; node[pos].litlen = litlen;
                    node[pos].price = (int)(lits_price + opt_match_price(o, match[k].off, pos));
                }
            last_pos = pos - 1;
        }
        for (cur = 1; cur <= last_pos; cur++) {
            const size_t inr = ip + cur;
            {   /* one more literal, if that is not dearer */
                const uint32_t litlen = node[cur - 1].mlen == 0 ? node[cur - 1].litlen + 1 : 1;
                const int price = node[cur - 1].price + (int)opt_literal_price(o, s[inr - 1]) + (int)opt_ll_price(o, litlen) - (int)opt_ll_price(o, litlen - 1);
                if (price <= node[cur].price) { node[cur].mlen = 0; node[cur].off = 0; node[cur].litlen = litlen; node[cur].price = price; }
            }
            if (node[cur].mlen != 0) opt_new_rep(node[cur].rep, node[cur - node[cur].mlen].rep, node[cur].off, node[cur].litlen == 0);
            else memcpy(node[cur].rep, node[cur - 1].rep, sizeof node[cur].rep);
            if ((int64_t)inr > ilimit) continue;                                 /* the last match starts at least 8 bytes before the end */
            if (cur == last_pos) break;
            if (node[cur + 1].price <= node[cur].price + OPT_BIT / 2) continue;  /* unpromising position */
            {
                const uint32_t ll0 = node[cur].mlen != 0, litlen = node[cur].mlen == 0 ? node[cur].litlen : 0;
                const uint32_t base = (uint32_t)node[cur].price + opt_ll_price(o, 0);
                const uint32_t nb = opt_matches(m, o, s, inr, end, node[cur].rep, ll0, minmatch);
                if (!nb) continue;
                if (match[nb - 1].len > sufficient || cur + match[nb - 1].len >= OPT_NUM) {
                    last.mlen = match[nb - 1].len; last.off = match[nb - 1].off; last.litlen = litlen; last.price = 0;
                    cur -= node[cur].mlen == 0 ? node[cur].litlen : 0;           /* may wrap: then it is the first sequence */
                    if (cur > OPT_NUM) cur = 0;
                    goto shortest_path;
                }
                for (uint32_t k = 0; k < nb; k++) {
                    const uint32_t first = k ? match[k - 1].len + 1 : minmatch;
                    for (uint32_t mlen = match[k].len; mlen >= first; mlen--) {  /* downwards */
                        const uint32_t pos = cur + mlen;
                        const int price = (int)base + (int)opt_match_price(o, match[k].off, mlen);
                        if (pos > last_pos || price < node[pos].price) {
                            while (last_pos < pos) node[++last_pos].price = OPT_MAXP;
                            node[pos].mlen = mlen; node[pos].off = match[k].off; node[pos].litlen = litlen; node[pos].price = price;
                        } else break;                                            /* optLevel 0: early abort */
                    }
                }
            }
        }
        last = node[last_pos];
        cur = last_pos > last.litlen + last.mlen ? last_pos - (last.litlen + last.mlen) : 0;
shortest_path:
        if (last.mlen != 0) opt_new_rep(rep, node[cur].rep, last.off, last.litlen == 0);
        else memcpy(rep, node[cur].rep, 3 * sizeof(uint32_t));
        {   /* walk back through the chosen arrivals, then emit them front to back */
            const uint32_t store_end = cur + 1;
            uint32_t store_start = store_end, seq_pos = cur;
            node[store_end] = last;
            while (seq_pos > 0) {
                const uint32_t back = node[seq_pos].litlen + node[seq_pos].mlen;
                store_start--;
                node[store_start] = node[seq_pos];
                seq_pos = seq_pos > back ? seq_pos - back : 0;
            }
            for (uint32_t k = store_start; k <= store_end; k++) {
                const uint32_t llen = node[k].litlen, mlen = node[k].mlen;
                if (mlen == 0) { ip = anchor + llen; continue; }                 /* trailing literals: the next series starts behind them */
                opt_count_sequence(o, llen, s + anchor, node[k].off, mlen);
                store_seq(m, s, anchor, llen, node[k].off, mlen);
                anchor += llen + mlen;
                ip = anchor;
            }
            opt_base_prices(o);
        }
    }
    return end - anchor;
}

/* ------------------------------------------------------------------------------------------------ frame */
/* clevels.h:25-130, levels 1..12 (4mz uses 1, 3, 6, 12; the JNI entry point passes any); strat: 1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2,
 * 6 btlazy2, 7 btopt */
static zparams level_params(int level, size_t n)
{
    static const zparams rows[12][4] = {           /* tables for > 256 KB, <= 256 KB, <= 128 KB, <= 16 KB */
        {{19, 13, 14, 1, 7, 0, 1}, {18, 13, 14, 1, 6, 0, 1}, {17, 12, 13, 1, 6, 0, 1}, {14, 14, 15, 1, 5, 0, 1}},
        {{20, 15, 16, 1, 6, 0, 1}, {18, 14, 14, 1, 5, 0, 2}, {17, 13, 15, 1, 5, 0, 1}, {14, 14, 15, 1, 4, 0, 1}},
        {{21, 16, 17, 1, 5, 0, 2}, {18, 16, 16, 1, 4, 0, 2}, {17, 15, 16, 2, 5, 0, 2}, {14, 14, 15, 2, 4, 0, 2}},
        {{21, 18, 18, 1, 5, 0, 2}, {18, 16, 17, 3, 5, 2, 3}, {17, 17, 17, 2, 4, 0, 2}, {14, 14, 14, 4, 4, 2, 3}},
        {{21, 18, 19, 3, 5, 2, 3}, {18, 17, 18, 5, 5, 2, 3}, {17, 16, 17, 3, 4, 2, 3}, {14, 14, 14, 3, 4, 4, 4}},
        {{21, 18, 19, 3, 5, 4, 4}, {18, 18, 19, 3, 5, 4, 4}, {17, 16, 17, 3, 4, 4, 4}, {14, 14, 14, 4, 4, 8, 5}},
        {{21, 19, 20, 4, 5, 8, 4}, {18, 18, 19, 4, 4, 4, 4}, {17, 16, 17, 3, 4, 8, 5}, {14, 14, 14, 6, 4, 8, 5}},
        {{21, 19, 20, 4, 5, 16, 5}, {18, 18, 19, 4, 4, 8, 5}, {17, 16, 17, 4, 4, 8, 5}, {14, 14, 14, 8, 4, 8, 5}},
        {{22, 20, 21, 4, 5, 16, 5}, {18, 18, 19, 5, 4, 8, 5}, {17, 16, 17, 5, 4, 8, 5}, {14, 15, 14, 5, 4, 8, 6}},
        {{22, 21, 22, 5, 5, 16, 5}, {18, 18, 19, 6, 4, 8, 5}, {17, 16, 17, 6, 4, 8, 5}, {14, 15, 14, 9, 4, 8, 6}},
        {{22, 21, 22, 6, 5, 16, 5}, {18, 18, 19, 5, 4, 12, 6}, {17, 17, 17, 5, 4, 8, 6}, {14, 15, 14, 3, 4, 12, 7}},
        {{22, 22, 23, 6, 5, 32, 5}, {18, 19, 19, 7, 4, 12, 6}, {17, 18, 17, 7, 4, 12, 6}, {14, 15, 14, 4, 3, 24, 7}}};
    zparams p = rows[level - 1][(n <= 256 * 1024) + (n <= 128 * 1024) + (n <= 16 * 1024)];
    const uint32_t src_log = n < 64 ? 6 : (uint32_t)hibit((uint32_t)(n - 1)) + 1;
    if (p.wlog > src_log) p.wlog = src_log;
    if (p.hlog > p.wlog + 1) p.hlog = p.wlog + 1;
    if (p.clog - (p.strat >= 6) > p.wlog) p.clog = p.wlog + (p.strat >= 6);     /* ZSTD_cycleLog: a binary tree has half as many nodes */
    if (p.wlog < 10) p.wlog = 10;
    return p;
}

static int is_rle(const uint8_t* s, size_t n)
{ for (size_t i = 1; i < n; i++) if (s[i] != s[0]) return 0; return 1; }

int64_t orc_zstd_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int level)
{
    zparams p;
    zmatch m;
    optstate opt;
    zentropy ent[2];
    int cur = 0, first = 1;
    size_t o = 0, pos = 0, block;
    int64_t result;
    uint8_t *llc, *ofc, *mlc;
    if (level < 1 || level > 12) return ORC_ZSTD_UNSUPPORTED;
    p = level_params(level, n);
    if (!p.strat) return ORC_ZSTD_UNSUPPORTED;
    if (cap < 18) return ERR_TOOSMALL;
    {   /* frame header: magic, descriptor, [window], content size */
        const uint32_t wsize = 1u << p.wlog;
        const int single = wsize >= n;
        const unsigned fcs = (n >= 256) + (n >= 65536 + 256) + (n >= 0xFFFFFFFFu);
        dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD;
        o = 4;
        dst[o++] = (uint8_t)((single << 5) + (fcs << 6));
        if (!single) dst[o++] = (uint8_t)((p.wlog - 10) << 3);
        if (fcs == 0) { if (single) dst[o++] = (uint8_t)n; }
        else if (fcs == 1) { const uint32_t v = (uint32_t)n - 256; dst[o++] = (uint8_t)v; dst[o++] = (uint8_t)(v >> 8); }
        else if (fcs == 2) { const uint32_t v = (uint32_t)n; memcpy(dst + o, &v, 4); o += 4; }
        else { const uint64_t v = n; memcpy(dst + o, &v, 8); o += 8; }
    }
    if (!n) {
        if (cap - o < 4) return ERR_TOOSMALL;
        dst[o] = 1; dst[o + 1] = 0; dst[o + 2] = 0;
        return (int64_t)o + 3;
    }
    block = (size_t)1 << p.wlog; if (block > n) block = n; if (block > BLOCK_MAX) block = BLOCK_MAX;
    m.p = p;
    m.table = (uint32_t*)calloc((size_t)1 << p.hlog, 4);
    m.small = (uint32_t*)calloc((size_t)1 << p.clog, 4);
    m.tags = (uint8_t*)calloc((size_t)2 << p.hlog, 1);
    memset(&opt, 0, sizeof opt);
    if (p.strat == 7) {
        opt.node = (optnode*)malloc((OPT_NUM + 2) * sizeof(optnode)); opt.match = (optmatch*)malloc((OPT_NUM + 2) * sizeof(optmatch));
        opt.hlog3 = p.mml == 3 ? (p.wlog < 17 ? p.wlog : 17) : 1;       /* ZSTD_reset_matchState: hashLog3 = MIN(ZSTD_HASHLOG3_MAX, windowLog) */
        opt.hash3 = (uint32_t*)calloc((size_t)1 << opt.hlog3, 4);
    }
    m.next_to_update = m.low_limit = m.dict_limit = 2;                /* ZSTD_WINDOW_START_INDEX */
    memset(m.hash_cache, 0, sizeof m.hash_cache);
    m.seq = (zseq*)malloc((BLOCK_MAX / 3 + 1) * sizeof(zseq));
    m.lit = (uint8_t*)malloc(BLOCK_MAX + 64);
    llc = (uint8_t*)malloc(3 * (BLOCK_MAX / 3 + 1)); ofc = llc + BLOCK_MAX / 3 + 1; mlc = ofc + BLOCK_MAX / 3 + 1;
    memset(ent, 0, sizeof ent);
    ent[0].rep[0] = 1; ent[0].rep[1] = 4; ent[0].rep[2] = 8;
    result = 0;
    while (pos < n) {
        const size_t len = n - pos < block ? n - pos : block;
        const int last = len >= n - pos;
        const zentropy* prev = &ent[cur];
        zentropy* next = &ent[cur ^ 1];
        size_t bcap;
        int64_t c = 0;
        if (cap - o < 3 + 2 + 1) { result = ERR_TOOSMALL; break; }
        bcap = cap - o - 3;
        {   /* ZSTD_window_enforceMaxDist(window, block START, ...) and the nextToUpdate floor (zstd_compress.c:4017-4021) */
            const uint32_t start_idx = (uint32_t)pos + 2, max_dist = 1u << p.wlog;
            if (start_idx > max_dist) {
                if (m.low_limit < start_idx - max_dist) m.low_limit = start_idx - max_dist;
                if (m.dict_limit < m.low_limit) m.dict_limit = m.low_limit;
            }
            if (m.next_to_update < m.low_limit) m.next_to_update = m.low_limit;
        }
        if (len >= 7) {
            uint8_t* const out = dst + o + 3;
            size_t tail;
            int64_t lsz, ssz;
            m.nseq = 0; m.nlit = 0;
            memcpy(next->rep, prev->rep, sizeof next->rep);
            {   /* limited catch-up after a very long match (ZSTD_buildSeqStore, zstd_compress.c:2890-2896) */
                const uint32_t curr = (uint32_t)pos + 2;
                if (curr > m.next_to_update + 384) { const uint32_t gap = curr - m.next_to_update - 384; m.next_to_update = curr - (gap < 192 ? gap : 192); }
            }
            if (p.strat == 7) tail = opt_block(&m, &opt, next->rep, src, pos, pos + len);
            else if (p.strat >= 3) tail = lazy_block(&m, next->rep, src, pos, pos + len, p.strat >= 5 ? 2 : p.strat == 3 ? 0 : 1, p.strat == 6 ? 2 : p.wlog > 14);
            else tail = p.strat == 2 ? dfast_block(&m, next->rep, src, pos, pos + len) : fast_block(&m, next->rep, src, pos, pos + len);
            memcpy(m.lit + m.nlit, src + pos + len - tail, tail); m.nlit += tail;
            lsz = compress_literals(prev, next, out, bcap, m.lit, m.nlit, m.nseq == 0 || m.nlit / m.nseq >= 20, p.strat);
            c = lsz;
            if (lsz >= 0) {
                ssz = encode_sequences(out + lsz, bcap - (size_t)lsz, m.seq, m.nseq, llc, ofc, mlc, p.strat, prev, next);
                c = ssz <= 0 ? ssz : lsz + ssz;
            }
            if (c == ERR_TOOSMALL && len <= bcap) c = 0;
            if (c < 0) { result = c; break; }
            if (c > 0 && (size_t)c >= len - ((len >> 6) + 2)) c = 0;
            if (!first && c < 25 && is_rle(src + pos, len)) { c = 1; out[0] = src[pos]; }
            if (c > 1) cur ^= 1;
        }
        if (c == 0) {
            const uint32_t h = (uint32_t)last + ((uint32_t)len << 3);
            if (len + 3 > cap - o) { result = ERR_TOOSMALL; break; }
            dst[o] = (uint8_t)h; dst[o + 1] = (uint8_t)(h >> 8); dst[o + 2] = (uint8_t)(h >> 16);
            memcpy(dst + o + 3, src + pos, len);
            o += 3 + len;
        } else {
            const uint32_t h = c == 1 ? (uint32_t)last + (1u << 1) + ((uint32_t)len << 3) : (uint32_t)last + (2u << 1) + ((uint32_t)c << 3);
            dst[o] = (uint8_t)h; dst[o + 1] = (uint8_t)(h >> 8); dst[o + 2] = (uint8_t)(h >> 16);
            o += 3 + (size_t)c;
        }
        pos += len; first = 0;
    }
    free(m.table); free(m.small); free(m.tags); free(m.seq); free(m.lit); free(llc); free(opt.node); free(opt.match); free(opt.hash3);
    return result < 0 ? result : (int64_t)o;
}

int orc_codec_zstd3(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; const int64_t r = orc_zstd_compress(src, (size_t)n, dst, cap < 0 ? 0 : (size_t)cap, 3); return r < 0 ? 0 : (int)r; }

int orc_codec_zstd1(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; const int64_t r = orc_zstd_compress(src, (size_t)n, dst, cap < 0 ? 0 : (size_t)cap, 1); return r < 0 ? 0 : (int)r; }
