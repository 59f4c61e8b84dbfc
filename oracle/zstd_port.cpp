/*
 * oracle/zstd_port.cpp - scalar CPU restatement of ZSTD frame decoding as 4mz uses it
 * (one frame per <=4 MiB block, no dictionary, no content checksum).
 * TEST INFRASTRUCTURE, NOT PRODUCT (see oracle.h).  Plain C++ compiled with g++; no GPU, no reference code.
 *
 * Follows ZSTD_decompress            native/zstd/decompress/zstd_decompress.c:1112-1127 -> :901-987
 *         block decode               native/zstd/decompress/zstd_decompress_block.c:2003-2074
 *         literals / Huffman         :120-330, common/entropy_common.c:235-340, decompress/huf_decompress.c:339-470
 *         FSE tables                 common/entropy_common.c:43-205, zstd_decompress_block.c:447-565
 *         sequences                  zstd_decompress_block.c:656-735, :1176-1296, :1565-1650
 * Parity: pinned against oracle/_ref (ZSTD_compress at levels 1/3/6/12 -> this decoder == input;
 * mutated frames: same accept/reject as ZSTD_decompress, tests/test_oracle_golden.py) and against
 * tests/golden/zstd_frames.json + the .4mz files in tests/golden/small_files.json.
 * Documented deviations: frames carrying a content checksum or a dictionary id are rejected
 * (4mz never writes them, native/4mc.c:467 uses plain ZSTD_compress).
 */
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

namespace {

static void wave_copy(uint8_t* d, const uint8_t* s, int n, int) { memmove(d, s, (size_t)n); }
static void copy_match(uint8_t* dst, int op, int off, int n, int) { for (int i = 0; i < n; i++) dst[op + i] = dst[op - off + i]; }

#define kBlockMax (128 << 10)
#define kErr (-1)

// ---- code -> (baseline, extra bits) tables (zstd_decompress_internal.h:30-55, zstd_internal.h:121-145)
static const uint32_t kLLBase[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000};
static const uint8_t  kLLBits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint32_t kMLBase[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003};
static const uint8_t  kMLBits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
// predefined distributions (zstd_internal.h:128-166)
static const int16_t kLLDef[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int16_t kMLDef[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static const int16_t kOFDef[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

static inline int hibit(uint32_t v) { return 31 - __builtin_clz(v); }

// per-wavefront entropy state in LDS
struct ZState {
    uint32_t ll[512], ml[512], of[256];     // FSE decode entries: nextState<<16 | nbBits<<8 | symbol
    uint32_t wt[64];                        // FSE table of the Huffman weights (log <= 6)
    uint16_t huf[4096];                     // Huffman X1 entries: nbBits<<8 | symbol
    int16_t  norm[256];
    uint16_t next[256];
    uint8_t  weights[256];
    uint8_t  spread[520];
    uint32_t rank[16];
};

// 8 bytes at p (any alignment), never reading at or beyond p+avail
static inline uint64_t load64_safe(const uint8_t* p, int avail)
{
    if (avail >= 8) { struct __attribute__((packed, aligned(1))) U8 { uint64_t v; }; return reinterpret_cast<const U8*>(p)->v; }
    uint64_t v = 0;
    for (int i = 0; i < avail; i++) v |= uint64_t(p[i]) << (8 * i);
    return v;
}

// backward bitstream (common/bitstream.h:252-300): bits are consumed from the end mark downwards;
// reading below bit 0 yields zeros (the reference's "overflow" state), pos goes negative.
struct BitsBack {
    const uint8_t* p; int len; int pos;
    inline bool init(const uint8_t* ptr, int n) {
        p = ptr; len = n;
        if (n < 1) return false;
        const uint32_t last = ptr[n - 1];
        if (last == 0) return false;
        pos = 8 * (n - 1) + hibit(last);
        return true;
    }
    inline uint32_t peek_at(int start, int n, const bool fast = false) const {       // bits [start, start+n), n <= 32
        if (n == 0) return 0;
        const uint64_t mask = (1ull << n) - 1;
        if (start >= 0) {
            const int byte = start >> 3;
            return uint32_t((load64_safe(p + byte, len - byte) >> (start & 7)) & mask);
        }
        // reading past the stream's start: the reference's reader computes its shift as (64 - bitsConsumed - n) & 63 (bitstream.h:
        // BIT_lookBits -> BIT_getMiddleBits) on a container that holds the stream's first eight bytes by then - it returns wrapped
        // container bits, not zeros, and a frame that still adds up is accepted with them
        // (BIT_readBitsFast - the extra bits of offsets and lengths - shifts the other way round, bitstream.h:344-349: what is left of
        // the stream followed by zeros, or, with nothing left, bits from the container's top)
        if (fast) return uint32_t((load64_safe(p, len) << ((64 - (start + n)) & 63)) >> ((64 - n) & 63));
        return uint32_t((load64_safe(p, len) >> (start & 63)) & mask);
    }
    inline uint32_t read(int n) { pos -= n; return peek_at(pos, n); }
    inline uint32_t read(int n, bool fast) { pos -= n; return peek_at(pos, n, fast); }
};

// forward bit reader for FSE table descriptions
struct BitsFwd {
    const uint8_t* p; int len; int bit;
    inline uint32_t peek(int n) const {
        const int byte = bit >> 3;
        if (byte >= len) return 0;
        return uint32_t((load64_safe(p + byte, len - byte) >> (bit & 7)) & ((1ull << n) - 1));
    }
};

// FSE_readNCount (common/entropy_common.c:43-205).  Returns bytes consumed or -1.
static int read_ncount(const uint8_t* p, int len, int16_t* norm, int* max_sym, int* table_log, int max_log)
{
    if (len < 1) return kErr;
    BitsFwd br{p, len, 0};
    const int al = int(br.peek(4)) + 5; br.bit += 4;
    if (al > max_log || al > 15) return kErr;
    *table_log = al;
    int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0;
    const int maxsv = *max_sym;
    bool prev0 = false;
    while (remaining > 1 && sym <= maxsv) {
        if (prev0) {
            int n0 = sym;
            for (;;) {
                const uint32_t r = br.peek(2); br.bit += 2;
                n0 += int(r);
                if (r != 3) break;
                if ((br.bit >> 3) > len) return kErr;
            }
            if (n0 > maxsv + 1) return kErr;
            while (sym < n0) norm[sym++] = 0;
            if (sym > maxsv) break;          // the reference re-tests the loop condition here
        }
        {
            const int maxv = (2 * threshold - 1) - remaining;
            int count;
            const uint32_t bits = br.peek(nbits);
            if (int(bits & uint32_t(threshold - 1)) < maxv) { count = int(bits & uint32_t(threshold - 1)); br.bit += nbits - 1; }
            else { count = int(bits & uint32_t(2 * threshold - 1)); if (count >= threshold) count -= maxv; br.bit += nbits; }
            count--;                                      // -1 encodes "less than one" probability
            remaining -= count < 0 ? -count : count;
            norm[sym++] = int16_t(count);
            prev0 = (count == 0);
            if (remaining < 1) return kErr;
            while (remaining < threshold && threshold > 1) { nbits--; threshold >>= 1; }
        }
        if ((br.bit >> 3) > len) return kErr;
    }
    if (remaining != 1) return kErr;
    if (sym > maxsv + 1) return kErr;
    for (int i = sym; i <= maxsv; i++) norm[i] = 0;
    *max_sym = sym - 1;
    const int used = (br.bit + 7) >> 3;
    if (used > len) return kErr;
    return used;
}

// FSE decode table (zstd_decompress_block.c:447-565 / fse_decompress.c:71-150): same spreading rule.
static void build_fse(uint32_t* tab, const int16_t* norm, int max_sym, int table_log, uint16_t* next)
{
    const int size = 1 << table_log, mask = size - 1;
    int high = size - 1;
    for (int s = 0; s <= max_sym; s++) {
        if (norm[s] == -1) { tab[high--] = uint32_t(s); next[s] = 1; }
        else next[s] = uint16_t(norm[s]);
    }
    const int step = (size >> 1) + (size >> 3) + 3;
    int pos = 0;
    for (int s = 0; s <= max_sym; s++)
        for (int i = 0; i < norm[s]; i++) {
            tab[pos] = uint32_t(s);
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    for (int u = 0; u < size; u++) {
        const uint32_t s = tab[u];
        const uint32_t ns = next[s]++;
        const uint32_t nb = uint32_t(table_log - hibit(ns));
        tab[u] = (((ns << nb) - uint32_t(size)) << 16) | (nb << 8) | s;
    }
}

// Huffman tree description -> X1 decode table (entropy_common.c:235-340, huf_decompress.c:339-470).
// Returns bytes consumed or -1; *log_out = table log.
static int read_huf_table(ZState* z, const uint8_t* p, int len, int* log_out)
{
    if (len < 1) return kErr;
    int isz = p[0], nsym;
    int consumed;
    if (isz >= 128) {                                   // direct 4-bit weights
        nsym = isz - 127;
        const int bytes = (nsym + 1) / 2;
        if (bytes + 1 > len || nsym >= 256) return kErr;
        for (int n = 0; n < nsym; n += 2) { z->weights[n] = p[1 + n / 2] >> 4; z->weights[n + 1] = p[1 + n / 2] & 15; }
        consumed = bytes + 1;
    } else {                                            // weights compressed with FSE (<= 6 bit table)
        if (isz + 1 > len) return kErr;
        int maxs = 255, tl;
        const int h = read_ncount(p + 1, isz, z->norm, &maxs, &tl, 6);
        if (h < 0) return kErr;
        build_fse(z->wt, z->norm, maxs, tl, z->next);    // own table: LL/OF/ML must survive for 'repeat' mode
        BitsBack bs;
        if (!bs.init(p + 1 + h, isz - h)) return kErr;
        uint32_t s1 = bs.read(tl), s2 = bs.read(tl);
        nsym = 0;
        for (;;) {                                      // two interleaved states (fse_decompress.c:268-287)
            if (nsym > 253) return kErr;
            uint32_t e = z->wt[s1];
            z->weights[nsym++] = uint8_t(e);
            s1 = (e >> 16) + bs.read(int((e >> 8) & 0xff));
            if (bs.pos < 0) { z->weights[nsym++] = uint8_t(z->wt[s2]); break; }
            if (nsym > 253) return kErr;
            e = z->wt[s2];
            z->weights[nsym++] = uint8_t(e);
            s2 = (e >> 16) + bs.read(int((e >> 8) & 0xff));
            if (bs.pos < 0) { z->weights[nsym++] = uint8_t(z->wt[s1]); break; }
        }
        consumed = isz + 1;
    }
    // weight statistics, implied last weight
    for (int i = 0; i < 16; i++) z->rank[i] = 0;
    uint32_t total = 0;
    for (int n = 0; n < nsym; n++) {
        const uint32_t w = z->weights[n];
        if (w > 12) return kErr;
        z->rank[w]++;
        total += (1u << w) >> 1;
    }
    if (total == 0) return kErr;
    const int tlog = hibit(total) + 1;
    if (tlog > 12) return kErr;
    {
        const uint32_t rest = (1u << tlog) - total;
        if (rest == 0 || (rest & (rest - 1))) return kErr;        // must be a clean power of two
        const uint32_t lastw = uint32_t(hibit(rest)) + 1;
        z->weights[nsym] = uint8_t(lastw);
        z->rank[lastw]++;
        nsym++;
    }
    if (z->rank[1] < 2 || (z->rank[1] & 1)) return kErr;
    // table: symbols ordered by weight, then by value; a symbol of weight w owns 2^(w-1) cells
    uint32_t start = 0;
    for (int w = 1; w <= tlog; w++) { const uint32_t c = z->rank[w]; z->rank[w] = start; start += c << (w - 1); }
    for (int sy = 0; sy < nsym; sy++) {
        const uint32_t w = z->weights[sy];
        if (!w) continue;
        const uint32_t n = 1u << (w - 1), at = z->rank[w];
        const uint16_t e = uint16_t(((tlog + 1 - w) << 8) | uint32_t(sy));
        for (uint32_t i = 0; i < n; i++) z->huf[at + i] = e;
        z->rank[w] = at + n;
    }
    *log_out = tlog;
    return consumed;
}

// literal source of a block: a pointer plus a mode (raw bytes in place, one repeated byte, scratch)
struct Lits { const uint8_t* p; uint32_t size; uint32_t pos; uint8_t rle; bool is_rle; };

static inline void copy_lits(uint8_t* dst, const Lits& l, uint32_t n, int)
{
    if (l.is_rle) memset(dst, l.rle, n);
    else memmove(dst, l.p + l.pos, n);
}

// Decodes `csize` bytes of zstd frames into dst[0..cap).  Returns bytes or < 0.

// Which Huffman decoder the reference takes for a 4-stream compressed literals section (huf_decompress.c:1595-1617, timings :1568-1587):
// 1 = the double-symbol decoder (X2).  On valid streams both decoders give the same bytes; they differ in what they make of the LAST
// symbol of a stream (see huf_stream_end_ok).
static inline bool huf_select_x2(uint32_t dst_size, uint32_t csrc_size)
{
    const uint32_t t0[16] = {0, 0, 150, 170, 177, 197, 221, 256, 359, 582, 688, 825, 976, 1180, 1377, 1412};
    const uint32_t d0[16] = {0, 0, 216, 205, 199, 194, 192, 189, 188, 187, 187, 186, 185, 186, 185, 185};
    const uint32_t t1[16] = {1, 1, 381, 514, 539, 644, 735, 881, 1167, 1570, 1712, 1965, 2131, 2070, 1731, 1695};
    const uint32_t d1[16] = {1, 1, 119, 112, 110, 107, 107, 106, 109, 114, 122, 136, 150, 175, 202, 202};
    const uint32_t Q = csrc_size >= dst_size ? 15u : uint32_t(uint64_t(csrc_size) * 16 / dst_size);
    const uint32_t D256 = dst_size >> 8;
    const uint32_t time0 = t0[Q] + d0[Q] * D256;
    uint32_t time1 = t1[Q] + d1[Q] * D256;
    time1 += time1 >> 5;
    return time1 < time0;
}

static int zstd_decode_frames(const uint8_t* src, int csize, uint8_t* dst, int cap, uint8_t* litbuf,
                                  ZState* z, int lane)
{
    int ip = 0, op = 0;
    while (csize - ip >= 5) {                                         // ZSTD_startingInputLength
        if (csize - ip < 4) return kErr;
        const uint32_t magic = uint32_t(src[ip]) | (uint32_t(src[ip + 1]) << 8) | (uint32_t(src[ip + 2]) << 16) | (uint32_t(src[ip + 3]) << 24);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                   // skippable frame
            if (csize - ip < 8) return kErr;
            const uint32_t sz = uint32_t(src[ip + 4]) | (uint32_t(src[ip + 5]) << 8) | (uint32_t(src[ip + 6]) << 16) | (uint32_t(src[ip + 7]) << 24);
            if (sz > uint32_t(csize - ip - 8)) return kErr;
            ip += 8 + int(sz);
            continue;
        }
        if (magic != 0xFD2FB528u) return kErr;
        // ---- frame header (zstd_decompress.c:443-545)
        if (csize - ip < 6) return kErr;
        const uint32_t fhd = src[ip + 4];
        const int fcs_id = fhd >> 6, single = (fhd >> 5) & 1, has_sum = (fhd >> 2) & 1, did = fhd & 3;
        if (fhd & 0x08) return kErr;                                  // reserved bit
        const int did_sz = did == 3 ? 4 : did;                         // 0,1,2,4 bytes
        const int fcs_sz = fcs_id == 0 ? single : (1 << fcs_id);       // (0|1),2,4,8 bytes
        const int hsize = 5 + (single ? 0 : 1) + did_sz + fcs_sz;
        if (csize - ip < hsize) return kErr;
        int hp = ip + 5;
        uint64_t window = 0;
        if (!single) {
            const uint32_t wb = src[hp++];
            const int wlog = int(wb >> 3) + 10;
            if (wlog > 31) return kErr;
            window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wb & 7);
        }
        uint32_t dict = 0;
        for (int i = 0; i < did_sz; i++) dict |= uint32_t(src[hp++]) << (8 * i);
        if (dict != 0) return kErr;                                   // no dictionaries on this path
        uint64_t fcs = ~0ull;
        if (fcs_id == 0) { if (single) fcs = src[hp++]; }
        else {
            fcs = 0;
            for (int i = 0; i < fcs_sz; i++) fcs |= uint64_t(src[hp++]) << (8 * i);
            if (fcs_id == 1) fcs += 256;
        }
        if (single) window = fcs;
        if (has_sum) return kErr;          // content checksum (XXH64) never occurs in 4mz frames; not implemented
        if (fcs != ~0ull && fcs > uint64_t(cap - op)) return kErr;    // dstSize_tooSmall
        (void)window;
        ip += hsize;
        const int frame_start = op;

        // ---- per-frame entropy state (zstd_decompress.c: ZSTD_decompressBegin)
        uint32_t rep0 = 1, rep1 = 4, rep2 = 8;                        // zstd_internal.h:70
        bool have_huf = false, have_fse = false;
        bool huf_x2 = false;                                           // the current Huffman table is the reference's double-symbol kind
        int huf_log = 0, ll_log = 0, of_log = 0, ml_log = 0;

        for (;;) {
            if (csize - ip < 3) return kErr;
            const uint32_t bh = uint32_t(src[ip]) | (uint32_t(src[ip + 1]) << 8) | (uint32_t(src[ip + 2]) << 16);
            ip += 3;
            const int last = bh & 1, btype = (bh >> 1) & 3;
            const uint32_t bsize = bh >> 3;
            if (btype == 3) return kErr;
            if (btype == 0) {                                         // raw
                if (bsize > uint32_t(csize - ip) || bsize > uint32_t(cap - op)) return kErr;
                wave_copy(dst + op, src + ip, int(bsize), lane);
                ip += int(bsize); op += int(bsize);
            } else if (btype == 1) {                                  // RLE
                if (csize - ip < 1 || bsize > uint32_t(cap - op)) return kErr;
                const uint8_t v = src[ip]; ip += 1;
                memset(dst + op, v, bsize);
                op += int(bsize);
            } else {
                // ---------------------------------------------------------------- compressed block
                if (bsize >= uint32_t(kBlockMax)) return kErr;         // zstd_decompress_block.c:2021
                if (bsize > uint32_t(csize - ip) || bsize < 2) return kErr;
                const uint8_t* bp = src + ip;
                const int bend = int(bsize);
                int bpos = 0;
                Lits lits; lits.pos = 0; lits.is_rle = false; lits.rle = 0; lits.p = litbuf; lits.size = 0;
                {   // ---- literals section (zstd_decompress_block.c:120-330)
                    const uint32_t b0 = bp[0];
                    const int ltype = b0 & 3, fmt = (b0 >> 2) & 3;
                    if (ltype >= 2) {
                        if (bend < 5) return kErr;
                        if (ltype == 3 && !have_huf) return kErr;
                        const uint32_t lhc = uint32_t(bp[0]) | (uint32_t(bp[1]) << 8) | (uint32_t(bp[2]) << 16) | (uint32_t(bp[3]) << 24);
                        uint32_t lh, lsize, lcsize; bool one_stream = false;
                        if (fmt <= 1) { one_stream = (fmt == 0); lh = 3; lsize = (lhc >> 4) & 0x3FF; lcsize = (lhc >> 14) & 0x3FF; }
                        else if (fmt == 2) { lh = 4; lsize = (lhc >> 4) & 0x3FFF; lcsize = lhc >> 18; }
                        else { lh = 5; lsize = (lhc >> 4) & 0x3FFFF; lcsize = (lhc >> 22) + (uint32_t(bp[4]) << 10); }
                        if (lsize > uint32_t(kBlockMax) || lcsize + lh > uint32_t(bend)) return kErr;
                        const uint8_t* hp8 = bp + lh;
                        int hlen = int(lcsize);
                        if (ltype == 2) {
                            int used = 0;
                            if (lane == 0) used = read_huf_table(z, hp8, hlen, &huf_log);
                            used = used;
                            huf_log = huf_log;
                            if (used < 0) return kErr;
                            hp8 += used; hlen -= used;
                            have_huf = true;
                            huf_x2 = !one_stream && huf_select_x2(lsize, lcsize);     // zstd_decompress_block.c:183-205: one stream -> X1, four -> by size
                        }
                        // stream layout: one stream, or 6-byte jump table + four streams (huf_decompress.c:561-590)
                        bool ok = true;
                        for (int lane_ = 0; lane_ < (one_stream ? 1 : 4); lane_++) {
                        int s_off = 0, s_len = hlen, o_off = 0, o_len = int(lsize);
                        if (!one_stream) {
                            if (hlen < 10 || lsize < 6) return kErr;
                            const int l1 = hp8[0] | (hp8[1] << 8), l2 = hp8[2] | (hp8[3] << 8), l3 = hp8[4] | (hp8[5] << 8);
                            const int l4 = hlen - 6 - l1 - l2 - l3;
                            if (l4 < 1) return kErr;
                            const int seg = (int(lsize) + 3) / 4;
                            if (3 * seg > int(lsize)) return kErr;
                            const int j = lane_ & 3;
                            s_off = 6 + (j > 0 ? l1 : 0) + (j > 1 ? l2 : 0) + (j > 2 ? l3 : 0);
                            s_len = j == 0 ? l1 : (j == 1 ? l2 : (j == 2 ? l3 : l4));
                            o_off = seg * j; o_len = (j == 3) ? int(lsize) - 3 * seg : seg;
                        }
                        {
                            BitsBack bs;
                            if (!bs.init(hp8 + s_off, s_len)) ok = false;
                            else {
                                // The double-symbol decoder (huf_decompress.c:1141-1221) takes one or two symbols per look-up: two when both
                                // codes fit its table log T.  `open` follows that pairing (length of a look-up's first symbol while its
                                // second is undecided), because its end rule depends on it: a stream whose last byte is the FIRST symbol of a
                                // look-up ends in HUF_decodeLastSymbolX2, which forgives a two-symbol entry that runs past the stream's
                                // start (:1150-1163).  Every other case needs the stream consumed exactly, as the single-symbol decoder does.
                                const int T = huf_log <= 11 ? 11 : 12;                    // :1083
                                int open = -1, last_r = 0, last_len = 0; bool last_alone = false;
                                for (int i = 0; i < o_len; i++) {
                                    const int before = bs.pos;
                                    const uint32_t idx = (bs.pos >= huf_log) ? bs.peek_at(bs.pos - huf_log, huf_log)
                                                                             : (bs.peek_at(0, bs.pos > 0 ? bs.pos : 0) << (huf_log - (bs.pos > 0 ? bs.pos : 0)));
                                    const uint32_t e = z->huf[idx];
                                    const int len = int(e >> 8);
                                    bs.pos -= len;
                                    litbuf[o_off + i] = uint8_t(e);
                                    if (open >= 0 && open + len <= T) { open = -1; continue; }     // second symbol of a look-up
                                    open = len;                                                     // first symbol of a look-up
                                    if (i == o_len - 1) { last_alone = true; last_r = before; last_len = len; }
                                }
                                if (huf_x2 && last_alone && o_len > 0) {
                                    const uint32_t idx2 = (bs.pos >= huf_log) ? bs.peek_at(bs.pos - huf_log, huf_log)
                                                                              : (bs.peek_at(0, bs.pos > 0 ? bs.pos : 0) << (huf_log - (bs.pos > 0 ? bs.pos : 0)));
                                    const int len2 = int(z->huf[idx2] >> 8);
                                    const bool pair = last_len + len2 <= T;
                                    if (last_r < 0) ok = false;                                     // over-consumed before the last symbol
                                    else if (last_r == 0) {
                                        // nothing left: BIT_lookBitsFast shifts by (bitsConsumed & 63) = 0 and looks at the TOP of its
                                        // container - the stream's first eight bytes - again (bitstream.h:332-337); a two-symbol entry
                                        // there is accepted without consuming anything, and its first symbol is the stream's last byte
                                        const uint32_t v = bs.peek_at(64 - T, T);
                                        const uint32_t ea = z->huf[v >> (T - huf_log)];
                                        const int la = int(ea >> 8);
                                        const int lb = int(z->huf[((v << la) & ((1u << T) - 1)) >> (T - huf_log)] >> 8);
                                        if (la + lb <= T) litbuf[o_off + o_len - 1] = uint8_t(ea); else ok = false;
                                    }
                                    else if (pair) { if (last_len + len2 < last_r) ok = false; }
                                    else if (last_len != last_r) ok = false;
                                } else if (bs.pos != 0) ok = false;       // every stream must end exactly at its start
                            }
                        }
                        }
                        if (!ok) return kErr;
                        lits.p = litbuf; lits.size = lsize;
                        bpos = int(lh + lcsize);
                    } else {
                        uint32_t lh, lsize;
                        if ((fmt & 1) == 0) { lh = 1; lsize = b0 >> 3; }
                        else if (fmt == 1) { if (bend < 2) return kErr; lh = 2; lsize = (b0 | (uint32_t(bp[1]) << 8)) >> 4; }
                        else { if (bend < 3) return kErr; lh = 3; lsize = (b0 | (uint32_t(bp[1]) << 8) | (uint32_t(bp[2]) << 16)) >> 4; }
                        if (lsize > uint32_t(kBlockMax)) return kErr;
                        if (ltype == 0) {                              // raw: used in place
                            if (lh + lsize > uint32_t(bend)) return kErr;
                            lits.p = bp + lh; lits.size = lsize; bpos = int(lh + lsize);
                        } else {                                       // RLE
                            if (lh + 1 > uint32_t(bend)) return kErr;
                            lits.is_rle = true; lits.rle = bp[lh]; lits.size = lsize; bpos = int(lh) + 1;
                        }
                    }
                }
                // ---- sequences header (zstd_decompress_block.c:656-735)
                if (bend - bpos < 1) return kErr;
                int nseq = bp[bpos++];
                if (nseq == 0) { if (bpos != bend) return kErr; }
                else {
                    if (nseq > 0x7F) {
                        if (nseq == 0xFF) { if (bpos + 2 > bend) return kErr; nseq = (bp[bpos] | (bp[bpos + 1] << 8)) + 0x7F00; bpos += 2; }
                        else { if (bpos >= bend) return kErr; nseq = ((nseq - 0x80) << 8) + bp[bpos++]; }
                    }
                    if (bpos + 1 > bend) return kErr;
                    const uint32_t modes = bp[bpos++];
                    // (the two reserved bits of the modes byte are not looked at: zstd_decompress_block.c:689-692)
                    for (int t = 0; t < 3; t++) {                      // LL, OF, ML in this order
                        const int mode_t = int((modes >> (6 - 2 * t)) & 3);
                        uint32_t* tab = t == 0 ? z->ll : (t == 1 ? z->of : z->ml);
                        int* logp = t == 0 ? &ll_log : (t == 1 ? &of_log : &ml_log);
                        const int maxsym = t == 0 ? 35 : (t == 1 ? 31 : 52), maxlog = t == 0 ? 9 : (t == 1 ? 8 : 9);
                        if (mode_t == 1) {                            // RLE: one symbol, zero-bit states
                            if (bpos >= bend) return kErr;
                            const uint32_t sy = bp[bpos++];
                            if (int(sy) > maxsym) return kErr;
                            if (lane == 0) tab[0] = sy;
                            *logp = 0;
                        } else if (mode_t == 0) {                     // predefined distribution
                            const int dlog = t == 1 ? 5 : 6, dmax = t == 0 ? 35 : (t == 1 ? 28 : 52);
                            if (lane == 0) {
                                const int16_t* d = t == 0 ? kLLDef : (t == 1 ? kOFDef : kMLDef);
                                for (int i = 0; i <= dmax; i++) z->norm[i] = d[i];
                                build_fse(tab, z->norm, dmax, dlog, z->next);
                            }
                            *logp = dlog;
                        } else if (mode_t == 2) {                     // described in the stream
                            int used = 0, tl = 0;
                            if (lane == 0) {
                                int ms = maxsym;
                                used = read_ncount(bp + bpos, bend - bpos, z->norm, &ms, &tl, maxlog);
                                if (used >= 0) build_fse(tab, z->norm, ms, tl, z->next);
                            }
                            used = used; tl = tl;
                            if (used < 0) return kErr;
                            bpos += used; *logp = tl;
                        } else {                                       // repeat the previous block's table
                            if (!have_fse) return kErr;
                        }
                    }
                    have_fse = true;
                    // ---- sequence bitstream (zstd_decompress_block.c:1565-1650)
                    BitsBack bs;
                    if (!bs.init(bp + bpos, bend - bpos)) return kErr;
                    uint32_t sl = bs.read(ll_log), so = bs.read(of_log), sm = bs.read(ml_log);
                    for (int n = 0; n < nseq; n++) {
                        const uint32_t el = z->ll[sl], eo = z->of[so], em = z->ml[sm];
                        const uint32_t lcode = el & 0xff, ocode = eo & 0xff, mcode = em & 0xff;
                        if (ocode > 31) return kErr;
                        uint32_t offset;
                        const uint32_t ll_base = kLLBase[lcode];
                        if (ocode > 1) {
                            offset = (1u << ocode) - 3 + bs.read(int(ocode), true);  // OF_base[code] = 2^code - 3
                            rep2 = rep1; rep1 = rep0; rep0 = offset;
                        } else {
                            const uint32_t ll0 = (ll_base == 0);
                            if (ocode == 0) {
                                offset = ll0 ? rep1 : rep0;
                                rep1 = ll0 ? rep0 : rep1; rep0 = offset;
                                if (ll0) { /* swapped */ }
                            } else {
                                const uint32_t v = 1 + ll0 + bs.read(1, true);    // OF_base[1] = 1
                                uint32_t t = (v == 3) ? rep0 - 1 : (v == 1 ? rep1 : rep2);
                                t += !t;
                                if (v != 1) rep2 = rep1;
                                rep1 = rep0; rep0 = offset = t;
                            }
                        }
                        const uint32_t mlen = kMLBase[mcode] + bs.read(int(kMLBits[mcode]), true);
                        const uint32_t llen = ll_base + bs.read(int(kLLBits[lcode]), true);
                        sl = (el >> 16) + bs.read(int((el >> 8) & 0xff));
                        sm = (em >> 16) + bs.read(int((em >> 8) & 0xff));
                        so = (eo >> 16) + bs.read(int((eo >> 8) & 0xff));
                        // ---- execute (zstd_decompress_block.c:956-1050)
                        if (llen > lits.size - lits.pos) return kErr;
                        if (llen + mlen > uint32_t(cap - op)) return kErr;
                        copy_lits(dst + op, lits, llen, lane);
                        lits.pos += llen; op += int(llen);
                        if (offset > uint32_t(op)) return kErr;        // before the start of the output (no dictionary)
                        copy_match(dst, op, int(offset), int(mlen), lane);
                        op += int(mlen);
                    }
                    if (bs.pos > 0) return kErr;                       // bits left over: corruption
                    (void)frame_start;
                }
                // ---- trailing literals
                {
                    const uint32_t rest = lits.size - lits.pos;
                    if (rest > uint32_t(cap - op)) return kErr;
                    copy_lits(dst + op, lits, rest, lane);
                    op += int(rest);
                }
                ip += int(bsize);
            }
            if (last) break;
        }
        if (fcs != ~0ull && uint64_t(op - frame_start) != fcs) return kErr;
    }
    if (ip != csize) return kErr;
    return op;
}


} // namespace

extern "C" int64_t orc_zstd_decompress(const uint8_t* src, size_t csize, uint8_t* dst, size_t cap)
{
    if (csize > 0x7FFFFFFFu || cap > 0x7FFFFFFFu) return -1;
    ZState* z = (ZState*)malloc(sizeof(ZState));
    uint8_t* lit = (uint8_t*)malloc(kBlockMax + 64);
    int r = zstd_decode_frames(src, (int)csize, dst, (int)cap, lit, z, 0);
    free(z); free(lit);
    return r;
}

extern "C" int orc_codec_zstd_decode(void* ctx, const uint8_t* src, int n, uint8_t* dst, int cap)
{ (void)ctx; return (int)orc_zstd_decompress(src, (size_t)n, dst, (size_t)cap); }
