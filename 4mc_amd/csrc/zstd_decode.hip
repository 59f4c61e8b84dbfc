// 4mc_amd/csrc/zstd_decode.hip — K9: batched ZSTD frame decode on gfx950 (one 4mz block = one frame).
//
// Replaces the per-block call ZSTD_decompress(out, usize, in, csize) of the reference
// (native/4mc.c:810, native/jniZstdDecompressor.c:90 -> native/zstd/decompress/zstd_decompress.c:1112-1127
//  -> ZSTD_decompressFrame :901-987 -> ZSTD_decompressBlock_internal zstd_decompress_block.c:2003-2074).
// Written from the format (RFC 8878) with the reference's acceptance rules:
//   frame header        zstd_decompress.c:443-545       block header   zstd_decompress_block.c:57-80
//   literals section    zstd_decompress_block.c:120-330 (HUF tree: common/entropy_common.c:235-340,
//                       weights by FSE: common/fse_decompress.c:232-290, X1 table huf_decompress.c:339-470)
//   sequences header    zstd_decompress_block.c:656-735 (FSE_readNCount common/entropy_common.c:43-205,
//                       decode table :447-565)          sequence decode :1176-1296, execution :956-1050
//
// One wavefront owns one frame.  The inner blocks of a frame are a serial chain (repeat offsets,
// "repeat" entropy tables and the window carry over), and inside a block the FSE state walk is
// bit-serial, so this first version runs the entropy control flow wave-uniformly and uses the 64
// lanes where the format allows it: the four Huffman literal streams decode on four lanes in
// parallel, literal runs and matches are copied 64 bytes (or 1 KiB) per step.  Entropy tables live
// in LDS (FSE LL/OF/ML 512+256+512 entries, Huffman <= 4096 entries); regenerated literals go to a
// per-wavefront 128 KiB scratch slot in HBM.  Algorithmic bytes: csize read + usize written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devcopy.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int kBlockMax = 128 << 10;                  // ZSTD_BLOCKSIZE_MAX (zstd.h:132-133)
constexpr int kErr = -1;
constexpr int kOwnBytes = 1024;                       // cap on the bytes one sequence batch may produce

// wave64 inclusive scans on the DPP network (see lz4_decode.hip)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{ return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROWMASK, 0xf, false)); }
__device__ __forceinline__ uint32_t scan_add(uint32_t v)
{
    v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t scan_max(uint32_t v)
{
    v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
    v = max(v, dpp0<0x142, 0xa>(v)); v = max(v, dpp0<0x143, 0xc>(v));
    return v;
}

// ---- code -> (baseline, extra bits) tables (zstd_decompress_internal.h:30-55, zstd_internal.h:121-145)
__constant__ uint32_t kLLBase[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000};
__constant__ uint8_t  kLLBits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__constant__ uint32_t kMLBase[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003};
__constant__ uint8_t  kMLBits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
// predefined distributions (zstd_internal.h:128-166)
__constant__ int16_t kLLDef[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__constant__ int16_t kMLDef[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
__constant__ int16_t kOFDef[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

__device__ __forceinline__ int hibit(uint32_t v) { return 31 - __builtin_clz(v); }

// per-wavefront entropy state in LDS
struct ZState {
    uint32_t ll[512], ml[512], of[256];     // FSE decode entries: nextState<<16 | nbBits<<8 | symbol
    uint32_t wt[64];                        // FSE table of the Huffman weights (log <= 6)
    uint16_t huf[4096];                     // Huffman X1 entries: nbBits<<8 | symbol
    union {                                 // FSE build scratch / LDS ring of the sequence bitstream
        struct { int16_t norm[256]; uint16_t next[256]; };
        uint32_t ring[264];
    };
    uint32_t rank[16];
    uint32_t llv[512], mlv[512];            // per FSE state: baseline | extra bits << 20 of its symbol
    uint32_t llb[36], mlb[56];              // code -> baseline / extra bits (copies of the constant tables)
    uint8_t  llx[36], mlx[56];
    union {                                 // table-build scratch and the copier's owner map are never live together
        uint8_t own[1024];                  // owner map of the batch copier
        struct { uint8_t weights[256]; uint8_t spread[520]; };
    };
};

// 8 bytes at p (any alignment), never reading at or beyond p+avail
__device__ __forceinline__ uint64_t load64_safe(const uint8_t* p, int avail)
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
    int wlo; uint64_t w;                    // cached window: w = stream bits [wlo, wlo + 64); wlo = INT_MAX: empty
    __device__ __forceinline__ bool init(const uint8_t* ptr, int n) {
        p = ptr; len = n; wlo = 0x7FFFFFFF; w = 0;
        if (n < 1) return false;
        const uint32_t last = ptr[n - 1];
        if (last == 0) return false;
        pos = 8 * (n - 1) + hibit(last);
        return true;
    }
    // bits [start, start+n), n <= 31.  Reads walk downwards, so one 8-byte load placed just above the
    // request serves the next ~57 bits; the common case is two compares, a shift and a mask.
    __device__ __forceinline__ uint32_t peek_at(int start, int n, const bool fast = false) {
        const uint32_t mask = (1u << n) - 1;
        if (start < wlo || start + n > wlo + 64) {
            if (start < 0) {
                // reading past the stream's start: the reference's reader computes its shift as (64 - bitsConsumed - n) & 63
                // (bitstream.h: BIT_lookBits -> BIT_getMiddleBits) on a container that holds the stream's first eight bytes by
                // then - it returns wrapped container bits, not zeros, and a frame that still adds up is accepted with them
                // (BIT_readBitsFast - the extra bits of offsets and lengths - shifts the other way round, bitstream.h:344-349: what is
                // left of the stream followed by zeros, or, with nothing left, bits from the container's top)
                if (wlo != 0) { wlo = 0; w = load64_safe(p, len); }
                if (fast) return n ? uint32_t((w << ((64 - (start + n)) & 63)) >> ((64 - n) & 63)) : 0u;
                return uint32_t(w >> (start & 63)) & mask;
            }
            const int byte = max(0, ((start + n + 7) >> 3) - 8);
            wlo = 8 * byte; w = load64_safe(p + byte, len - byte);
        }
        return uint32_t(w >> (start - wlo)) & mask;
    }
    __device__ __forceinline__ uint32_t read(int n) { pos -= n; return peek_at(pos, n); }
    __device__ __forceinline__ uint32_t read_fast(int n) { pos -= n; return peek_at(pos, n, true); }   // BIT_readBitsFast: extra bits of a sequence
};

// The sequence bitstream is read by the whole wave (uniform position), so it is staged 1 KiB at a time in an
// LDS ring: a window refill is an LDS read (~100 clk) instead of a dependent HBM/L2 load per ~2 sequences.
struct SeqBits {
    const uint8_t* p; int len; int pos; int wlo; uint64_t w; int rlo; uint32_t* ring; int lane;
    __device__ __forceinline__ bool init(const uint8_t* ptr, int n, uint32_t* ring_, int lane_) {
        p = ptr; len = n; wlo = 1 << 30; w = 0; rlo = 1 << 30; ring = ring_; lane = lane_;
        if (n < 1) return false;
        const uint32_t last = ptr[n - 1];
        if (last == 0) return false;
        pos = 8 * (n - 1) + hibit(last);
        return true;
    }
    __device__ __forceinline__ void stage(int byte) {               // ring <- stream bytes [rlo, rlo + 1040), zeros beyond len
        rlo = max(0, byte - 1016) & ~3;
        const int at = rlo + 16 * lane;
        uint8_t* const r8 = reinterpret_cast<uint8_t*>(ring);
        if (at + 16 <= len) *reinterpret_cast<uint4*>(ring + 4 * lane) = ld16u(p + at);
        else {
#pragma unroll 1
            for (int k = 0; k < 16; k++) r8[16 * lane + k] = (at + k < len) ? p[at + k] : uint8_t(0);
        }
        if (lane < 16) { const int a2 = rlo + 1024 + lane; r8[1024 + lane] = (a2 < len) ? p[a2] : uint8_t(0); }
    }
    // w <- stream bits [8*byte, 8*byte + 64)
    __device__ __forceinline__ void window(int byte) {
        if (byte < rlo || byte + 8 > rlo + 1040) stage(byte);
        const int q = byte - rlo;
        const uint32_t d0 = ring[q >> 2], d1 = ring[(q >> 2) + 1], d2 = ring[(q >> 2) + 2];
        const uint32_t sh = uint32_t(q & 3);
        const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
        wlo = 8 * byte; w = (uint64_t(hi) << 32) | lo;
    }
    // after refill() the next 57 bits below pos are in the window (fewer only at the very start of the stream)
    __device__ __forceinline__ void refill() { if (pos < wlo + 57 || pos > wlo + 64) window(max(0, ((pos + 7) >> 3) - 8)); }
    // n <= 31 bits below pos; needs a refill() at most 57 bits ago.  Below bit 0 the stream reads as zeros.
    __device__ __forceinline__ uint32_t read(int n, const bool fast = false) {
        const uint32_t mask = (1u << n) - 1;
        pos -= n;
        if (pos < wlo) return fast ? (n ? uint32_t((w << ((64 - (pos + n)) & 63)) >> ((64 - n) & 63)) : 0u)
                                   : uint32_t(w >> (pos & 63)) & mask;      // only when wlo == 0: past the stream's start (see BitsBack)
        return uint32_t(w >> (pos - wlo)) & mask;
    }
};

// forward bit reader for FSE table descriptions
struct BitsFwd {
    const uint8_t* p; int len; int bit;
    __device__ __forceinline__ uint32_t peek(int n) const {
        const int byte = bit >> 3;
        if (byte >= len) return 0;
        return uint32_t((load64_safe(p + byte, len - byte) >> (bit & 7)) & ((1ull << n) - 1));
    }
};

// FSE_readNCount (common/entropy_common.c:43-205).  Returns bytes consumed or -1.
__device__ __forceinline__ int read_ncount(const uint8_t* p, int len, int16_t* norm, int* max_sym, int* table_log, int max_log)
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
__device__ __forceinline__ void build_fse(uint32_t* tab, const int16_t* norm, int max_sym, int table_log, uint16_t* next)
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
__device__ __forceinline__ int read_huf_table(ZState* z, const uint8_t* p, int len, int* log_out)
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

__device__ __forceinline__ void copy_lits(uint8_t* dst, const Lits& l, uint32_t n, int lane)
{
    if (n == 0) return;
    if (l.is_rle) { for (uint32_t k = lane; k < n; k += 64) dst[k] = l.rle; return; }
    if (n <= 64) { if (uint32_t(lane) < n) dst[lane] = l.p[l.pos + lane]; return; }
    wave_copy(dst, l.p + l.pos, int(n), lane);
}

// One wavefront decodes `csize` bytes of zstd frames into dst[0..cap).  Returns bytes or < 0.

// Which Huffman decoder the reference takes for a 4-stream compressed literals section (huf_decompress.c:1595-1617, timings :1568-1587):
// 1 = the double-symbol decoder (X2).  On valid streams both decoders give the same bytes; they differ in what they make of the LAST
// symbol of a stream (see huf_stream_end_ok).
__device__ __forceinline__ bool huf_select_x2(uint32_t dst_size, uint32_t csrc_size)
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

__device__ __forceinline__ int zstd_decode_frames(const uint8_t* src, int csize, uint8_t* dst, int cap, uint8_t* litbuf,
                                  ZState* z, int lane)
{
    int ip = 0, op = 0;
    uint64_t t_lit = 0, t_hdr = 0, t_seq = 0, t_exec = 0, t0 = __builtin_readcyclecounter(), t1;   // phase cycle counters (profiling aid)
#define ZPH(acc) do { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } while (0)
    if (lane < 36) { z->llb[lane] = kLLBase[lane]; z->llx[lane] = kLLBits[lane]; }
    if (lane < 53) { z->mlb[lane] = kMLBase[lane]; z->mlx[lane] = kMLBits[lane]; }
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
        uint32_t pv = 0; int p_base = 0, p_n = 0;                     // pending 64-byte output step of the batch copier
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
                for (uint32_t k = lane; k < bsize; k += 64) dst[op + k] = v;
                op += int(bsize);
            } else {
                // ---------------------------------------------------------------- compressed block
                if (bsize >= uint32_t(kBlockMax)) return kErr;         // zstd_decompress_block.c:2021
                if (bsize > uint32_t(csize - ip) || bsize < 2) return kErr;
                const uint8_t* bp = src + ip;
                const int bend = int(bsize);
                int bpos = 0;
                Lits lits; lits.pos = 0; lits.is_rle = false; lits.rle = 0; lits.p = litbuf; lits.size = 0;
                ZPH(t_hdr);
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
                            used = __builtin_amdgcn_readfirstlane(used);
                            huf_log = __builtin_amdgcn_readfirstlane(huf_log);
                            if (used < 0) return kErr;
                            hp8 += used; hlen -= used;
                            have_huf = true;
                            huf_x2 = !one_stream && huf_select_x2(lsize, lcsize);     // zstd_decompress_block.c:183-205: one stream -> X1, four -> by size
                        }
                        // stream layout: one stream, or 6-byte jump table + four streams (huf_decompress.c:561-590)
                        int s_off = 0, s_len = hlen, o_off = 0, o_len = int(lsize);
                        bool ok = true;
                        if (!one_stream) {
                            if (hlen < 10 || lsize < 6) return kErr;
                            const int l1 = hp8[0] | (hp8[1] << 8), l2 = hp8[2] | (hp8[3] << 8), l3 = hp8[4] | (hp8[5] << 8);
                            const int l4 = hlen - 6 - l1 - l2 - l3;
                            if (l4 < 1) return kErr;
                            const int seg = (int(lsize) + 3) / 4;
                            if (3 * seg > int(lsize)) return kErr;
                            const int j = lane & 3;
                            s_off = 6 + (j > 0 ? l1 : 0) + (j > 1 ? l2 : 0) + (j > 2 ? l3 : 0);
                            s_len = j == 0 ? l1 : (j == 1 ? l2 : (j == 2 ? l3 : l4));
                            o_off = seg * j; o_len = (j == 3) ? int(lsize) - 3 * seg : seg;
                        }
                        if (lane < (one_stream ? 1 : 4)) {
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
                        if (__ballot(!ok)) return kErr;
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
                ZPH(t_lit);
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
                            used = __builtin_amdgcn_readfirstlane(used); tl = __builtin_amdgcn_readfirstlane(tl);
                            if (used < 0) return kErr;
                            bpos += used; *logp = tl;
                        } else {                                       // repeat the previous block's table
                            if (!have_fse) return kErr;
                        }
                    }
                    have_fse = true;
                    for (int u = lane; u < (1 << ll_log); u += 64) { const uint32_t c = z->ll[u] & 0xff; z->llv[u] = c < 36 ? z->llb[c] | (uint32_t(z->llx[c]) << 20) : 0xFFFFFFFFu; }
                    for (int u = lane; u < (1 << ml_log); u += 64) { const uint32_t c = z->ml[u] & 0xff; z->mlv[u] = c < 53 ? z->mlb[c] | (uint32_t(z->mlx[c]) << 20) : 0xFFFFFFFFu; }
                    // ---- sequence bitstream (zstd_decompress_block.c:1565-1650): sequences are decoded serially
                    // (three FSE states) into lanes, then executed up to 64 at a time by the batch copier below.
                    ZPH(t_hdr);
                    SeqBits bs;
                    if (!bs.init(bp + bpos, bend - bpos, z->ring, lane)) return kErr;
                    bs.refill();
                    uint32_t sl = bs.read(ll_log), so = bs.read(of_log), sm = bs.read(ml_log);
                    int n = 0;
                    bool have_c = false; uint32_t c_ll = 0, c_ml = 0, c_off = 0;
                    for (;;) {
                        uint32_t my_ll = 0, my_ml = 0, my_off = 0, T = 0, Lsum = 0;
                        int cnt = 0;
                        for (;;) {
                            uint32_t llen, mlen, offset;
                            if (have_c) { llen = c_ll; mlen = c_ml; offset = c_off; have_c = false; }
                            else if (n < nseq) {
                                bs.refill();
                                const uint32_t el = z->ll[sl], eo = z->of[so], em = z->ml[sm];
                                const uint32_t lv = z->llv[sl], mv = z->mlv[sm];
                                const uint32_t ocode = eo & 0xff;
                                if (ocode > 31) return kErr;
                                const uint32_t ll_base = lv & 0xFFFFF;
                                if (ocode > 1) {
                                    offset = (1u << ocode) - 3 + bs.read(int(ocode), true);  // OF_base[code] = 2^code - 3
                                    rep2 = rep1; rep1 = rep0; rep0 = offset;
                                } else {
                                    const uint32_t ll0 = (ll_base == 0);
                                    if (ocode == 0) {
                                        offset = ll0 ? rep1 : rep0;
                                        rep1 = ll0 ? rep0 : rep1; rep0 = offset;
                                    } else {
                                        const uint32_t v = 1 + ll0 + bs.read(1, true);    // OF_base[1] = 1
                                        uint32_t t = (v == 3) ? rep0 - 1 : (v == 1 ? rep1 : rep2);
                                        t += !t;
                                        if (v != 1) rep2 = rep1;
                                        rep1 = rep0; rep0 = offset = t;
                                    }
                                }
                                mlen = (mv & 0xFFFFF) + bs.read(int(mv >> 20), true);
                                bs.refill();
                                llen = ll_base + bs.read(int(lv >> 20), true);
                                sl = (el >> 16) + bs.read(int((el >> 8) & 0xff));
                                sm = (em >> 16) + bs.read(int((em >> 8) & 0xff));
                                so = (eo >> 16) + bs.read(int((eo >> 8) & 0xff));
                                n++;
                            } else break;
                            if (cnt == 64 || T + llen + mlen > uint32_t(kOwnBytes)) { have_c = true; c_ll = llen; c_ml = mlen; c_off = offset; break; }
                            if (lane == cnt) { my_ll = llen; my_ml = mlen; my_off = offset; }
                            cnt++; T += llen + mlen; Lsum += llen;
                        }
                        ZPH(t_seq);
                        if (cnt > 0 && Lsum <= lits.size - lits.pos && uint32_t(op) + T + 64 <= uint32_t(cap)) {
                            // ---- batch execute: every output byte finds its sequence through the owner map; literal
                            // bytes come from the literal buffer, match bytes from memory / the pending step / this step
                            const uint32_t sz = my_ll + my_ml;
                            const uint32_t ostart = scan_add(sz) - sz, lstart = scan_add(my_ll) - my_ll;
                            const bool bad = lane < cnt && my_off > uint32_t(op) + ostart + my_ll;
                            if (__ballot(bad)) return kErr;
                            for (uint32_t k = 4u * lane; k < T; k += 256) *reinterpret_cast<uint32_t*>(z->own + k) = 0;
                            if (lane < cnt) z->own[ostart] = uint8_t(lane + 1);
                            const uint8_t* const lbase = lits.is_rle ? dst : lits.p + lits.pos;
                            uint32_t carry = 0;
                            if (p_n == 0) p_base = op;
                            for (uint32_t c0 = 0; c0 < T; c0 += 64) {
                                const uint32_t o = c0 + lane;
                                const bool live = o < T;
                                uint32_t m = live ? uint32_t(z->own[o]) : 0u;
                                m = max(scan_max(m), carry);
                                carry = uint32_t(__builtin_amdgcn_readlane(int(m), 63));
                                const int tl = (int(m) - 1) & 63;
                                const uint32_t os = __shfl(ostart, tl), lt = __shfl(my_ll, tl), offt = __shfl(my_off, tl), ls = __shfl(lstart, tl);
                                const uint32_t rel = o - os;
                                const bool is_lit = live && rel < lt;
                                const int sp = op + int(o) - int(offt);
                                const int cs = op + int(c0);
                                const bool is_match = live && !is_lit;
                                const bool from_mem = is_match && sp < cs - p_n;
                                const bool in_pend = is_match && sp >= cs - p_n && sp < cs;
                                const uint8_t* const addr = (is_lit && !lits.is_rle) ? lbase + ls + rel : dst + (from_mem ? sp : 0);
                                const uint32_t ld = *addr;
                                dst[p_base + lane] = uint8_t(pv);                 // previous step, all 64 lanes (see lz4_decode.hip)
                                const uint32_t fw = __shfl(pv, (sp - p_base) & 63);
                                uint32_t v = ld;
                                if (is_lit && lits.is_rle) v = lits.rle;
                                if (in_pend) v = fw;
                                bool done = !is_match || from_mem || in_pend;
                                int dep = sp - cs;
                                while (__ballot(!done)) {
                                    const int d = dep & 63;
                                    const uint32_t v2 = __shfl(v, d);
                                    const int dn = __shfl(int(done), d);
                                    const int dd = __shfl(dep, d);
                                    if (!done) { if (dn) { v = v2; done = true; } else dep = dd; }
                                }
                                pv = v; p_base = cs; p_n = min(64, int(T - c0));
                            }
                            op += int(T); lits.pos += Lsum;
                        } else {
                            // ---- one sequence at a time (strict checks; long sequences, block ends)
                            if (lane < p_n) dst[p_base + lane] = uint8_t(pv);
                            p_n = 0;
                            for (int k = 0; k < cnt; k++) {
                                const uint32_t llen = uint32_t(__builtin_amdgcn_readlane(int(my_ll), k)), mlen = uint32_t(__builtin_amdgcn_readlane(int(my_ml), k));
                                const uint32_t offset = uint32_t(__builtin_amdgcn_readlane(int(my_off), k));
                                if (llen > lits.size - lits.pos) return kErr;
                                if (llen + mlen > uint32_t(cap - op)) return kErr;
                                copy_lits(dst + op, lits, llen, lane);
                                lits.pos += llen; op += int(llen);
                                if (offset > uint32_t(op)) return kErr;        // before the start of the output (no dictionary)
                                copy_match(dst, op, int(offset), int(mlen), lane);
                                op += int(mlen);
                            }
                            if (have_c && cnt == 0) {                          // a sequence larger than a whole batch
                                have_c = false;
                                if (c_ll > lits.size - lits.pos) return kErr;
                                if (c_ll + c_ml > uint32_t(cap - op)) return kErr;
                                copy_lits(dst + op, lits, c_ll, lane);
                                lits.pos += c_ll; op += int(c_ll);
                                if (c_off > uint32_t(op)) return kErr;
                                copy_match(dst, op, int(c_off), int(c_ml), lane);
                                op += int(c_ml);
                            }
                        }
                        ZPH(t_exec);
                        if (!have_c && n >= nseq) break;
                    }
                    if (lane < p_n) dst[p_base + lane] = uint8_t(pv);
                    p_n = 0;
                    if (bs.pos > 0) return kErr;                       // bits left over: corruption
                    (void)frame_start;
                }
                ZPH(t_exec);
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
    ZPH(t_exec);
    if (lane == 0) { uint64_t* c = reinterpret_cast<uint64_t*>(litbuf + kBlockMax); c[0] = t_lit; c[1] = t_hdr; c[2] = t_seq; c[3] = t_exec; }
    return op;
}

// ================================================================================================ lane-parallel path
// A 4mz payload is ONE frame of up to 32 inner blocks whose entropy decoding is independent (tables repeat
// only by reference, repcodes are resolved at execution).  Decoding a block's literals and sequences is a
// serial bit/state chain - one lane's worth of work - so here LANE i decodes inner block i: all Huffman and
// FSE chains of a frame advance together, tables live in a per-lane slot of the HBM workspace.  Execution
// (repcode resolution + the batch copier) then runs over the blocks in order with the whole wave.
// Anything unusual (several frames, skippable frames, > 64 blocks, too many sequences) declines and the
// serial path above decides.
constexpr int      kDecline  = -1000000007;
constexpr int      kMaxInner = 64;
constexpr uint32_t kSeqArea  = 1024 * 1024;                   // sequences of one 4mc block held at once, 8 bytes each:
                                                              // ll (18 bits) | ml low 14 << 18, Offset_Value (28 bits) | ml high 4 << 28
constexpr size_t   kV2State  = (sizeof(ZState) + 255) & ~size_t(255);
constexpr size_t   kV2Lit    = (size_t(4) << 20) + kMaxInner * 64 + 256;
constexpr size_t   kV2Bytes  = kMaxInner * kV2State + kV2Lit + size_t(kSeqArea) * 8 + (kBlockMax + 64);

struct V2Info {                                               // LDS, aliases ZState::huf (unused on this path)
    uint32_t off[kMaxInner], size[kMaxInner];                // block content offset / size in the payload
    uint32_t lit_off[kMaxInner], lit_size[kMaxInner];        // regenerated literals: where (src offset or literal area) / how many
    uint32_t nseq[kMaxInner], seq_off[kMaxInner];
    uint8_t  type[kMaxInner], lit_kind[kMaxInner], lit_rle[kMaxInner], bad[kMaxInner];   // lit_kind: 0 in place, 1 rle, 2 literal area
    uint32_t nblk, pad[3];                                   // (for the helper wave of a two-wave launch)
};
static_assert(sizeof(V2Info) <= sizeof(uint16_t) * 4096, "V2Info must fit the aliased table");
constexpr int kPendingExec = -1000000011;                     // block result between the two kernels: entropy stage done, execution owed
constexpr int kRetryZ      = -1000000013;                     // the execute kernel declined: the one-wave kernel decodes the block from scratch
struct V2Pending { uint32_t nblk, pad; uint64_t fcs; V2Info info; };
static_assert(sizeof(V2Pending) <= size_t(kBlockMax), "the pending header lives in the serial path's literal buffer");

struct LitHdr { int ltype; uint32_t lh, lsize, lcsize; bool one; };
__device__ __forceinline__ bool parse_lit_hdr(const uint8_t* bp, int bend, LitHdr& h)
{
    if (bend < 1) return false;
    const uint32_t b0 = bp[0];
    const int fmt = (b0 >> 2) & 3;
    h.ltype = b0 & 3; h.one = false;
    if (h.ltype >= 2) {
        if (bend < 5) return false;
        const uint32_t lhc = uint32_t(bp[0]) | (uint32_t(bp[1]) << 8) | (uint32_t(bp[2]) << 16) | (uint32_t(bp[3]) << 24);
        if (fmt <= 1) { h.one = (fmt == 0); h.lh = 3; h.lsize = (lhc >> 4) & 0x3FF; h.lcsize = (lhc >> 14) & 0x3FF; }
        else if (fmt == 2) { h.lh = 4; h.lsize = (lhc >> 4) & 0x3FFF; h.lcsize = lhc >> 18; }
        else { h.lh = 5; h.lsize = (lhc >> 4) & 0x3FFFF; h.lcsize = (lhc >> 22) + (uint32_t(bp[4]) << 10); }
        return h.lsize <= uint32_t(kBlockMax) && h.lcsize + h.lh <= uint32_t(bend);
    }
    if ((fmt & 1) == 0) { h.lh = 1; h.lsize = b0 >> 3; }
    else if (fmt == 1) { if (bend < 2) return false; h.lh = 2; h.lsize = (b0 | (uint32_t(bp[1]) << 8)) >> 4; }
    else { if (bend < 3) return false; h.lh = 3; h.lsize = (b0 | (uint32_t(bp[1]) << 8) | (uint32_t(bp[2]) << 16)) >> 4; }
    if (h.lsize > uint32_t(kBlockMax)) return false;
    h.lcsize = h.ltype == 0 ? h.lsize : 1;
    return h.lh + h.lcsize <= uint32_t(bend);
}

// sequences header at bp[pos..): count, modes byte; returns false on error.  pos ends behind the modes byte (or the count if 0)
__device__ __forceinline__ bool parse_seq_hdr(const uint8_t* bp, int bend, int& pos, int& nseq, uint32_t& modes)
{
    if (bend - pos < 1) return false;
    nseq = bp[pos++]; modes = 0;
    if (nseq == 0) return pos == bend;
    if (nseq > 0x7F) {
        if (nseq == 0xFF) { if (pos + 2 > bend) return false; nseq = (bp[pos] | (bp[pos + 1] << 8)) + 0x7F00; pos += 2; }
        else { if (pos >= bend) return false; nseq = ((nseq - 0x80) << 8) + bp[pos++]; }
    }
    if (pos + 1 > bend) return false;
    modes = bp[pos++];
    return true;                                                        // (reserved bits: ignored, as the reference does)
}

// builds table t (0 LL, 1 OF, 2 ML) of a block whose sequence header's table descriptions start at bp[pos]; `pos` advances.
// mode 3 is resolved by the caller.  Returns the table log or -1.
__device__ __forceinline__ int build_seq_table_lane(ZState* zl, const uint8_t* bp, int bend, int& pos, int t, int mode)
{
    uint32_t* tab = t == 0 ? zl->ll : (t == 1 ? zl->of : zl->ml);
    const int maxsym = t == 0 ? 35 : (t == 1 ? 31 : 52), maxlog = t == 0 ? 9 : (t == 1 ? 8 : 9);
    if (mode == 1) {
        if (pos >= bend) return -1;
        const uint32_t sy = bp[pos++];
        if (int(sy) > maxsym) return -1;
        tab[0] = sy;
        return 0;
    }
    if (mode == 0) {
        const int dlog = t == 1 ? 5 : 6, dmax = t == 0 ? 35 : (t == 1 ? 28 : 52);
        const int16_t* d = t == 0 ? kLLDef : (t == 1 ? kOFDef : kMLDef);
        for (int i = 0; i <= dmax; i++) zl->norm[i] = d[i];
        build_fse(tab, zl->norm, dmax, dlog, zl->next);
        return dlog;
    }
    int ms = maxsym, tl = 0;
    const int used = read_ncount(bp + pos, bend - pos, zl->norm, &ms, &tl, maxlog);
    if (used < 0) return -1;
    build_fse(tab, zl->norm, ms, tl, zl->next);
    pos += used;
    return tl;
}

// skips the description of one table (to reach a later one in an earlier block's header); false on error
__device__ __forceinline__ bool skip_seq_table(ZState* zl, const uint8_t* bp, int bend, int& pos, int t, int mode)
{
    if (mode == 1) { if (pos >= bend) return false; pos++; return true; }
    if (mode == 2) {
        int ms = t == 0 ? 35 : (t == 1 ? 31 : 52), tl = 0;
        const int used = read_ncount(bp + pos, bend - pos, zl->norm, &ms, &tl, t == 1 ? 8 : 9);
        if (used < 0) return false;
        pos += used;
    }
    return true;
}

// step 1c/1d of the lane-parallel path for ONE lane: the three FSE tables of the lane's block (its own descriptions, or those of the
// block it repeats) in the lane's workspace slot `zl`, then the block's sequences as (ll, ml, Offset_Value) triples into the
// sequence area.  Returns false on corrupt input.
__device__ __forceinline__ bool v2_sequences_lane(ZState* zl, const ZState* z, const V2Info* I, const uint8_t* src, const uint8_t* bp, const int bend,
                                                  const int spos, const int nseq, const uint32_t modes, const uint32_t soff, uint32_t* seqarea, const int lane)
{
    bool bad = false;
    if (nseq > 0) {
        uint32_t* const logs = zl->rank + 12;                      // table logs of LL / OF / ML (rank[] is free after the Huffman build)
        logs[0] = logs[1] = logs[2] = 0;
        int pos = spos;
        for (int t = 0; t < 3 && !bad; t++) {
            int mode = int((modes >> (6 - 2 * t)) & 3);
            if (mode != 3) { const int lg = build_seq_table_lane(zl, bp, bend, pos, t, mode); if (lg < 0) bad = true; else logs[t] = uint32_t(lg); continue; }
            // repeat: the most recent earlier block with sequences defines it (possibly itself by repeating)
            int j = lane - 1; bool found = false;
            for (; j >= 0 && !bad; j--) {
                if (I->type[j] != 2 || I->nseq[j] == 0) continue;
                const uint8_t* bj = src + I->off[j]; const int be = int(I->size[j]);
                LitHdr hj; int pj, nj; uint32_t mj;
                if (!parse_lit_hdr(bj, be, hj)) { bad = true; break; }
                pj = int(hj.lh + hj.lcsize);
                if (!parse_seq_hdr(bj, be, pj, nj, mj)) { bad = true; break; }
                const int mt = int((mj >> (6 - 2 * t)) & 3);
                if (mt == 3) continue;
                for (int u = 0; u < t && !bad; u++) if (!skip_seq_table(zl, bj, be, pj, u, int((mj >> (6 - 2 * u)) & 3))) bad = true;
                if (!bad) { const int lg = build_seq_table_lane(zl, bj, be, pj, t, mt); if (lg < 0) bad = true; else logs[t] = uint32_t(lg); }
                found = true;
                break;
            }
            if (!found) bad = true;
        }
        if (!bad) {
            const int ll_log = int(logs[0]), of_log = int(logs[1]), ml_log = int(logs[2]);
            // one 8-byte entry per LL / ML state: {FSE entry, baseline | extra bits << 20}, in the slot's Huffman table space (free
            // here): a sequence costs three table reads instead of five - at 2048 frames x 32 lanes every one of them goes to memory
            uint2* const cl = reinterpret_cast<uint2*>(zl->huf);
            uint2* const cm = cl + 512;
            for (int u = 0; u < (1 << ll_log); u++) { const uint32_t e = zl->ll[u], c = e & 0xff; cl[u] = make_uint2(e, z->llb[c] | (uint32_t(z->llx[c]) << 20)); }
            for (int u = 0; u < (1 << ml_log); u++) { const uint32_t e = zl->ml[u], c = e & 0xff; cm[u] = make_uint2(e, z->mlb[c] | (uint32_t(z->mlx[c]) << 20)); }
            BitsBack bs;
            if (!bs.init(bp + pos, bend - pos)) bad = true;
            else {
                uint32_t sl = bs.read(ll_log), so = bs.read(of_log), sm = bs.read(ml_log);
                uint32_t* out = seqarea + size_t(soff) * 2;
                for (int n = 0; n < nseq; n++) {
                    const uint2 cle = cl[sl], cme = cm[sm];
                    const uint32_t el = cle.x, eo = zl->of[so], em = cme.x;
                    const uint32_t lv = cle.y, mv = cme.y;
                    const uint32_t ocode = eo & 0xff;
                    if (ocode > 31) { bad = true; break; }
                    const uint32_t ov = (1u << ocode) + bs.read_fast(int(ocode));            // Offset_Value: 1..3 repeat codes, else offset + 3
                    const uint32_t mlen = (mv & 0xFFFFF) + bs.read_fast(int(mv >> 20));
                    const uint32_t llen = (lv & 0xFFFFF) + bs.read_fast(int(lv >> 20));
                    sl = (el >> 16) + bs.read(int((el >> 8) & 0xff));
                    sm = (em >> 16) + bs.read(int((em >> 8) & 0xff));
                    so = (eo >> 16) + bs.read(int((eo >> 8) & 0xff));
                    if (ov >> 28) { bad = true; break; }                                   // no such distance inside one 4mc block
                    *reinterpret_cast<uint2*>(out + 2 * n) = make_uint2(llen | (mlen << 18), ov | ((mlen >> 14) << 28));
                }
                if (bs.pos > 0) bad = true;                        // bits left over: corruption
            }
        }
    }
    return !bad;
}


// The helper wave of a two-wave launch (mode 1): waits for the first wave's go (block table complete), decodes the sequences of
// every inner block - lane i block i, in the upper half of the workspace slots - and reports.
// flags[0]: 0 wait, 1 go (block table complete), 2 nothing to do;  flags[1]: 0 running, 1 done, 2 corrupt input
__device__ __forceinline__ void v2_helper_wave(const uint8_t* src, uint8_t* work, ZState* z, uint32_t* flags, int lane)
{
    V2Info* const I = reinterpret_cast<V2Info*>(z->huf);
    for (uint32_t spins = 0;; spins++) {
        const uint32_t g = __hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (g == 1) break;
        if (g == 2) return;
        __builtin_amdgcn_s_sleep(8);
        if (spins > (1u << 22)) return;                                // (the first wave gives up on seq_done the same way)
    }
    const int nblk = int(I->nblk);
    ZState* const zl = reinterpret_cast<ZState*>(work + size_t(lane + 32) * kV2State);
    uint8_t* const litarea = work + kMaxInner * kV2State;
    uint32_t* const seqarea = reinterpret_cast<uint32_t*>(litarea + kV2Lit);
    bool bad = false;
    if (lane < nblk && I->type[lane] == 2 && I->nseq[lane] > 0) {
        const uint8_t* const bp = src + I->off[lane];
        const int bend = int(I->size[lane]);
        LitHdr lh; int spos, nseq; uint32_t modes;
        if (!parse_lit_hdr(bp, bend, lh)) bad = true;                  // (validated by the first wave already)
        else {
            spos = int(lh.lh + lh.lcsize);
            if (!parse_seq_hdr(bp, bend, spos, nseq, modes)) bad = true;
            else bad = !v2_sequences_lane(zl, z, I, src, bp, bend, spos, nseq, modes, I->seq_off[lane], seqarea, lane);
        }
    }
    const bool anybad = __ballot(bad) != 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                  // the triples are in memory before the report is seen
    if (lane == 0) __hip_atomic_store(&flags[1], anybad ? 2u : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int zstd_decode_frame_v2(const uint8_t* src, int csize, uint8_t* dst, int cap, uint8_t* work, ZState* z, int lane, const bool split, uint32_t* flags)
{
    const bool two_wave = flags != nullptr;
    V2Info* const I = reinterpret_cast<V2Info*>(z->huf);
    ZState* const zl = reinterpret_cast<ZState*>(work + size_t(lane) * kV2State);
    uint8_t* const litarea = work + kMaxInner * kV2State;
    uint32_t* const seqarea = reinterpret_cast<uint32_t*>(litarea + kV2Lit);
    uint64_t t_lit = 0, t_hdr = 0, t_seq = 0, t_exec = 0, t0 = __builtin_readcyclecounter(), t1;   // phase cycle counters (profiling aid)
    // ---------------------------------------------------------------- step 0: frame header + block headers (wave-uniform)
    if (csize < 9) return kDecline;
    const uint32_t magic = uint32_t(src[0]) | (uint32_t(src[1]) << 8) | (uint32_t(src[2]) << 16) | (uint32_t(src[3]) << 24);
    if (magic != 0xFD2FB528u) return kDecline;
    const uint32_t fhd = src[4];
    const int fcs_id = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    if ((fhd & 0x0C) || did) return kDecline;                          // reserved bit, checksum, dictionary: the serial path decides
    const int fcs_sz = fcs_id == 0 ? single : (1 << fcs_id);
    const int hsize = 5 + (single ? 0 : 1) + fcs_sz;
    if (csize < hsize) return kDecline;
    int hp = 5;
    if (!single) { const uint32_t wb = src[hp++]; if (int(wb >> 3) + 10 > 31) return kDecline; }
    uint64_t fcs = ~0ull;
    if (fcs_id == 0) { if (single) fcs = src[hp++]; }
    else { fcs = 0; for (int i = 0; i < fcs_sz; i++) fcs |= uint64_t(src[hp++]) << (8 * i); if (fcs_id == 1) fcs += 256; }
    if (fcs != ~0ull && fcs > uint64_t(cap)) return kErr;              // dstSize_tooSmall
    int ip = hsize, nblk = 0;
    for (;;) {
        if (csize - ip < 3) return kDecline;
        const uint32_t bh = uint32_t(src[ip]) | (uint32_t(src[ip + 1]) << 8) | (uint32_t(src[ip + 2]) << 16);
        ip += 3;
        const int last = bh & 1, btype = (bh >> 1) & 3;
        const uint32_t bsize = bh >> 3, content = btype == 1 ? 1u : bsize;
        if (btype == 3 || nblk == kMaxInner || content > uint32_t(csize - ip)) return kDecline;
        if (btype == 2 && (bsize >= uint32_t(kBlockMax) || bsize < 2)) return kDecline;
        if (lane == 0) { I->off[nblk] = uint32_t(ip); I->size[nblk] = bsize; I->type[nblk] = uint8_t(btype); }
        nblk++; ip += int(content);
        if (last) break;
    }
    if (ip != csize) return kDecline;                                  // more frames follow
    if (lane < 36) { z->llb[lane] = kLLBase[lane]; z->llx[lane] = kLLBits[lane]; }
    if (lane < 53) { z->mlb[lane] = kMLBase[lane]; z->mlx[lane] = kMLBits[lane]; }
    // ---------------------------------------------------------------- step 1a: literal / sequence headers, one block per lane
    const bool mine = lane < nblk && I->type[lane < nblk ? lane : 0] == 2;
    const uint8_t* const bp = src + (lane < nblk ? I->off[lane] : 0u);
    const int bend = lane < nblk ? int(I->size[lane]) : 0;
    LitHdr lh; lh.ltype = 0; lh.lh = 0; lh.lsize = 0; lh.lcsize = 0; lh.one = false;
    int spos = 0, nseq = 0; uint32_t modes = 0;
    bool bad = false;
    if (mine) {
        bad = !parse_lit_hdr(bp, bend, lh);
        if (!bad) { spos = int(lh.lh + lh.lcsize); bad = !parse_seq_hdr(bp, bend, spos, nseq, modes); }
    }
    if (__ballot(bad)) return kErr;
    {
        const uint32_t need = (mine && lh.ltype >= 2) ? ((lh.lsize + 63u) & ~63u) : 0u;
        const uint32_t loff = scan_add(need) - need, soff = scan_add(uint32_t(nseq)) - uint32_t(nseq);
        if (uint32_t(__builtin_amdgcn_readlane(int(soff + uint32_t(nseq)), 63)) > kSeqArea) return kDecline;
        if (uint32_t(__builtin_amdgcn_readlane(int(loff + need), 63)) > uint32_t(kV2Lit) - 256u) return kDecline;
        if (lane < nblk) {
            I->nseq[lane] = uint32_t(nseq); I->seq_off[lane] = soff; I->lit_size[lane] = lh.lsize;
            I->lit_kind[lane] = uint8_t(lh.ltype == 0 ? 0 : lh.ltype == 1 ? 1 : 2);
            I->lit_off[lane] = lh.ltype >= 2 ? loff : I->off[lane] + lh.lh;
            I->lit_rle[lane] = (mine && lh.ltype == 1) ? bp[lh.lh] : uint8_t(0);
        }
        const bool helper_go = two_wave && nblk <= 32;
        if (helper_go) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) { I->nblk = uint32_t(nblk); __hip_atomic_store(&flags[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
        ZPH(t_hdr);
        // ------------------------------------------------------------ step 1b: Huffman literals (table of this block or of the block it repeats)
        // With <= 32 inner blocks the upper half-wave would idle: lane L works on block L & 31 and decodes the stream
        // pair L >> 5 of its four Huffman streams (both halves read the same table: see below).
        const bool split = nblk <= 32 && !helper_go;                             // (the helper wave owns the upper workspace slots)
        const int hb = split ? (lane & 31) : lane, half = split ? (lane >> 5) : 0;
        LitHdr hh = lh; const uint8_t* hbp = bp; uint32_t hloff = loff; bool hmine = mine;
        if (split && lane >= 32) {
            hmine = hb < nblk && I->type[hb < nblk ? hb : 0] == 2;
            hbp = src + (hb < nblk ? I->off[hb] : 0u);
            hh.ltype = 0;
            if (hmine) { parse_lit_hdr(hbp, int(I->size[hb]), hh); hloff = I->lit_off[hb]; }      // already validated by lane hb
        }
        // Huffman tables.  Built once per DEFINING block (literal type 2, by its own lane, in its workspace slot); a block of type 3
        // ("treeless") reads the table of the nearest defining block before it - text-like frames have one table for all 32 blocks.
        // The first three tables are copied into LDS (the FSE table space of the one-wave path, idle here): a symbol is a chain of
        // table read -> bits -> next read, 130 clk per link from LDS against a trip to memory (2048 frames x 32 lanes x 4 KiB of
        // tables do not stay in any cache).
        int huf_log = 0, used = 0;
        const bool definer = hmine && hh.ltype == 2 && half == 0;
        if (definer) { used = read_huf_table(zl, hbp + hh.lh, int(hh.lcsize), &huf_log); if (used < 0) bad = true; }
        if (__ballot(bad)) return kDecline;             // (the serial path decides: it has the reference's end rules of both Huffman decoders)
        const unsigned long long defmask = __ballot(definer);
        int defj = -1;
        if (hmine && hh.ltype >= 2) {
            const unsigned long long cand = hh.ltype == 2 ? (1ull << hb) : (defmask & ((1ull << hb) - 1));
            if (cand) defj = 63 - __builtin_clzll(cand); else bad = true;
        }
        if (__ballot(bad)) return kDecline;             // (the serial path decides: it has the reference's end rules of both Huffman decoders)
        const int dj = defj < 0 ? 0 : defj;
        const int my_log = __shfl(huf_log, dj);
        const int used_dj = __shfl(used, dj);
        const int my_used = hh.ltype == 2 ? used_dj : 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");             // the tables are in memory before other lanes read them
        const uint16_t* tab = reinterpret_cast<const ZState*>(work + size_t(dj) * kV2State)->huf;
        {
            uint16_t* const slot[3] = {reinterpret_cast<uint16_t*>(z->ll), reinterpret_cast<uint16_t*>(z->llv), z->huf + 1024};
            int ns = 0;
            for (unsigned long long dm = defmask; dm && ns < 3; dm &= dm - 1) {
                const int j = __builtin_ctzll(dm);
                const int lg = __builtin_amdgcn_readlane(huf_log, j);
                if (lg > 11) continue;                                 // 8 KiB: stays in memory
                const uint8_t* from = reinterpret_cast<const uint8_t*>(reinterpret_cast<const ZState*>(work + size_t(j) * kV2State)->huf);
                const int bytes = max(2 << lg, 16);
                for (int k = 16 * lane; k < bytes; k += 1024)
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(slot[ns]) + k) = *reinterpret_cast<const uint4*>(from + k);
                if (defj == j) tab = slot[ns];
                ns++;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const int huf_log_mine = my_log;
        if (hmine && hh.ltype >= 2) {
            {
                const uint8_t* hp8 = hbp + hh.lh + my_used;
                const int hlen = int(hh.lcsize) - my_used;
                uint8_t* const out = litarea + hloff;
                int nstreams = 1, l1 = 0, l2 = 0, l3 = 0, seg = int(hh.lsize);
                if (!hh.one) {
                    if (hlen < 10 || hh.lsize < 6) bad = true;
                    else {
                        l1 = hp8[0] | (hp8[1] << 8); l2 = hp8[2] | (hp8[3] << 8); l3 = hp8[4] | (hp8[5] << 8);
                        seg = (int(hh.lsize) + 3) / 4; nstreams = 4;
                        if (hlen - 6 - l1 - l2 - l3 < 1 || 3 * seg > int(hh.lsize)) bad = true;
                    }
                }
                const int j0 = split ? (nstreams == 4 ? 2 * half : 0) : 0, j1 = split ? (nstreams == 4 ? 2 * half + 2 : (half == 0 ? 1 : 0)) : nstreams;
                for (int j = j0; j < j1 && !bad; j++) {
                    const int s_off = nstreams == 1 ? 0 : 6 + (j > 0 ? l1 : 0) + (j > 1 ? l2 : 0) + (j > 2 ? l3 : 0);
                    const int s_len = nstreams == 1 ? hlen : (j == 0 ? l1 : (j == 1 ? l2 : (j == 2 ? l3 : hlen - 6 - l1 - l2 - l3)));
                    const int o_off = nstreams == 1 ? 0 : seg * j, o_len = nstreams == 1 ? int(hh.lsize) : ((j == 3) ? int(hh.lsize) - 3 * seg : seg);
                    BitsBack bs;
                    if (!bs.init(hp8 + s_off, s_len)) { bad = true; break; }
                    auto symbol = [&]() -> uint32_t {
                        const uint32_t idx = (bs.pos >= huf_log_mine) ? bs.peek_at(bs.pos - huf_log_mine, huf_log_mine)
                                                                 : (bs.peek_at(0, bs.pos > 0 ? bs.pos : 0) << (huf_log_mine - (bs.pos > 0 ? bs.pos : 0)));
                        const uint32_t e = tab[idx];
                        bs.pos -= int(e >> 8);
                        return e & 0xff;
                    };
                    // four symbols per store: a byte store per symbol and lane is a partial-sector write each (2048 frames x 32
                    // lanes of them at a time)
                    struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
                    int i = 0;
                    for (; i + 4 <= o_len; i += 4) {
                        uint32_t w4 = symbol(); w4 |= symbol() << 8; w4 |= symbol() << 16; w4 |= symbol() << 24;
                        reinterpret_cast<U4*>(out + o_off + i)->v = w4;
                    }
                    for (; i < o_len; i++) out[o_off + i] = uint8_t(symbol());
                    if (bs.pos != 0) bad = true;
                }
            }
        }
        if (__ballot(bad)) return kDecline;             // (the serial path decides: it has the reference's end rules of both Huffman decoders)
        ZPH(t_lit);
        // ------------------------------------------------------------ step 1c/1d: sequence tables, then the sequences themselves
        // (two-wave launches: the helper wave has been decoding the sequences of every block while this one decoded the literals)
        if (helper_go) {
            for (uint32_t spins = 0;; spins++) {
                const uint32_t d = __hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (d) { if (d == 2) bad = true; break; }
                __builtin_amdgcn_s_sleep(16);
                if (spins > (1u << 22)) { bad = true; break; }
            }
        } else if (mine && nseq > 0) bad = !v2_sequences_lane(zl, z, I, src, bp, bend, spos, nseq, modes, soff, seqarea, lane);
        if (__ballot(bad)) return kErr;
    }
    ZPH(t_seq);
    if (split && cap <= (4 << 20)) {
        // the execution belongs to the second kernel (zstd_exec_kernel): leave what it needs - the block table - next to the
        // literal and sequence areas (in the slot's tail, which only the serial path uses)
        V2Pending* const P = reinterpret_cast<V2Pending*>(work + kV2Bytes - (kBlockMax + 64));
        const uint32_t* from = reinterpret_cast<const uint32_t*>(I);
        uint32_t* to = reinterpret_cast<uint32_t*>(&P->info);
        for (uint32_t i = lane; i < sizeof(V2Info) / 4; i += 64) to[i] = from[i];
        if (lane == 0) { P->nblk = uint32_t(nblk); P->fcs = fcs; uint64_t* c = reinterpret_cast<uint64_t*>(work + kV2Bytes - 64); c[0] = t_lit; c[1] = t_hdr; c[2] = t_seq; c[3] = 0; }
        return kPendingExec;
    }
    // ---------------------------------------------------------------- step 2: execute the blocks in order
    int op = 0;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    uint32_t pv = 0; int p_base = 0, p_n = 0;
    for (int b = 0; b < nblk; b++) {
        const int btype = I->type[b];
        const uint32_t bsize = I->size[b];
        if (btype == 0) {
            if (bsize > uint32_t(cap - op)) return kErr;
            wave_copy(dst + op, src + I->off[b], int(bsize), lane);
            op += int(bsize);
            continue;
        }
        if (btype == 1) {
            if (bsize > uint32_t(cap - op)) return kErr;
            const uint8_t v = src[I->off[b]];
            for (uint32_t k = lane; k < bsize; k += 64) dst[op + k] = v;
            op += int(bsize);
            continue;
        }
        Lits lits; lits.pos = 0; lits.size = I->lit_size[b]; lits.is_rle = I->lit_kind[b] == 1; lits.rle = I->lit_rle[b];
        lits.p = I->lit_kind[b] == 2 ? litarea + I->lit_off[b] : src + I->lit_off[b];
        const int nsq = int(I->nseq[b]);
        const uint32_t* sq = seqarea + size_t(I->seq_off[b]) * 2;
        int n = 0;
        while (n < nsq) {
            // ---- 64 sequences per coalesced load.  How many fit one batch follows from the lengths alone (prefix sum);
            // repeat offsets are the only serial part and cost work only where a repeat code actually occurs: runs of
            // real offsets between two repeat codes just shift the history.
            const int take = min(64, nsq - n);
            uint32_t my_ll = 0, my_ml = 0, r_ov = 4;
            if (lane < take) {
                const uint32_t w0 = sq[2 * (n + lane)], w1 = sq[2 * (n + lane) + 1];
                my_ll = w0 & 0x3FFFF; my_ml = (w0 >> 18) | ((w1 >> 28) << 14); r_ov = w1 & 0x0FFFFFFF;
            }
            const uint32_t incl = scan_add(my_ll + my_ml);
            const unsigned long long over = __ballot(lane < take && incl > uint32_t(kOwnBytes));
            int cnt = over ? __builtin_ctzll(over) : take;
            const bool single = cnt == 0;                              // one sequence larger than a whole batch: executed alone below
            if (single) cnt = 1;
            if (lane >= cnt) { my_ll = 0; my_ml = 0; }
            const uint32_t T = uint32_t(__builtin_amdgcn_readlane(int(incl), cnt - 1));
            const uint32_t Lsum = uint32_t(__builtin_amdgcn_readlane(int(scan_add(my_ll)), 63));
            uint32_t my_off = r_ov - 3;                                // real offsets; repeat codes are patched below
            {
                const unsigned long long rmask = __ballot(lane < cnt && r_ov <= 3);
                int prev = -1;                                         // last position whose effect is folded into rep0..2
                auto advance = [&](int upto) {                         // fold the real offsets of positions (prev, upto) into the history
                    const int g = upto - 1 - prev;
                    if (g >= 3) { rep0 = uint32_t(__builtin_amdgcn_readlane(int(my_off), upto - 1)); rep1 = uint32_t(__builtin_amdgcn_readlane(int(my_off), upto - 2)); rep2 = uint32_t(__builtin_amdgcn_readlane(int(my_off), upto - 3)); }
                    else if (g == 2) { rep2 = rep0; rep0 = uint32_t(__builtin_amdgcn_readlane(int(my_off), upto - 1)); rep1 = uint32_t(__builtin_amdgcn_readlane(int(my_off), upto - 2)); }
                    else if (g == 1) { rep2 = rep1; rep1 = rep0; rep0 = uint32_t(__builtin_amdgcn_readlane(int(my_off), upto - 1)); }
                };
                for (unsigned long long todo = rmask; todo; todo &= todo - 1) {
                    const int k = __builtin_ctzll(todo);
                    advance(k);
                    const uint32_t ov = uint32_t(__builtin_amdgcn_readlane(int(r_ov), k));
                    const uint32_t idx = ov - 1 + (uint32_t(__builtin_amdgcn_readlane(int(my_ll), k)) == 0 ? 1u : 0u);
                    uint32_t offset;
                    if (idx == 0) offset = rep0;
                    else {
                        uint32_t t = idx == 1 ? rep1 : (idx == 2 ? rep2 : rep0 - 1);
                        t += !t;
                        if (idx != 1) rep2 = rep1;
                        rep1 = rep0; rep0 = offset = t;
                    }
                    if (lane == k) my_off = offset;
                    prev = k;
                }
                advance(cnt);
            }
            n += cnt;
            if (!single && Lsum <= lits.size - lits.pos && uint32_t(op) + T + 64 <= uint32_t(cap)) {
                const uint32_t sz = my_ll + my_ml;
                const uint32_t ostart = scan_add(sz) - sz, lstart = scan_add(my_ll) - my_ll;
                const bool badq = lane < cnt && my_off > uint32_t(op) + ostart + my_ll;
                if (__ballot(badq)) return kErr;
                for (uint32_t k = 4u * lane; k < T; k += 256) *reinterpret_cast<uint32_t*>(z->own + k) = 0;
                if (lane < cnt) z->own[ostart] = uint8_t(lane + 1);
                const uint8_t* const lbase = lits.is_rle ? dst : lits.p + lits.pos;
                uint32_t carry = 0;
                if (p_n == 0) p_base = op;
                for (uint32_t c0 = 0; c0 < T; c0 += 64) {
                    const uint32_t o = c0 + lane;
                    const bool live = o < T;
                    uint32_t m = live ? uint32_t(z->own[o]) : 0u;
                    m = max(scan_max(m), carry);
                    carry = uint32_t(__builtin_amdgcn_readlane(int(m), 63));
                    const int tl = (int(m) - 1) & 63;
                    const uint32_t os = __shfl(ostart, tl), lt = __shfl(my_ll, tl), offt = __shfl(my_off, tl), ls = __shfl(lstart, tl);
                    const uint32_t rel = o - os;
                    const bool is_lit = live && rel < lt;
                    const int sp = op + int(o) - int(offt);
                    const int cs = op + int(c0);
                    const bool is_match = live && !is_lit;
                    const bool from_mem = is_match && sp < cs - p_n;
                    const bool in_pend = is_match && sp >= cs - p_n && sp < cs;
                    const uint8_t* const addr = (is_lit && !lits.is_rle) ? lbase + ls + rel : dst + (from_mem ? sp : 0);
                    const uint32_t ld = *addr;
                    dst[p_base + lane] = uint8_t(pv);
                    const uint32_t fw = __shfl(pv, (sp - p_base) & 63);
                    uint32_t v = ld;
                    if (is_lit && lits.is_rle) v = lits.rle;
                    if (in_pend) v = fw;
                    bool done = !is_match || from_mem || in_pend;
                    int dep = sp - cs;
                    while (__ballot(!done)) {
                        const int d = dep & 63;
                        const uint32_t v2 = __shfl(v, d);
                        const int dn = __shfl(int(done), d);
                        const int dd = __shfl(dep, d);
                        if (!done) { if (dn) { v = v2; done = true; } else dep = dd; }
                    }
                    pv = v; p_base = cs; p_n = min(64, int(T - c0));
                }
                op += int(T); lits.pos += Lsum;
            } else {
                if (lane < p_n) dst[p_base + lane] = uint8_t(pv);
                p_n = 0;
                for (int k = 0; k < cnt; k++) {
                    const uint32_t llen = uint32_t(__builtin_amdgcn_readlane(int(my_ll), k)), mlen = uint32_t(__builtin_amdgcn_readlane(int(my_ml), k));
                    const uint32_t offset = uint32_t(__builtin_amdgcn_readlane(int(my_off), k));
                    if (llen > lits.size - lits.pos) return kErr;
                    if (llen + mlen > uint32_t(cap - op)) return kErr;
                    copy_lits(dst + op, lits, llen, lane);
                    lits.pos += llen; op += int(llen);
                    if (offset > uint32_t(op)) return kErr;
                    copy_match(dst, op, int(offset), int(mlen), lane);
                    op += int(mlen);
                }
            }
        }
        if (lane < p_n) dst[p_base + lane] = uint8_t(pv);
        p_n = 0;
        {
            const uint32_t rest = lits.size - lits.pos;
            if (rest > uint32_t(cap - op)) return kErr;
            copy_lits(dst + op, lits, rest, lane);
            op += int(rest);
        }
    }
    if (fcs != ~0ull && uint64_t(op) != fcs) return kErr;
    ZPH(t_exec);
    if (lane == 0) { uint64_t* c = reinterpret_cast<uint64_t*>(work + kV2Bytes - 64); c[0] = t_lit; c[1] = t_hdr; c[2] = t_seq; c[3] = t_exec; }
    return op;
}

// container_mode as in lz4_decode.hip (BADSUM skip, stored copy, negative -> CORRUPT)
// mode 0: the whole decode in this kernel; 1: entropy stage here, execution in zstd_exec_kernel (result = kPendingExec) unless the
// frame is of a shape only the serial path takes; 2: only the blocks the execute kernel handed back (result == kRetryZ), whole decode
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2)))
void zstd_decode_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks,
                        uint32_t nblocks, uint8_t* scratch, int container_mode, int mode)
{
    __shared__ ZState zs;
    __shared__ uint32_t flags[2];
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    if (mode == 2 && blk.result != kRetryZ) return;
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    uint8_t* const work = scratch + size_t(b) * kV2Bytes;
    uint8_t* litbuf = work + kV2Bytes - (kBlockMax + 64);              // the serial path's literal buffer: the tail of the slot
    const bool split = mode == 1;
    const bool two_wave = blockDim.x > 64;                              // mode 1 launches: a helper wave for the sequences
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if (container_mode && blk.result == FOURMC_BLK_BADSUM) return;
    if (two_wave) {
        if (threadIdx.x == 0) { flags[0] = 0; flags[1] = 0; }
        __syncthreads();
        if (wave == 1) { v2_helper_wave(src, work, &zs, flags, lane); return; }
    }
    uint32_t* const fl = two_wave ? flags : nullptr;
    auto release_helper = [&] {   // whatever made the first wave leave before its go: the helper wave has nothing to do
        if (two_wave && lane == 0 && __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
            __hip_atomic_store(&flags[0], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int r;
    if (container_mode && blk.src_len == blk.dst_cap) { release_helper(); wave_copy(dst, src, int(blk.src_len), lane); r = int(blk.src_len); }
    else {
        r = zstd_decode_frame_v2(src, int(blk.src_len), dst, int(blk.dst_cap), work, &zs, lane, split, fl);
        release_helper();
        // A helper wave that was given its go reads the block table and the tables of the shared state until it reports: the serial
        // path below rebuilds those tables, so it must not start (after a decline on damaged input, say) before the helper is done
        if (two_wave && r == kDecline && __hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 1u) {
            bool reported = false;
            for (uint32_t spins = 0; spins < (1u << 24); spins++) {
                if (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) { reported = true; break; }
                __builtin_amdgcn_s_sleep(16);
            }
            // a helper that never reports is still reading the tables the serial path would rebuild: the block fails (a generic
            // zstd error; corrupt in a container) rather than being decoded under it
            if (!reported) r = -1;
        }
        if (r == kDecline) r = zstd_decode_frames(src, int(blk.src_len), dst, int(blk.dst_cap), litbuf, &zs, lane);
        if (container_mode && r < 0 && r != kPendingExec) r = FOURMC_BLK_CORRUPT;
    }
    if (lane == 0) blocks[b].result = r;
}

#include "zstd_exec.inc"

} // namespace

extern "C" size_t fourmc_zstd_scratch_bytes(uint32_t n) { return size_t(n) * kV2Bytes; }
extern "C" size_t fourmc_zstd_dec_counter_offset(void) { return kV2Bytes - 64; }   // phase counters of block 0 (profiling aid)

// test aid: blocks the execute kernel completed / handed back to the one-wave kernel since the last call (this device)
#ifdef FOURMC_RESEARCH
extern "C" int fourmc_gpu_debug_zstd_exec_counts(unsigned long long* executed, unsigned long long* handed_back)
{
    unsigned long long c[2] = {0, 0}, z[2] = {0, 0};
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_x_counts), sizeof c) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_x_counts), z, sizeof z) != hipSuccess) return -1;
    if (executed) *executed = c[0];
    if (handed_back) *handed_back = c[1];
    return 0;
}
#endif

// FOURMC_ZDECODE = split (default: entropy kernel + execute kernel + hand-backs) | single (everything in the one-wave kernel)
static int g_zdecode_split = -1;
extern "C" void fourmc_gpu_set_zstd_decode_split(int on) { g_zdecode_split = on ? 1 : 0; }
extern "C" int fourmc_gpu_get_zstd_decode_split(void)
{
    if (g_zdecode_split < 0) { const char* e = getenv("FOURMC_ZDECODE"); g_zdecode_split = (e && !strcmp(e, "single")) ? 0 : 1; }
    return g_zdecode_split;
}

extern "C" hipError_t fourmc_launch_zstd_decode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                                void* d_scratch, int container_mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const int split = fourmc_gpu_get_zstd_decode_split();
    static const int helper = [] { const char* e = getenv("FOURMC_ZHELPER"); return e ? atoi(e) : -1; }();
    // the helper wave pays while the frames' chains are what a launch waits for; a full chip is bound by the table reads instead
    const bool two_wave = split && (helper < 0 ? n <= 256u : helper != 0);
    hipLaunchKernelGGL(zstd_decode_kernel, dim3(n), dim3(two_wave ? 128 : 64), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n,
                       static_cast<uint8_t*>(d_scratch), container_mode, split ? 1 : 0);
    if (split) {
        hipLaunchKernelGGL(zstd_exec_kernel, dim3(n), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n,
                           static_cast<uint8_t*>(d_scratch), container_mode);
        hipLaunchKernelGGL(zstd_decode_kernel, dim3(n), dim3(64), 0, stream,
                           static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n,
                           static_cast<uint8_t*>(d_scratch), container_mode, 2);
    }
    return hipGetLastError();
}
