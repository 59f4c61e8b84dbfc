/*
 * 4mc_amd/csrc/framing.c — byte-exact 4mc/4mz container framing + footer-index queries (host C).
 *
 * Restates the framing bytes produced/consumed by the reference:
 *   header        native/4mc.c:264-268 (writer)  :575-585 (reader)
 *   block header  native/4mc.c:309-312           :609-614
 *   end mark      native/4mc.c:336-340           :616
 *   footer        native/4mc.c:344-358           :670-688
 *   index queries java/hadoop-4mc/src/main/java/com/fing/compression/fourmc/FourMcBlockIndex.java:92-173
 * The only arithmetic here is big-endian packing and XXH32 over a handful of framing bytes;
 * payload checksums are computed on the GPU (xxh32.hip).
 */
#include <string.h>
#include "fourmc.h"
#include "fourmc_gpu.h"

/* ---- XXH32 (host scalar).  Algorithm: native/lz4/xxhash.c:392-415; primes :263-267. -------- */
static uint32_t rol(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t get_le32(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

unsigned fourmc_XXH32(const void* input, size_t len, unsigned seed)
{
    static const uint32_t A = 2654435761u, B = 2246822519u, C = 3266489917u, D = 668265263u, E = 374761393u;
    const uint8_t* p = (const uint8_t*)input;
    size_t left = len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v[4] = { seed + A + B, seed + B, seed, seed - A };
        for (; left >= 16; left -= 16, p += 16) {
            int i;
            for (i = 0; i < 4; i++) v[i] = rol(v[i] + get_le32(p + 4 * i) * B, 13) * A;
        }
        h = rol(v[0], 1) + rol(v[1], 7) + rol(v[2], 12) + rol(v[3], 18);
    } else h = seed + E;
    h += (uint32_t)len;
    for (; left >= 4; left -= 4, p += 4) h = rol(h + get_le32(p) * C, 17) * D;
    for (; left; left--, p++) h = rol(h + *p * E, 11) * A;
    h ^= h >> 15; h *= B; h ^= h >> 13; h *= C; h ^= h >> 16;
    return h;
}

/* ---- big-endian fields ---------------------------------------------------------------------- */
static void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static uint32_t get_be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void fourmc_frame_header(uint8_t out[12], uint32_t magic)
{
    put_be32(out, magic);
    put_be32(out + 4, 1);                                   /* FOURMC_VERSION */
    put_be32(out + 8, fourmc_XXH32(out, 8, 0));
}

int fourmc_frame_check_header(const uint8_t in[12], uint32_t magic)
{
    if (get_be32(in) != magic) return 1;
    if (get_be32(in + 4) != 1) return 2;
    if (get_be32(in + 8) != fourmc_XXH32(in, 8, 0)) return 3;
    return 0;
}

void fourmc_frame_block_header(uint8_t out[12], uint32_t usize, uint32_t csize, uint32_t xxh32)
{
    put_be32(out, usize); put_be32(out + 4, csize); put_be32(out + 8, xxh32);
}

void fourmc_frame_parse_block_header(const uint8_t in[12], uint32_t* usize, uint32_t* csize, uint32_t* xxh32)
{
    *usize = get_be32(in); *csize = get_be32(in + 4); *xxh32 = get_be32(in + 8);
}

size_t fourmc_frame_footer(uint8_t* out, uint32_t magic, const uint64_t* off, uint32_t n)
{
    const uint32_t size = FOURMC_FOOTERSIZE(n);
    uint32_t i;
    put_be32(out, size);
    put_be32(out + 4, 1);                                   /* footer version */
    for (i = 0; i < n; i++)                                 /* delta to the previous block; first = absolute */
        put_be32(out + 8 + 4 * i, (uint32_t)(i ? off[i] - off[i - 1] : off[0]));
    put_be32(out + 8 + 4 * n, size);
    put_be32(out + 12 + 4 * n, magic);
    put_be32(out + 16 + 4 * n, fourmc_XXH32(out, size - 4, 0));
    return size;
}

int64_t fourmc_frame_parse_footer(const uint8_t* foot, size_t len, uint32_t magic, uint64_t* off)
{
    uint32_t size, n, i;
    uint64_t abs = 0;
    if (len < 20) return -1;
    size = get_be32(foot);
    if (size < 20 || size > len || ((size - 20) & 3)) return -1;
    if (get_be32(foot + size - 4) != fourmc_XXH32(foot, size - 4, 0)) return -2;
    if (get_be32(foot + 4) != 1) return -3;
    if (get_be32(foot + size - 12) != size || get_be32(foot + size - 8) != magic) return -4;
    n = (size - 20) / 4;
    for (i = 0; i < n; i++) { abs += get_be32(foot + 8 + 4 * i); if (off) off[i] = abs; }
    return n;
}

/* ---- index queries ---------------------------------------------------------------------------- */
/* smallest i with off[i] >= pos, or n */
static uint32_t lower_bound(const uint64_t* off, uint32_t n, uint64_t pos)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (off[mid] < pos) lo = mid + 1; else hi = mid; }
    return lo;
}

int64_t fourmc_index_find_next(const uint64_t* off, uint32_t n, uint64_t pos)
{
    uint32_t i = lower_bound(off, n, pos);
    return i < n ? (int64_t)off[i] : -1;
}

int64_t fourmc_index_find_block(const uint64_t* off, uint32_t n, uint64_t pos)
{
    uint32_t i = lower_bound(off, n, pos);
    if (i < n && off[i] == pos) return i;
    return i == 0 ? -1 : (int64_t)i - 1;
}

uint64_t fourmc_index_align_start(const uint64_t* off, uint32_t n, uint64_t start, uint64_t end)
{
    int64_t p;
    if (start == 0) return 0;
    p = fourmc_index_find_next(off, n, start);
    if (p < 0 || (uint64_t)p >= end) return (uint64_t)-1;
    return (uint64_t)p;
}

uint64_t fourmc_index_align_end(const uint64_t* off, uint32_t n, uint64_t end, uint64_t file_size)
{
    int64_t p = fourmc_index_find_next(off, n, end);
    return p >= 0 ? (uint64_t)p : file_size;
}
