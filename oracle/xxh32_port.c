/*
 * oracle/xxh32_port.c — scalar restatement of XXH32.  TEST INFRASTRUCTURE, NOT PRODUCT.
 * Follows native/lz4/xxhash.c:392-415 (entry), :352-389 (stripe loop), :276-281 (round),
 * :291-345 (tail), :283-290 (avalanche); primes :263-267.  Parity: pinned against
 * oracle/_ref (XXH32) and the header/footer checksums in tests/golden/.
 */
#include "oracle.h"

#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4  668265263u
#define P5  374761393u

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint32_t lane_round(uint32_t acc, uint32_t w) { return rotl32(acc + w * P2, 13) * P1; }

uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* const end = p + len;
    uint32_t h;

    if (len >= 16) {
        uint32_t a = seed + P1 + P2, b = seed + P2, c = seed, d = seed - P1;
        const uint8_t* const stripes_end = end - 16;
        do {
            a = lane_round(a, le32(p));
            b = lane_round(b, le32(p + 4));
            c = lane_round(c, le32(p + 8));
            d = lane_round(d, le32(p + 12));
            p += 16;
        } while (p <= stripes_end);
        h = rotl32(a, 1) + rotl32(b, 7) + rotl32(c, 12) + rotl32(d, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + le32(p) * P3, 17) * P4; p += 4; }
    while (p < end)      { h = rotl32(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2;
    h ^= h >> 13; h *= P3;
    h ^= h >> 16;
    return h;
}
