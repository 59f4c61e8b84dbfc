/*
 * oracle/container_port.c — in-memory restatement of the 4mc/4mz container writer and reader.
 * TEST INFRASTRUCTURE, NOT PRODUCT (see oracle.h).
 *
 *   writer : fourMCcompressFilename / fourMZcompressFilename  native/4mc.c:220-386, :389-553
 *            header :264-274, per-block loop :280-333, end mark :336-341, footer :344-362
 *   reader : decodeFourMC / decodeFourMZ                        native/4mc.c:560-707, :709-857
 *   spec   : 4mc-format-spec:1-36, 4mz-format-spec:1-35 (all fields big-endian u32)
 *
 * Parity: pinned — whole-file bytes compared with oracle/_ref/4mc_ref (the reference CLI built
 * from its own sources) and with the golden files in tests/golden/.
 */
#include <string.h>
#include "oracle.h"

static void be32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static uint32_t rbe32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

size_t orc_container_bound(size_t n)
{
    size_t nb = (n + ORC_BLOCKSIZE - 1) / ORC_BLOCKSIZE;
    return 12 + n + 12 * nb + 12 + 20 + 4 * nb;
}

int64_t orc_container_compress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                               uint32_t magic, orc_block_codec_fn compress, void* ctx)
{
    size_t nb = (n + ORC_BLOCKSIZE - 1) / ORC_BLOCKSIZE, b, pos = 0, prev = 0, fsz;
    uint8_t* foot;
    if (cap < orc_container_bound(n)) return -1;

    be32(dst, magic); be32(dst + 4, 1); be32(dst + 8, orc_xxh32(dst, 8, 0));        /* :264-268 */
    pos = 12;
    /* the footer is assembled at the tail of dst while blocks are laid down */
    fsz = 20 + 4 * nb;
    foot = dst + cap - fsz;
    for (b = 0; b < nb; b++) {
        const uint8_t* in = src + b * (size_t)ORC_BLOCKSIZE;
        uint32_t usz = (uint32_t)((n - b * (size_t)ORC_BLOCKSIZE < ORC_BLOCKSIZE) ? n - b * (size_t)ORC_BLOCKSIZE : ORC_BLOCKSIZE);
        int csz = compress(ctx, in, (int)usz, dst + pos + 12, (int)usz - 1);        /* :301     */
        be32(foot + 8 + 4 * b, (uint32_t)(pos - prev));                               /* :349-352 */
        prev = pos;
        be32(dst + pos, usz);
        if (csz > 0) {
            be32(dst + pos + 4, (uint32_t)csz);
            be32(dst + pos + 8, orc_xxh32(dst + pos + 12, (size_t)csz, 0));           /* :311     */
            pos += 12 + (size_t)csz;
        } else {                                                                      /* :318-329 */
            be32(dst + pos + 4, usz);
            be32(dst + pos + 8, orc_xxh32(in, usz, 0));
            memcpy(dst + pos + 12, in, usz);
            pos += 12 + usz;
        }
    }
    memset(dst + pos, 0, 12); pos += 12;                                              /* :336-340 */
    be32(foot, (uint32_t)fsz); be32(foot + 4, 1);
    be32(foot + 8 + 4 * nb, (uint32_t)fsz); be32(foot + 12 + 4 * nb, magic);
    be32(foot + 16 + 4 * nb, orc_xxh32(foot, fsz - 4, 0));                            /* :357     */
    memmove(dst + pos, foot, fsz); pos += fsz;
    return (int64_t)pos;
}

int64_t orc_container_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                                 uint32_t magic, orc_block_codec_fn decompress, void* ctx,
                                 size_t* consumed)
{
    size_t pos = 0, out = 0;
    uint32_t fsz;
    if (n < 4) return -4;                                   /* "Magic Number unreadable" :870     */
    if (rbe32(src) != magic) return -4;                     /* "not a 4mc file"          :873     */
    if (n < 12) return -4;                                  /* "Unreadable header"       :579     */
    if (rbe32(src + 4) != 1) return -4;                     /* "Wrong version number"    :583     */
    if (rbe32(src + 8) != orc_xxh32(src, 8, 0)) return -4;  /* "Wrong header checksum"   :584     */
    pos = 12;
    for (;;) {
        uint32_t usz, csz, sum;
        if (n - pos < 12) return -2;                        /* "cannot read next block size" :610 */
        usz = rbe32(src + pos); csz = rbe32(src + pos + 4); sum = rbe32(src + pos + 8);
        pos += 12;
        if (usz == 0 && csz == 0 && sum == 0) break;        /* end mark                  :616     */
        if (csz > ORC_BLOCKSIZE) return -4;                 /* "beyond 4MB limit"        :618     */
        if (n - pos < csz) return -2;                       /* "cannot read data block"  :632     */
        if (orc_xxh32(src + pos, csz, 0) != sum) return -4; /* "invalid block checksum"  :637,645 */
        if (usz == csz) {                                   /* stored                    :635-642 */
            if (cap - out < usz) return -3;
            memcpy(dst + out, src + pos, usz); out += usz;
        } else {
            int r;
            if (usz > ORC_BLOCKSIZE) return -4;             /* :651                               */
            if (cap - out < usz) return -3;
            r = decompress(ctx, src + pos, (int)csz, dst + out, (int)usz);            /* :661     */
            if (r < 0) return -4;                           /* "Corrupted input detected" :662    */
            out += (size_t)r;
        }
        pos += csz;
    }
    if (n - pos < 4) return -1;                             /* "Unreadable footer" (exit 1) :672   */
    fsz = rbe32(src + pos);
    if (fsz < 20 || n - pos < fsz) return -2;               /* "cannot read footer"      :680     */
    if (orc_xxh32(src + pos, fsz - 4, 0) != rbe32(src + pos + fsz - 4)) return -4;    /* :684-685 */
    if (rbe32(src + pos + 4) != 1) return -4;               /* footer version            :687     */
    pos += fsz;
    if (consumed) *consumed = pos;
    return (int64_t)out;
}
