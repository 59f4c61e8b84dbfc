/*
 * 4mc_amd/csrc/shard.c - one .4mc / .4mz file written by several ranks (one process per GPU).
 *
 * The reference writes a file with ONE loop (native/4mc.c:280-333): block header, payload, next block; the footer index
 * (native/4mc.c:344-358) is the list of the file offsets the loop passed through.  Blocks are independent, so here rank r
 * of `world` compresses the contiguous block range [first, first + count) on its own GPU, and the only thing the ranks
 * have to tell each other is how many bytes each block became: ONE all-gather of 4 bytes per block.  After it every rank
 * holds the same exclusive prefix sum - the footer index - and writes its own byte range with pwrite(); rank 0 adds the file
 * header (native/4mc.c:264-268), the end mark (:336-340) and the footer.  The file is byte-identical to the serial one.
 *
 * The collective is injected (fourmc_allgather_fn): torch.distributed over RCCL / gloo from a Python launcher, ncclAllGather
 * or MPI_Allgather from a C one.  libhadoop-4mc.so itself links no communication library - it keeps the dependency set of
 * the reference's library (libc, libstdc++, HIP).
 */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include "fourmc.h"
#include "fourmc_gpu.h"

void fourmc_shard_range(uint64_t nblocks, int rank, int world, uint64_t* first, uint64_t* count)
{
    const uint64_t per = world > 0 ? (nblocks + (uint64_t)world - 1) / (uint64_t)world : nblocks;
    uint64_t lo = (uint64_t)rank * per, hi;
    if (lo > nblocks) lo = nblocks;
    hi = lo + per; if (hi > nblocks) hi = nblocks;
    *first = lo; *count = hi - lo;
}

/* off_all[b] = file offset of block b's 12-byte header = 12 + sum_{j<b} (12 + csize_j)   (native/4mc.c:293) */
void fourmc_shard_offsets(const uint32_t* csize_all, uint64_t nblocks, uint64_t* off_all)
{
    uint64_t pos = 12, b;
    for (b = 0; b < nblocks; b++) { off_all[b] = pos; pos += 12ull + csize_all[b]; }
}

static int pwrite_all(int fd, const void* buf, size_t n, uint64_t at)
{
    const uint8_t* p = (const uint8_t*)buf;
    while (n) {
        ssize_t w = pwrite(fd, p, n, (off_t)at);
        if (w <= 0) return -1;
        p += w; n -= (size_t)w; at += (uint64_t)w;
    }
    return 0;
}

/* usize / csize / xxh32 / payload_off: this rank's `count` blocks; payload b = payloads + payload_off[b].
 * off_all: all nblocks offsets.  Returns 0, or -1 on a write error. */
int fourmc_shard_write(int fd, uint32_t magic, int rank, uint64_t first, uint64_t count, uint64_t nblocks, const uint64_t* off_all,
                       const uint32_t* csize_all, const uint32_t* usize, const uint32_t* xxh32, const uint8_t* payloads, const uint64_t* payload_off)
{
    uint8_t hdr[12];
    uint64_t b;
    if (rank == 0) {
        /* The reference opens its output with "wb" (native/4mc.c:200): an older, longer file of that name must not leave bytes
         * behind the footer - readers find the footer from the END of the file.  After the gather every rank knows the final
         * size; rank 0 sets it.  Cutting to the FINAL size cannot hurt what the other ranks have written already. */
        const uint64_t final_size = (nblocks ? off_all[nblocks - 1] + 12 + csize_all[nblocks - 1] : 12) + 12 + (20 + 4 * nblocks);
        if (ftruncate(fd, (off_t)final_size) != 0) return -1;
        fourmc_frame_header(hdr, magic);
        if (pwrite_all(fd, hdr, 12, 0)) return -1;
    }
    for (b = 0; b < count; b++) {
        const uint32_t cs = csize_all[first + b];
        fourmc_frame_block_header(hdr, usize[b], cs, xxh32[b]);
        if (pwrite_all(fd, hdr, 12, off_all[first + b])) return -1;
        if (pwrite_all(fd, payloads + payload_off[b], cs, off_all[first + b] + 12)) return -1;
    }
    if (rank == 0) {
        const uint64_t end = nblocks ? off_all[nblocks - 1] + 12ull + csize_all[nblocks - 1] : 12;
        const size_t fsz = FOURMC_FOOTERSIZE(nblocks);
        uint8_t* foot = (uint8_t*)malloc(fsz);
        if (!foot) return -1;
        memset(hdr, 0, 12);
        fourmc_frame_footer(foot, magic, off_all, (uint32_t)nblocks);
        if (pwrite_all(fd, hdr, 12, end) || pwrite_all(fd, foot, fsz, end + 12)) { free(foot); return -1; }
        free(foot);
    }
    return 0;
}

/* Every rank calls this with the same arguments and its own rank, after selecting its device (fourmc_gpu_init).
 * level / magic as fourM{C,Z}compressFilename.  Returns 0; -1 input, -2 output, -3 engine, -4 collective, -5 memory,
 * -6 another rank failed (this one had nothing to report).
 * ONE RANK'S FAILURE ENDS THE CALL ON EVERY RANK: the row a rank contributes to the all-gather is {status, sizes...}, a rank that
 * failed before the exchange still takes part in it (with its status and zeros), and after it every rank that sees a status other
 * than 0 leaves without writing - with its own code, or -6 when the failure was a peer's.  (Round 5 skipped the collective on the
 * failing rank: the others then waited in ncclAllGather forever.)  What cannot be reported this way: a rank that cannot stat the
 * input or cannot allocate its two gather rows returns at once, like a rank that was never started.
 * FOURMC_SHARD_FAIL_RANK=r (test aid): rank r reports an engine failure without encoding anything. */
int fourmc_file_compress_sharded(const char* in_name, const char* out_name, int level, uint32_t magic, int rank, int world,
                                 fourmc_allgather_fn allgather, void* ctx)
{
    struct stat st;
    uint64_t nblocks, first, count, b, per;
    uint8_t *in_buf = NULL, *out_buf = NULL, *store = NULL;
    fourmc_block* blk = NULL;
    uint32_t *cs_mine = NULL, *cs_pad = NULL, *cs_all = NULL, *usz = NULL, *xs = NULL;
    uint64_t *off_all = NULL, *poff = NULL;
    int codec, codec_level = 0, rc = 0, fd = -1;

    if (magic == FOURMC_MAGIC_4MC) {                                  /* native/4mc.c:243-253   */
        if (level <= 1) codec = FOURMC_CODEC_LZ4_FAST;
        else if (level == 2) codec = FOURMC_CODEC_LZ4_MC;
        else { codec = FOURMC_CODEC_LZ4_HC; codec_level = (level == 3) ? 4 : 8; }
    } else {                                                          /* native/4mc.c:411-419   */
        codec = FOURMC_CODEC_ZSTD;
        codec_level = level <= 1 ? 1 : level == 2 ? 3 : level == 3 ? 6 : 12;
    }
    if (stat(in_name, &st) != 0) return -1;
    nblocks = ((uint64_t)st.st_size + FOURMC_BLOCKSIZE - 1) / FOURMC_BLOCKSIZE;
    fourmc_shard_range(nblocks, rank, world, &first, &count);
    per = world > 0 ? (nblocks + (uint64_t)world - 1) / (uint64_t)world : nblocks;

    /* The rank's range goes through the engine in batches of FOURMC_BATCH_BLOCKS blocks (default 512 = 2 GiB: the kernels need
     * hundreds of blocks per launch), so that the input is never in memory as a whole (64 GiB over 8 ranks would be 8 GiB of
     * pageable memory per rank); what has to stay until the gather is the COMPRESSED range, kept back to back in `store`. */
    {
        const char* e = getenv("FOURMC_BATCH_BLOCKS");
        uint64_t nbatch = e ? (uint64_t)atol(e) : 512, store_cap = 0, store_len = 0, b0;
        int fdin;
        if (nbatch < 1) nbatch = 1;
        if (nbatch > 4096) nbatch = 4096;
        if (nbatch > count) nbatch = count ? count : 1;
        in_buf = (uint8_t*)malloc(nbatch * FOURMC_BLOCKSIZE); out_buf = (uint8_t*)malloc(nbatch * FOURMC_BLOCKSIZE);
        blk = (fourmc_block*)calloc(nbatch + 1, sizeof *blk);
        /* the gather rows first: a rank that has them can always take part in the exchange */
        cs_pad = (uint32_t*)calloc(per + 2, 4); cs_all = (uint32_t*)calloc((per + 1) * (uint64_t)(world > 0 ? world : 1) + 1, 4);
        if (!cs_pad || !cs_all) { rc = -5; goto done; }
        usz = (uint32_t*)calloc(count + 1, 4); xs = (uint32_t*)calloc(count + 1, 4);
        off_all = (uint64_t*)calloc(nblocks + 1, 8); poff = (uint64_t*)calloc(count + 1, 8);
        cs_mine = cs_pad + 1;                                      /* word 0 of the row: this rank's status */
        if (!in_buf || !out_buf || !blk || !usz || !xs || !off_all || !poff) { rc = -5; goto gather; }
        { const char* fr = getenv("FOURMC_SHARD_FAIL_RANK"); if (fr && atoi(fr) == rank) { rc = -3; goto gather; } }
        fdin = open(in_name, O_RDONLY);
        if (fdin < 0) { rc = -1; goto gather; }
        for (b0 = 0; b0 < count; b0 += nbatch) {
            const uint64_t nb = count - b0 < nbatch ? count - b0 : nbatch;
            const uint64_t at = (first + b0) * FOURMC_BLOCKSIZE;
            const uint64_t bytes = ((uint64_t)st.st_size - at < nb * FOURMC_BLOCKSIZE) ? (uint64_t)st.st_size - at : nb * FOURMC_BLOCKSIZE;
            uint64_t got = 0, need = 0;
            while (got < bytes) { ssize_t r = pread(fdin, in_buf + got, (size_t)(bytes - got), (off_t)(at + got)); if (r <= 0) break; got += (uint64_t)r; }
            if (got != bytes) { close(fdin); rc = -1; goto gather; }
            for (b = 0; b < nb; b++) {
                const uint64_t left = bytes - b * FOURMC_BLOCKSIZE;
                blk[b].src_off = b * FOURMC_BLOCKSIZE; blk[b].dst_off = b * FOURMC_BLOCKSIZE;
                blk[b].src_len = (uint32_t)(left < FOURMC_BLOCKSIZE ? left : FOURMC_BLOCKSIZE);
                blk[b].dst_cap = blk[b].src_len; blk[b].result = 0; blk[b].xxh32 = 0;
            }
            if (fourmc_host_4mc_encode(in_buf, (size_t)bytes, out_buf, (size_t)(nb * FOURMC_BLOCKSIZE), blk, (uint32_t)nb, codec, codec_level) != FOURMC_OK) { close(fdin); rc = -3; goto gather; }
            for (b = 0; b < nb; b++) {
                /* a per-block failure code must not become a 4 GiB size that every rank then builds its offsets on */
                if (blk[b].result <= 0 || (uint32_t)blk[b].result > blk[b].src_len) { close(fdin); rc = -3; goto gather; }
                need += (uint32_t)blk[b].result;
            }
            if (store_len + need > store_cap) {
                uint8_t* ns;
                store_cap = (store_len + need) + (store_len + need) / 4 + (1u << 20);
                ns = (uint8_t*)realloc(store, (size_t)store_cap);
                if (!ns) { close(fdin); rc = -5; goto gather; }
                store = ns;
            }
            for (b = 0; b < nb; b++) {
                const uint32_t cs = (uint32_t)blk[b].result;
                memcpy(store + store_len, out_buf + blk[b].dst_off, cs);
                cs_mine[b0 + b] = cs; usz[b0 + b] = blk[b].src_len; xs[b0 + b] = blk[b].xxh32; poff[b0 + b] = store_len;
                store_len += cs;
            }
        }
        close(fdin);
    }

gather:
    /* the one exchange of the path: {status, per-block compressed sizes padded to equal counts per rank} */
    cs_pad[0] = (uint32_t)rc;
    if (rc) memset(cs_pad + 1, 0, per * 4);
    if (allgather) { if (allgather(ctx, cs_pad, (per + 1) * 4, cs_all) != 0) { if (!rc) rc = -4; goto done; } }   /* also with one rank: the launcher's collective is the real one */
    else if (world <= 1) memcpy(cs_all, cs_pad, (per + 1) * 4);
    else { if (!rc) rc = -4; goto done; }
    {   /* every rank sees every status: all leave together (before anything is written) when any of them failed */
        const uint64_t w = (uint64_t)(world > 0 ? world : 1);
        uint64_t r;
        int peer_failed = 0;
        for (r = 0; r < w; r++) if (cs_all[r * (per + 1)] != 0) peer_failed = 1;
        if (rc) goto done;
        if (peer_failed) { rc = -6; goto done; }
        /* ranks' rows without their status words -> one array in block order (row r holds blocks [r * per, ..)) */
        for (r = 0; r < w; r++) memmove(cs_all + r * per, cs_all + r * (per + 1) + 1, (size_t)per * 4);
        fourmc_shard_offsets(cs_all, nblocks, off_all);
    }
    fd = open(out_name, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) { rc = -2; goto done; }
    if (fourmc_shard_write(fd, magic, rank, first, count, nblocks, off_all, cs_all, usz, xs, store ? store : out_buf, poff) != 0) rc = -2;
    if (close(fd) != 0 && rc == 0) rc = -2;
done:
    free(in_buf); free(out_buf); free(store); free(blk); free(cs_pad); free(cs_all); free(usz); free(xs); free(off_all); free(poff);
    return rc;
}

/* ---- the other direction: one .4mc / .4mz file decompressed by several ranks ------------------------------------------------
 * Nothing has to be exchanged: the footer index (native/4mc.c:344-358, FourMcBlockIndex.java:92-173) tells every rank where its
 * blocks lie, and every block but the last decodes to FOURMC_BLOCKSIZE bytes, so block b goes to offset b * FOURMC_BLOCKSIZE of
 * the output.  Rank r decodes the block range fourmc_shard_range() gives it, in batches (fourmc_file_decode_blocks: one read of
 * the byte range, one launch, checksums verified on the device), and pwrite()s them; the rank that owns the last block sets the
 * file's final size (an older, longer file of that name must not leave bytes behind).
 * Returns 0; -1 input, -2 output, -3 engine / corrupt block (the code of fourmc_file_decode_blocks is in *detail), -5 memory. */
int fourmc_file_decompress_sharded(const char* in_name, const char* out_name, int rank, int world, long long* detail)
{
    int is_zstd = 0, fd = -1, rc = 0;
    const int64_t nblocks = fourmc_file_block_count(in_name, &is_zstd);
    uint64_t first, count, b0, nbatch;
    uint8_t* buf = NULL;
    const char* e = getenv("FOURMC_BATCH_BLOCKS");
    if (detail) *detail = 0;
    if (nblocks < 0) { if (detail) *detail = nblocks; return -1; }
    fourmc_shard_range((uint64_t)nblocks, rank, world, &first, &count);
    nbatch = e ? (uint64_t)atol(e) : 512;
    if (nbatch < 1) nbatch = 1;
    if (nbatch > 4096) nbatch = 4096;
    if (nbatch > count) nbatch = count ? count : 1;
    fd = open(out_name, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) return -2;
    if (nblocks == 0) { if (rank == 0 && ftruncate(fd, 0) != 0) rc = -2; close(fd); return rc; }
    buf = (uint8_t*)malloc((size_t)(nbatch * FOURMC_BLOCKSIZE));
    if (!buf) { close(fd); return -5; }
    for (b0 = 0; b0 < count && rc == 0; b0 += nbatch) {
        const uint64_t nb = count - b0 < nbatch ? count - b0 : nbatch;
        const int64_t got = fourmc_file_decode_blocks(in_name, (uint32_t)(first + b0), (uint32_t)nb, buf, (size_t)(nb * FOURMC_BLOCKSIZE));
        const int last_batch = first + b0 + nb == (uint64_t)nblocks;
        if (got < 0) { if (detail) *detail = got; rc = -3; break; }
        /* every block in front of the file's last one is a full one: anything else cannot be placed by its index alone */
        if (last_batch ? ((uint64_t)got <= (nb - 1) * FOURMC_BLOCKSIZE || (uint64_t)got > nb * FOURMC_BLOCKSIZE) : (uint64_t)got != nb * FOURMC_BLOCKSIZE) { if (detail) *detail = got; rc = -3; break; }
        if (pwrite_all(fd, buf, (size_t)got, (first + b0) * FOURMC_BLOCKSIZE)) { rc = -2; break; }
        if (last_batch && ftruncate(fd, (off_t)((first + b0) * FOURMC_BLOCKSIZE + (uint64_t)got)) != 0) rc = -2;
    }
    free(buf);
    if (close(fd) != 0 && rc == 0) rc = -2;
    return rc;
}
