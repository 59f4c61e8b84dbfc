/*
 * 4mc_amd/csrc/fourmc_file.c — the reference's library API (native/4mc.h:36-41) on the GPU engine.
 *
 * Same signatures, stderr text, display levels and exit codes as native/4mc.c:220-386 /:389-553
 * (compress) and :560-964 (decompress); the difference is the shape of the hot loop: instead of
 * one codec call + one XXH32 per 4 MiB block (native/4mc.c:301,311 / :637,661), a batch of
 * independent blocks is handed to ONE launch sequence of the block engine (fourmc_gpu.h), and
 * framing is laid around the results.  Output files are byte-identical to the reference's.
 *
 * No CPU codec exists in this build: if the engine cannot run (no gfx950 device, unsupported
 * level) the call ends with exit code 1 and a message, it never silently degrades.
 *
 * Deliberate deviation: the reference compares the output name with `nulmark` by POINTER
 * (native/4mc.c:190), which never matches across translation units, so `4mc -t` asks whether
 * /dev/null may be overwritten; here the name is compared as a string and /dev/null is never
 * treated as an existing file.
 */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/stat.h>
#include <pthread.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include "fourmc.h"
#include "fourmc_gpu.h"

#define BLOCKSIZE FOURMC_BLOCKSIZE

#define PRINT(...)          fprintf(stderr, __VA_ARGS__)
#define PRINT_LEVEL(l, ...) do { if (displayLevel >= (l)) PRINT(__VA_ARGS__); } while (0)
#define DIE(code, ...)      do { PRINT_LEVEL(1, __VA_ARGS__); PRINT_LEVEL(1, "\n"); exit(code); } while (0)

static unsigned batch_blocks(void)
{
    const char* e = getenv("FOURMC_BATCH_BLOCKS");
    long v = e ? atol(e) : 64;
    if (v < 1) v = 1;
    if (v > 4096) v = 4096;
    return (unsigned)v;
}

/* Staging buffers of the streaming path: plain memory.  On the bench host a copy from or into pageable memory runs at
 * 50 GB/s (tools/ubench/host_io.cpp) while page-locking the four 256 MiB buffers costs 0.35 s of every process
 * (tools/startup_timing.py: `4mc -d` of one block 0.64 s with, 0.3 s without); FOURMC_PINNED=1 brings them back. */
typedef struct { void* p; int pinned; } hbuf;
static hbuf hbuf_alloc(size_t n)
{
    const char* e = getenv("FOURMC_PINNED");
    hbuf b; b.p = NULL; b.pinned = 0;
    if (e && !strcmp(e, "1")) { b.p = fourmc_host_alloc(n); b.pinned = b.p != NULL; }
    if (!b.p) b.p = malloc(n);
    return b;
}
static void hbuf_free(hbuf b) { if (b.pinned) fourmc_host_free(b.p); else free(b.p); }

/* One batch in flight on the engine while the caller reads the next one / writes the previous one: the engine call runs
 * on a helper thread, the two stdio streams stay on the calling thread (two buffer sets, strict alternation). */
typedef struct {
    pthread_t th; int running;
    int encode; const void* src; size_t src_bytes; void* dst; size_t dst_bytes; fourmc_block* blk; uint32_t n; int codec, level;
    int rc; char err[256];
} job_t;
static void* job_main(void* arg)
{
    job_t* j = (job_t*)arg;
    j->rc = j->encode ? fourmc_host_4mc_encode(j->src, j->src_bytes, j->dst, j->dst_bytes, j->blk, j->n, j->codec, j->level)
                      : fourmc_host_4mc_decode(j->src, j->src_bytes, j->dst, j->dst_bytes, j->blk, j->n, j->codec);
    if (j->rc != FOURMC_OK) snprintf(j->err, sizeof j->err, "%s", fourmc_gpu_last_error());   /* last_error is per thread */
    return NULL;
}
static void job_start(job_t* j)
{
    j->rc = FOURMC_OK; j->err[0] = 0;
    if (pthread_create(&j->th, NULL, job_main, j) == 0) j->running = 1;
    else { j->running = 0; job_main(j); }                 /* no thread: run it here, nothing overlaps */
}
static int job_wait(job_t* j) { if (j->running) { pthread_join(j->th, NULL); j->running = 0; } return j->rc; }

/* The output side of a batch on a thread of its own: the calling thread is already reading the next batch while the one
 * before this is on the GPU.  Compress: 12-byte block header + payload per block, offsets recorded for the footer;
 * decompress: the decoded blocks.  An error is kept (code, message) and raised by the caller once nothing else runs. */
typedef struct {
    pthread_t th; int running;
    int compress; FILE* fout; const uint8_t* out_buf; fourmc_block* blk; uint32_t n; int displayLevel;   /* blk: the writer's own copy (the caller reuses its array for the batch it is reading) */
    unsigned long long *filesize, *outsize; uint64_t** offsets; size_t *noff, *capoff;
    int code; const char* msg;
} wjob_t;
static void* wjob_main(void* arg)
{
    wjob_t* w = (wjob_t*)arg;
    const int displayLevel = w->displayLevel;
    uint32_t b; uint8_t hdr[12];
    for (b = 0; b < w->n && !w->msg; b++) {
        const fourmc_block* pb = w->blk + b;
        if (w->compress) {
            const uint32_t usize = pb->src_len, csize = (uint32_t)pb->result;
            if (*w->noff == *w->capoff) {
                *w->capoff = *w->capoff ? *w->capoff * 2 : 1024;
                *w->offsets = (uint64_t*)realloc(*w->offsets, *w->capoff * sizeof **w->offsets);
                if (!*w->offsets) { w->code = 1; w->msg = "Allocation error : not enough memory"; break; }
            }
            (*w->offsets)[(*w->noff)++] = *w->outsize;
            *w->filesize += usize;
            PRINT_LEVEL(3, "\rRead : %i MB   ", (int)(*w->filesize >> 20));
            fourmc_frame_block_header(hdr, usize, csize, pb->xxh32);
            if (fwrite(hdr, 1, 12, w->fout) != 12) { w->code = 3; w->msg = "Write error : cannot write block header"; break; }
            if (fwrite(w->out_buf + pb->dst_off, 1, csize, w->fout) != csize)
                { w->code = 3; w->msg = csize == usize ? "Write error : cannot write block" : "Write error : cannot write compressed block"; break; }
            *w->outsize += 12ull + csize;
            PRINT_LEVEL(3, "==> %.2f%%   ", (double)*w->outsize / *w->filesize * 100);
        } else {
            if (pb->result == FOURMC_BLK_BADSUM)  { w->code = 4; w->msg = "Error : invalid block checksum detected"; break; }
            if (pb->result < 0)                   { w->code = 4; w->msg = "Decoding Failed ! Corrupted input detected !"; break; }
            if (fwrite(w->out_buf + pb->dst_off, 1, (size_t)pb->result, w->fout) != (size_t)pb->result)
                { w->code = 3; w->msg = pb->src_len == pb->dst_cap ? "Write error : cannot write data block" : "Write error : cannot write decoded block\n"; break; }
            *w->filesize += (unsigned long long)pb->result;
        }
    }
    return NULL;
}
static void wjob_start(wjob_t* w)
{
    w->code = 0; w->msg = NULL;
    if (pthread_create(&w->th, NULL, wjob_main, w) == 0) w->running = 1;
    else { w->running = 0; wjob_main(w); }
}
static void wjob_wait(wjob_t* w) { if (w->running) { pthread_join(w->th, NULL); w->running = 0; } }

/* native/4mc.c:164-209 */
static void open_io(int displayLevel, int overwrite, const char* in_name, const char* out_name, FILE** fin, FILE** fout)
{
    if (!strcmp(in_name, FOURMC_STDINMARK)) { PRINT_LEVEL(4, "Using stdin for input\n"); *fin = stdin; }
    else *fin = fopen(in_name, "rb");

    if (!strcmp(out_name, FOURMC_STDOUTMARK)) { PRINT_LEVEL(4, "Using stdout for output\n"); *fout = stdout; }
    else {
        FILE* probe = NULL;
        if (strcmp(out_name, FOURMC_NULMARK)) probe = fopen(out_name, "rb");
        if (probe) {
            fclose(probe);
            if (!overwrite) {
                int ch;
                PRINT_LEVEL(2, "Warning : %s already exists\n", out_name);
                PRINT_LEVEL(2, "Overwrite ? (Y/N) : ");
                if (displayLevel <= 1) DIE(3, "Operation aborted : %s already exists", out_name);
                ch = getchar();
                if (ch != 'Y' && ch != 'y') DIE(3, "Operation aborted : %s already exists", out_name);
            }
        }
        *fout = fopen(out_name, "wb");
    }
    if (!*fin)  DIE(2, "Cannot open input file: %s", in_name);
    if (!*fout) DIE(3, "Cannot open output file: %s", out_name);
}

/* ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------ */
/* Regular file in, regular file out: no stdio and no staging buffers in between.  The input is mapped and handed to the
 * engine as it lies (the H2D copy reads the page cache), the output file is sized and mapped and the engine's D2H copies
 * land in it (on the bench host a copy from or into pageable memory runs at 20-50 GB/s; what is left is the allocation
 * of the output file's pages, tools/ubench/host_io.cpp).  Launches take FOURMC_BATCH_BLOCKS blocks, 512 by default: the
 * kernels are latency-bound per block and need hundreds of blocks to fill the chip.  Pipes, stdin / stdout and anything
 * irregular (any framing or content error) take the streaming path below, which reproduces the reference's behaviour on
 * errors byte for byte; FOURMC_MMAP=0 switches this path off. */
static unsigned fast_batch_blocks(void)
{
    const char* e = getenv("FOURMC_BATCH_BLOCKS");
    long v = e ? atol(e) : 512;
    if (v < 1) v = 1;
    if (v > 4096) v = 4096;
    return (unsigned)v;
}
/* FOURMC_MMAP: unset = compression mapped, decompression streamed (measured on the bench host, 8 GiB on tmpfs: what bounds
 * decompression is the allocation of the output file's pages, 6-7 GB/s however they are written - the streaming path's
 * writer thread does it beside the GPU, the mapped path inside the D2H copies: 2.25 s against 2.58 s); 1 = both mapped;
 * 0 = both streamed */
static int fast_path_wanted(FILE* fin, FILE* fout, struct stat* sin, int decode)
{
    const char* e = getenv("FOURMC_MMAP");
    struct stat so;
    if (e && !strcmp(e, "0")) return 0;
    if (decode && !(e && !strcmp(e, "1"))) return 0;
    if (fin == stdin || fout == stdout) return 0;
    if (fstat(fileno(fin), sin) != 0 || !S_ISREG(sin->st_mode) || sin->st_size <= 0) return 0;
    if (fstat(fileno(fout), &so) != 0 || !S_ISREG(so.st_mode)) return 0;
    return 1;
}

/* returns 0 when the file has been written; -1: nothing of consequence done, take the streaming path */
static int compress_file_mapped(int displayLevel, FILE* fin, const char* out_name, const struct stat* sin, int level, int codec, int codec_level,
                                uint32_t magic, unsigned long long* filesize_out, unsigned long long* outsize_out)
{
    const uint64_t N = (uint64_t)sin->st_size, nblocks = (N + BLOCKSIZE - 1) / BLOCKSIZE;
    const unsigned nbatch = fast_batch_blocks();
    const uint64_t bound = 12 + nblocks * 12 + N + 12 + FOURMC_FOOTERSIZE(nblocks);
    const int fdi = fileno(fin);
    int fdo;                                     /* a descriptor of its own: a shared writable mapping needs O_RDWR, stdio opened "wb" */
    uint8_t *in, *out;
    uint64_t *offsets, *ioff, pos = 12, b0;
    fourmc_block* blk;
    (void)level;
    if (nblocks > 0xFFFFFFFFull / 8) return -1;
    fdo = open(out_name, O_RDWR);
    if (fdo < 0) return -1;
    in = (uint8_t*)mmap(NULL, (size_t)N, PROT_READ, MAP_PRIVATE, fdi, 0);
    if (in == MAP_FAILED) { close(fdo); return -1; }
    if (ftruncate(fdo, (off_t)bound) != 0) { munmap(in, (size_t)N); close(fdo); return -1; }
    out = (uint8_t*)mmap(NULL, (size_t)bound, PROT_READ | PROT_WRITE, MAP_SHARED, fdo, 0);
    if (out == MAP_FAILED) { munmap(in, (size_t)N); if (ftruncate(fdo, 0) != 0) {} close(fdo); return -1; }
    (void)posix_madvise(in, (size_t)N, POSIX_MADV_SEQUENTIAL);
    offsets = (uint64_t*)malloc((size_t)(nblocks + 1) * 8); ioff = (uint64_t*)malloc((size_t)nbatch * 8);
    blk = (fourmc_block*)calloc(nbatch, sizeof *blk);
    if (!offsets || !ioff || !blk) DIE(1, "Allocation error : not enough memory");
    /* the frame header's 12 bytes are reserved like every launch's blocks below (a store into a hole of a sparse mapping on a full
     * disk is a SIGBUS, not a "Write error"): nothing has been written if this fails, the streaming path takes over */
    {
        const int fe = posix_fallocate(fdo, 0, 12);
        if (fe != 0 && fe != EOPNOTSUPP && fe != EINVAL) {
            munmap(out, (size_t)bound); munmap(in, (size_t)N); free(offsets); free(ioff); free(blk);
            if (ftruncate(fdo, 0) != 0) {}
            close(fdo);
            return -1;
        }
    }
    fourmc_frame_header(out, magic);
    for (b0 = 0; b0 < nblocks; b0 += nbatch) {
        const unsigned nb = (unsigned)(nblocks - b0 < nbatch ? nblocks - b0 : nbatch);
        const uint64_t base = b0 * BLOCKSIZE, bytes = (N - base < (uint64_t)nb * BLOCKSIZE) ? N - base : (uint64_t)nb * BLOCKSIZE;
        size_t ib = 0; unsigned b; int rc;
        for (b = 0; b < nb; b++) {
            blk[b].src_off = (uint64_t)b * BLOCKSIZE; blk[b].dst_off = (uint64_t)b * BLOCKSIZE;
            blk[b].src_len = (uint32_t)((bytes - (uint64_t)b * BLOCKSIZE < BLOCKSIZE) ? bytes - (uint64_t)b * BLOCKSIZE : BLOCKSIZE);
            blk[b].dst_cap = blk[b].src_len; blk[b].result = 0; blk[b].xxh32 = 0;
        }
        /* the blocks this launch may write are RESERVED before it: a store into a hole of a sparse mapping on a full disk (or over
         * quota) is a SIGBUS in the middle of a device copy, where the reference - and the streaming path - report "Write error" and
         * exit 3.  Only the frame header has been written by then at the first launch (the file is truncated and the streaming path
         * takes over); later: the reference's exit.  The last launch also reserves the end mark and the footer behind its blocks. */
        {
            const uint64_t want = (uint64_t)nb * 12 + bytes + (b0 + nb >= nblocks ? 12 + FOURMC_FOOTERSIZE(nblocks) : 0);
            const int fe = posix_fallocate(fdo, (off_t)pos, (off_t)(want < bound - pos ? want : bound - pos));
            if (fe != 0 && fe != EOPNOTSUPP && fe != EINVAL) {
                if (b0 != 0) DIE(3, "Write error : cannot write compressed block");
                munmap(out, (size_t)bound); munmap(in, (size_t)N); free(offsets); free(ioff); free(blk);
                if (ftruncate(fdo, 0) != 0) {}
                close(fdo);
                return -1;
            }
        }
        rc = fourmc_host_4mc_encode_image(in + base, (size_t)bytes, blk, nb, codec, codec_level, out + pos, (size_t)(bound - pos), ioff, &ib);
        if (rc != FOURMC_OK) DIE(1, "GPU engine error %d : %s", rc, fourmc_gpu_last_error());
        for (b = 0; b < nb; b++) offsets[b0 + b] = pos + ioff[b];
        pos += ib;
        PRINT_LEVEL(3, "\rRead : %i MB   ==> %.2f%%   ", (int)((base + bytes) >> 20), (double)pos / (double)(base + bytes) * 100);
    }
    memset(out + pos, 0, 12); pos += 12;                              /* end of stream mark     */
    fourmc_frame_footer(out + pos, magic, offsets, (uint32_t)nblocks);
    pos += FOURMC_FOOTERSIZE(nblocks);
    munmap(out, (size_t)bound); munmap(in, (size_t)N);
    if (ftruncate(fdo, (off_t)pos) != 0 || close(fdo) != 0) DIE(3, "Write error : cannot write end of stream");
    free(offsets); free(ioff); free(blk);
    *filesize_out = N; *outsize_out = pos;
    return 0;
}

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* returns 0 when every stream of the file has been decoded into `fout`; -1: take the streaming path from the start (the
 * output is empty again) - whatever is irregular is reported by the code that mirrors the reference */
static int decompress_file_mapped(int displayLevel, FILE* fin, const char* out_name, const struct stat* sin, uint32_t magic, int codec,
                                  unsigned long long* filesize_out)
{
    const uint64_t N = (uint64_t)sin->st_size;
    const unsigned nbatch = fast_batch_blocks();
    const int fdi = fileno(fin);
    int fdo = -1;
    uint8_t *in, *out = NULL;
    uint64_t p = 0, total = 0, nblk = 0, cap = 0, b0;
    struct { uint64_t src, dst; uint32_t csize, usize, sum; } *dsc = NULL;
    fourmc_block* blk = NULL;
    int ok = 0;
    in = (uint8_t*)mmap(NULL, (size_t)N, PROT_READ, MAP_PRIVATE, fdi, 0);
    if (in == MAP_FAILED) return -1;
    /* pass 1: the framing of every stream, as decodeFourMC walks it (native/4mc.c:575-688); anything else -> streaming path */
    while (p < N) {
        if (N - p < 12 || be32(in + p) != magic || fourmc_frame_check_header(in + p, magic) != 0) goto done;
        p += 12;
        for (;;) {
            uint32_t usize, csize, sum;
            if (N - p < 12) goto done;
            fourmc_frame_parse_block_header(in + p, &usize, &csize, &sum);
            p += 12;
            if (usize == 0 && csize == 0 && sum == 0) break;
            if (csize > BLOCKSIZE || usize > BLOCKSIZE || usize == 0 || csize > N - p || csize > usize) goto done;
            if (nblk == cap) {
                cap = cap ? cap * 2 : 4096;
                dsc = realloc(dsc, (size_t)cap * sizeof *dsc);
                if (!dsc) DIE(1, "Allocation error : not enough memory");
            }
            dsc[nblk].src = p; dsc[nblk].dst = total; dsc[nblk].csize = csize; dsc[nblk].usize = usize; dsc[nblk].sum = sum;
            nblk++; p += csize; total += usize;
        }
        {   /* footer: size, checksum, version (native/4mc.c:670-688) */
            uint32_t fsz;
            if (N - p < 4) goto done;
            fsz = be32(in + p);
            if (fsz < 8 || fsz > N - p) goto done;
            if (fourmc_XXH32(in + p, fsz - 4, 0) != be32(in + p + fsz - 4) || be32(in + p + 4) != 1) goto done;
            if (displayLevel >= 3) goto done;                          /* the index listing of -v: the streaming path prints it */
            p += fsz;
        }
    }
    if (nblk == 0) goto done;
    fdo = open(out_name, O_RDWR);
    if (fdo < 0 || ftruncate(fdo, (off_t)total) != 0) goto done;
    out = (uint8_t*)mmap(NULL, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fdo, 0);
    if (out == MAP_FAILED) { out = NULL; goto undo; }
    blk = (fourmc_block*)calloc(nbatch, sizeof *blk);
    if (!blk) DIE(1, "Allocation error : not enough memory");
    (void)posix_madvise(in, (size_t)N, POSIX_MADV_SEQUENTIAL);
    for (b0 = 0; b0 < nblk; b0 += nbatch) {
        const unsigned nb = (unsigned)(nblk - b0 < nbatch ? nblk - b0 : nbatch);
        const uint64_t sbase = dsc[b0].src, send = dsc[b0 + nb - 1].src + dsc[b0 + nb - 1].csize;
        const uint64_t dbase = dsc[b0].dst, dend = dsc[b0 + nb - 1].dst + dsc[b0 + nb - 1].usize;
        unsigned b; int rc;
        for (b = 0; b < nb; b++) {
            blk[b].src_off = dsc[b0 + b].src - sbase; blk[b].dst_off = dsc[b0 + b].dst - dbase;
            blk[b].src_len = dsc[b0 + b].csize; blk[b].dst_cap = dsc[b0 + b].usize;
            blk[b].result = 0; blk[b].xxh32 = dsc[b0 + b].sum;
        }
        rc = fourmc_host_4mc_decode(in + sbase, (size_t)(send - sbase), out + dbase, (size_t)(dend - dbase), blk, nb, codec);
        if (rc != FOURMC_OK) DIE(1, "GPU engine error %d : %s", rc, fourmc_gpu_last_error());
        for (b = 0; b < nb; b++) if (blk[b].result != (int32_t)dsc[b0 + b].usize) goto undo;   /* checksum / content: the streaming path reports it */
    }
    ok = 1;
undo:
    if (out) munmap(out, (size_t)total);
    if (!ok && ftruncate(fdo, 0) != 0) DIE(3, "Write error : cannot write decoded block\n");
done:
    if (fdo >= 0 && close(fdo) != 0 && ok) DIE(3, "Write error : cannot write decoded block\n");
    munmap(in, (size_t)N);
    free(dsc); free(blk);
    if (!ok) return -1;
    *filesize_out = total;
    return 0;
}

static int compress_file(int displayLevel, int overwrite, char* in_name, char* out_name, int level,
                         uint32_t magic)
{
    const unsigned nbatch = batch_blocks();
    unsigned long long filesize = 0, outsize = 0;
    uint64_t* offsets = NULL; size_t noff = 0, capoff = 0;
    uint8_t *in_buf, hdr[12];
    fourmc_block* blk;
    hbuf hin[2], hout[2]; fourmc_block* blks[2]; job_t jobs[2]; wjob_t wj;
    FILE *fin, *fout;
    int codec, codec_level = 0, k, have_prev = 0;
    clock_t t0 = clock(), t1;
    size_t got;

    if (displayLevel == 2 && level > 1) displayLevel = 3;            /* native/4mc.c:241       */
    if (magic == FOURMC_MAGIC_4MC) {                                  /* native/4mc.c:243-253   */
        if (level <= 1) codec = FOURMC_CODEC_LZ4_FAST;
        else if (level == 2) codec = FOURMC_CODEC_LZ4_MC;
        else { codec = FOURMC_CODEC_LZ4_HC; codec_level = (level == 3) ? 4 : 8; }
    } else {                                                          /* native/4mc.c:411-419   */
        codec = FOURMC_CODEC_ZSTD;
        codec_level = level <= 1 ? 1 : level == 2 ? 3 : level == 3 ? 6 : 12;
    }
    open_io(displayLevel, overwrite, in_name, out_name, &fin, &fout);
    {
        struct stat sin;
        if (fast_path_wanted(fin, fout, &sin, 0) &&
            compress_file_mapped(displayLevel, fin, out_name, &sin, level, codec, codec_level, magic, &filesize, &outsize) == 0) {
            fclose(fin); fclose(fout);
            goto report;
        }
    }

    for (k = 0; k < 2; k++) {
        hin[k] = hbuf_alloc((size_t)nbatch * BLOCKSIZE); hout[k] = hbuf_alloc((size_t)nbatch * BLOCKSIZE);
        blks[k] = (fourmc_block*)calloc(nbatch, sizeof *blk);
        if (!hin[k].p || !hout[k].p || !blks[k]) DIE(1, "Allocation error : not enough memory");
        memset(&jobs[k], 0, sizeof jobs[k]);
    }

    fourmc_frame_header(hdr, magic);
    if (fwrite(hdr, 1, 12, fout) != 12) DIE(3, "Write error : cannot write header");
    outsize = 12;

    /* batch i is read while batch i - 1 is on the GPU and batch i - 2 is written (three threads, two buffer sets: the output
     * buffer of batch i is the one batch i - 2 was written from, so its writer is joined before batch i starts) */
    memset(&wj, 0, sizeof wj);
    wj.compress = 1; wj.fout = fout; wj.displayLevel = displayLevel;
    wj.blk = (fourmc_block*)calloc(nbatch, sizeof *wj.blk);
    if (!wj.blk) DIE(1, "Allocation error : not enough memory");
    wj.filesize = &filesize; wj.outsize = &outsize; wj.offsets = &offsets; wj.noff = &noff; wj.capoff = &capoff;
    for (k = 0;; k ^= 1) {
        unsigned nb = 0, b;
        in_buf = (uint8_t*)hin[k].p; blk = blks[k];
        got = fread(in_buf, 1, (size_t)nbatch * BLOCKSIZE, fin);
        if (got > 0) {
            nb = (unsigned)((got + BLOCKSIZE - 1) / BLOCKSIZE);
            for (b = 0; b < nb; b++) {
                blk[b].src_off = (uint64_t)b * BLOCKSIZE;
                blk[b].dst_off = (uint64_t)b * BLOCKSIZE;
                blk[b].src_len = (uint32_t)((got - (size_t)b * BLOCKSIZE < BLOCKSIZE) ? got - (size_t)b * BLOCKSIZE : BLOCKSIZE);
                blk[b].dst_cap = blk[b].src_len;
                blk[b].result = 0; blk[b].xxh32 = 0;
            }
        }
        if (have_prev) {                                  /* results of the previous batch */
            const int p = k ^ 1;
            if (job_wait(&jobs[p]) != FOURMC_OK) { wjob_wait(&wj); DIE(1, "GPU engine error %d : %s", jobs[p].rc, jobs[p].err); }
        }
        wjob_wait(&wj);                                   /* the batch before that is on disk */
        if (wj.msg) DIE(wj.code, "%s", wj.msg);
        if (got > 0) {
            jobs[k].encode = 1; jobs[k].src = in_buf; jobs[k].src_bytes = got; jobs[k].dst = hout[k].p; jobs[k].dst_bytes = (size_t)nb * BLOCKSIZE;
            jobs[k].blk = blk; jobs[k].n = nb; jobs[k].codec = codec; jobs[k].level = codec_level;
            job_start(&jobs[k]);
        }
        if (have_prev) {
            const int p = k ^ 1;
            wj.out_buf = (const uint8_t*)hout[p].p; wj.n = jobs[p].n; memcpy(wj.blk, blks[p], (size_t)wj.n * sizeof *wj.blk);
            wjob_start(&wj);
        }
        have_prev = got > 0;
        if (!have_prev) break;
    }
    wjob_wait(&wj);
    if (wj.msg) DIE(wj.code, "%s", wj.msg);
    memset(hdr, 0, 12);                                               /* end of stream mark     */
    if (fwrite(hdr, 1, 12, fout) != 12) DIE(3, "Write error : cannot write end of stream");
    outsize += 12;
    {
        const size_t fsz = FOURMC_FOOTERSIZE(noff);
        uint8_t* foot = (uint8_t*)malloc(fsz);
        if (!foot) DIE(1, "Allocation error : not enough memory");
        fourmc_frame_footer(foot, magic, offsets, (uint32_t)noff);
        if (fwrite(foot, 1, fsz, fout) != fsz) DIE(3, "Write error : cannot write end of stream");
        outsize += fsz;
        free(foot);
    }
    for (k = 0; k < 2; k++) { hbuf_free(hin[k]); hbuf_free(hout[k]); free(blks[k]); }
    free(wj.blk);
    free(offsets);
    fclose(fin); fclose(fout);

report:
    t1 = clock();
    PRINT_LEVEL(2, "\r%79s\r", "");
    PRINT_LEVEL(2, "Compressed (%s) %llu bytes into %llu bytes ==> %.2f%% (Ratio=%.3f)\n",
                level <= 1 ? "fast" : (level == 2 ? "medium" : (level == 3 ? "high" : "ultra")),
                filesize, outsize, (double)outsize / filesize * 100, (double)100.0 / ((double)outsize / filesize * 100));
    {
        double seconds = (double)(t1 - t0) / CLOCKS_PER_SEC;
        PRINT_LEVEL(4, "Done in %.2f s ==> %.2f MB/s\n", seconds, (double)filesize / seconds / 1024 / 1024);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* One stream (header .. footer).  Returns decoded bytes; 0 with *eof=1 when the input is at EOF. */
static unsigned long long decode_stream(int displayLevel, FILE* fin, FILE* fout, uint32_t magic, int codec, int* eof)
{
    const unsigned nbatch = batch_blocks();
    unsigned long long filesize = 0;
    uint8_t hdr[12], *in_buf;
    hbuf hin[2], hout[2]; fourmc_block* blks[2]; job_t jobs[2]; wjob_t wj;
    fourmc_block* blk;
    size_t n;
    int done = 0, k, have_prev = 0;

    *eof = 0;
    n = fread(hdr, 1, 4, fin);
    if (n == 0) { *eof = 1; return 0; }
    if (n != 4) DIE(4, "Unrecognized header : Magic Number unreadable");
    if ((((uint32_t)hdr[0] << 24) | ((uint32_t)hdr[1] << 16) | ((uint32_t)hdr[2] << 8) | hdr[3]) != magic)
        DIE(4, "Unrecognized header : not a 4mc file");
    if (fread(hdr + 4, 1, 8, fin) != 8) DIE(4, "Unreadable header");
    switch (fourmc_frame_check_header(hdr, magic)) {
        case 2: DIE(4, "Wrong version number");
        case 3: DIE(4, "Wrong header checksum");
        default: break;
    }
    for (k = 0; k < 2; k++) {
        hin[k] = hbuf_alloc((size_t)nbatch * BLOCKSIZE); hout[k] = hbuf_alloc((size_t)nbatch * BLOCKSIZE);
        blks[k] = (fourmc_block*)calloc(nbatch, sizeof *blk);
        if (!hin[k].p || !hout[k].p || !blks[k]) DIE(1, "Allocation error : not enough memory");
        memset(&jobs[k], 0, sizeof jobs[k]);
    }

    /* batch i is gathered while batch i - 1 is on the GPU and batch i - 2 is written (three threads, two buffer sets).  A
     * framing error found while gathering is raised only after the blocks before it have been decoded and written, as the
     * serial reference would; nothing exits while an engine call or a writer is still running. */
    memset(&wj, 0, sizeof wj);
    wj.compress = 0; wj.fout = fout; wj.displayLevel = displayLevel; wj.filesize = &filesize;
    wj.blk = (fourmc_block*)calloc(nbatch, sizeof *wj.blk);
    if (!wj.blk) DIE(1, "Allocation error : not enough memory");
    for (k = 0;; k ^= 1) {
        unsigned nb = 0;
        size_t in_used = 0;
        int pending_code = 0; const char* pending_msg = NULL;
        in_buf = (uint8_t*)hin[k].p; blk = blks[k];
        while (!done && nb < nbatch) {
            uint32_t usize, csize, sum;
            if (fread(hdr, 1, 12, fin) != 12) { pending_code = 2; pending_msg = "Read error : cannot read next block size"; break; }
            fourmc_frame_parse_block_header(hdr, &usize, &csize, &sum);
            if (usize == 0 && csize == 0 && sum == 0) { done = 1; break; }
            if (csize > BLOCKSIZE) { pending_code = 4; pending_msg = "Read error: block size beyond 4MB limit"; break; }
            if (fread(in_buf + in_used, 1, csize, fin) != csize) { pending_code = 2; pending_msg = "Read error : cannot read data block"; break; }
            if (usize != csize && usize > BLOCKSIZE) {
                /* the reference verifies the checksum first (native/4mc.c:645-652) */
                if (fourmc_XXH32(in_buf + in_used, csize, 0) != sum) { pending_code = 4; pending_msg = "Error : invalid block checksum detected"; }
                else { pending_code = 4; pending_msg = "Read error: uncompressed block size beyond 4MB limit"; }
                break;
            }
            blk[nb].src_off = in_used; blk[nb].dst_off = (uint64_t)nb * BLOCKSIZE;
            blk[nb].src_len = csize;   blk[nb].dst_cap = usize;
            blk[nb].result = 0;        blk[nb].xxh32 = sum;
            in_used += csize; nb++;
        }
        if (have_prev) {
            const int p = k ^ 1;
            if (job_wait(&jobs[p]) != FOURMC_OK) { wjob_wait(&wj); DIE(1, "GPU engine error %d : %s", jobs[p].rc, jobs[p].err); }
        }
        wjob_wait(&wj);                                       /* the batch before that is written (its buffer is this batch's) */
        if (wj.msg) DIE(wj.code, "%s", wj.msg);
        if (nb) {
            jobs[k].encode = 0; jobs[k].src = in_buf; jobs[k].src_bytes = in_used; jobs[k].dst = hout[k].p; jobs[k].dst_bytes = (size_t)nb * BLOCKSIZE;
            jobs[k].blk = blk; jobs[k].n = nb; jobs[k].codec = codec; jobs[k].level = 0;
            job_start(&jobs[k]);
        }
        if (have_prev) {
            const int p = k ^ 1;
            wj.out_buf = (const uint8_t*)hout[p].p; wj.n = jobs[p].n; memcpy(wj.blk, blks[p], (size_t)wj.n * sizeof *wj.blk);
            wjob_start(&wj);
        }
        have_prev = nb > 0;
        if (done || pending_msg) {                            /* end of the stream: everything in flight, in order */
            wjob_wait(&wj);
            if (wj.msg) { if (have_prev) job_wait(&jobs[k]); DIE(wj.code, "%s", wj.msg); }
            if (have_prev) {
                if (job_wait(&jobs[k]) != FOURMC_OK) DIE(1, "GPU engine error %d : %s", jobs[k].rc, jobs[k].err);
                wj.out_buf = (const uint8_t*)hout[k].p; wj.n = jobs[k].n; memcpy(wj.blk, blks[k], (size_t)wj.n * sizeof *wj.blk);
                wjob_start(&wj); wjob_wait(&wj);
                if (wj.msg) DIE(wj.code, "%s", wj.msg);
            }
            if (pending_msg) DIE(pending_code, "%s", pending_msg);
            break;
        }
    }
    /* footer (native/4mc.c:670-688) */
    {
        uint32_t fsz; uint8_t* foot; int64_t r;
        if (fread(hdr, 1, 4, fin) != 4) DIE(1, "Unreadable footer");
        fsz = ((uint32_t)hdr[0] << 24) | ((uint32_t)hdr[1] << 16) | ((uint32_t)hdr[2] << 8) | hdr[3];
        if (fsz < 8) DIE(2, "Read error : cannot read footer");
        foot = (uint8_t*)malloc(fsz);
        if (!foot) DIE(1, "Allocation error : not enough memory");
        memcpy(foot, hdr, 4);
        if (fread(foot + 4, 1, fsz - 4, fin) != fsz - 4) DIE(2, "Read error : cannot read footer");
        if (fourmc_XXH32(foot, fsz - 4, 0) != (((uint32_t)foot[fsz - 4] << 24) | ((uint32_t)foot[fsz - 3] << 16) | ((uint32_t)foot[fsz - 2] << 8) | foot[fsz - 1]))
            DIE(4, "Error : invalid footer checksum detected");
        if ((((uint32_t)foot[4] << 24) | ((uint32_t)foot[5] << 16) | ((uint32_t)foot[6] << 8) | foot[7]) != 1)
            DIE(4, "Read error : unsupported footer version");
        if (displayLevel >= 3 && fsz >= 20) {
            const uint32_t total = (fsz - 20) / 4; uint32_t i; unsigned long long abs = 0;
            PRINT_LEVEL(3, "\nBlock index %u entries:\n", total);
            for (i = 0; i < total; i++) {
                const uint8_t* p = foot + 8 + 4 * i;
                const uint32_t delta = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
                abs += delta;
                PRINT_LEVEL(3, " * Block #%u at %llu (+%u)\n", i, abs, delta);
            }
        }
        (void)r;
        free(foot);
    }
    for (k = 0; k < 2; k++) { hbuf_free(hin[k]); hbuf_free(hout[k]); free(blks[k]); }
    free(wj.blk);
    return filesize;
}

static int decompress_file(int displayLevel, int overwrite, char* in_name, char* out_name, uint32_t magic, int codec)
{
    unsigned long long filesize = 0, got;
    FILE *fin, *fout;
    clock_t t0 = clock(), t1;
    int eof = 0;
    open_io(displayLevel, overwrite, in_name, out_name, &fin, &fout);
    {
        struct stat sin;
        if (fast_path_wanted(fin, fout, &sin, 1) && decompress_file_mapped(displayLevel, fin, out_name, &sin, magic, codec, &filesize) == 0) goto report;
        filesize = 0;
    }
    do {                                                              /* concatenated streams   */
        got = decode_stream(displayLevel, fin, fout, magic, codec, &eof);
        filesize += got;
    } while (got);
report:
    t1 = clock();
    PRINT_LEVEL(2, "\r%79s\r", "");
    PRINT_LEVEL(2, "Successfully decoded %llu bytes \n", filesize);
    {
        double seconds = (double)(t1 - t0) / CLOCKS_PER_SEC;
        PRINT_LEVEL(4, "Done in %.2f s ==> %.2f MB/s\n", seconds, (double)filesize / seconds / 1024 / 1024);
    }
    fclose(fin); fclose(fout);
    return 0;
}

int fourMCcompressFilename(int displayLevel, int overwrite, char* in, char* out, int level)
{ return compress_file(displayLevel, overwrite, in, out, level, FOURMC_MAGIC_4MC); }

int fourMZcompressFilename(int displayLevel, int overwrite, char* in, char* out, int level)
{ return compress_file(displayLevel, overwrite, in, out, level, FOURMC_MAGIC_4MZ); }

int fourMcDecompressFileName(int displayLevel, int overwrite, char* in, char* out)
{ return decompress_file(displayLevel, overwrite, in, out, FOURMC_MAGIC_4MC, FOURMC_CODEC_LZ4_FAST); }

int fourMZDecompressFileName(int displayLevel, int overwrite, char* in, char* out)
{ return decompress_file(displayLevel, overwrite, in, out, FOURMC_MAGIC_4MZ, FOURMC_CODEC_ZSTD); }

/* ------------------------------------------------------------------------------------------
 * Random access through the footer block index — what a Hadoop split does with a .4mc / .4mz file
 * (FourMcInputStream.readIndex, java/hadoop-4mc/.../FourMcInputStream.java:163-239, and
 * FourMcBlockIndex.java:92-173): read the trailer, validate the footer, then decode any block
 * range without touching the rest of the file.  Library calls, so errors are RETURNED (negative),
 * never exit():  -1 I/O, -2 bad header/footer, -3 range, -4 corrupt block (checksum / decode),
 * -5 dst too small, -6 engine (fourmc_gpu_last_error()).
 */
static int64_t read_index(FILE* f, uint32_t* magic_out, uint64_t** off_out, uint64_t* data_end)
{
    uint8_t hdr[12], tail[12];
    uint32_t magic, fsz;
    uint8_t* foot;
    int64_t n;
    long long fsize;
    if (fread(hdr, 1, 12, f) != 12) return -1;
    magic = ((uint32_t)hdr[0] << 24) | ((uint32_t)hdr[1] << 16) | ((uint32_t)hdr[2] << 8) | hdr[3];
    if (magic != FOURMC_MAGIC_4MC && magic != FOURMC_MAGIC_4MZ) return -2;
    if (fourmc_frame_check_header(hdr, magic) != 0) return -2;
    if (fseeko(f, -12, SEEK_END) != 0) return -1;
    fsize = (long long)ftello(f) + 12;
    if (fread(tail, 1, 12, f) != 12) return -1;
    fsz = ((uint32_t)tail[0] << 24) | ((uint32_t)tail[1] << 16) | ((uint32_t)tail[2] << 8) | tail[3];
    if (fsz < 20 || (long long)fsz + 24 > fsize) return -2;
    foot = (uint8_t*)malloc(fsz);
    *off_out = (uint64_t*)malloc(((fsz - 20) / 4 + 1) * sizeof(uint64_t));
    if (!foot || !*off_out) { free(foot); return -1; }
    if (fseeko(f, -(long long)fsz, SEEK_END) != 0 || fread(foot, 1, fsz, f) != fsz) { free(foot); return -1; }
    n = fourmc_frame_parse_footer(foot, fsz, magic, *off_out);
    free(foot);
    if (n < 0) return -2;
    *magic_out = magic;
    *data_end = (uint64_t)fsize - fsz - 12;              /* end of the last block = start of the 12 zero bytes */
    return n;
}

int64_t fourmc_file_block_count(const char* path, int* is_zstd)
{
    FILE* f = fopen(path, "rb");
    uint32_t magic = 0; uint64_t* off = NULL; uint64_t end = 0;
    int64_t n;
    if (!f) return -1;
    n = read_index(f, &magic, &off, &end);
    fclose(f); free(off);
    if (n >= 0 && is_zstd) *is_zstd = (magic == FOURMC_MAGIC_4MZ);
    return n;
}

int64_t fourmc_file_decode_blocks(const char* path, uint32_t first, uint32_t count, void* dst, size_t dst_cap)
{
    FILE* f = fopen(path, "rb");
    uint32_t magic = 0, b; uint64_t* off = NULL; uint64_t end = 0, lo, hi, pos;
    int64_t n, total = 0;
    uint8_t* buf = NULL; fourmc_block* blk = NULL;
    if (!f) return -1;
    n = read_index(f, &magic, &off, &end);
    if (n < 0) { fclose(f); free(off); return n; }
    if ((uint64_t)first + count > (uint64_t)n) { fclose(f); free(off); return -3; }
    if (count == 0) { fclose(f); free(off); return 0; }
    lo = off[first]; hi = (first + count < (uint64_t)n) ? off[first + count] : end;
    buf = (uint8_t*)malloc((size_t)(hi - lo) + 64);
    blk = (fourmc_block*)calloc(count, sizeof *blk);
    total = -1;
    if (buf && blk && hi > lo && fseeko(f, (long long)lo, SEEK_SET) == 0 && fread(buf, 1, (size_t)(hi - lo), f) == (size_t)(hi - lo)) {
        uint64_t out = 0;
        total = 0;
        for (b = 0, pos = 0; b < count; b++) {
            uint32_t usize, csize, sum;
            if (off[first + b] - lo != pos || pos + 12 > hi - lo) { total = -2; break; }   /* index and block headers disagree */
            fourmc_frame_parse_block_header(buf + pos, &usize, &csize, &sum);
            if (csize > BLOCKSIZE || usize > BLOCKSIZE || pos + 12 + csize > hi - lo) { total = -4; break; }
            if (out + usize > dst_cap) { total = -5; break; }
            blk[b].src_off = pos + 12; blk[b].dst_off = out; blk[b].src_len = csize; blk[b].dst_cap = usize; blk[b].xxh32 = sum;
            pos += 12 + csize; out += usize;
        }
        if (total == 0) {
            if (fourmc_host_4mc_decode(buf, (size_t)(hi - lo), dst, (size_t)out, blk, count,
                                       magic == FOURMC_MAGIC_4MZ ? FOURMC_CODEC_ZSTD : FOURMC_CODEC_LZ4_FAST) != FOURMC_OK) total = -6;
            else {
                for (b = 0; b < count; b++) if (blk[b].result < 0 || (uint32_t)blk[b].result != blk[b].dst_cap) { total = -4; break; }
                if (total == 0) total = (int64_t)out;
            }
        }
    }
    fclose(f); free(off); free(buf); free(blk);
    return total;
}
