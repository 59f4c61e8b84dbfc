/*
 * tools/cpu_baseline.c - the host-CPU comparator of bench.py's `cpu_baseline` object.  BASELINE ONLY: never linked into
 * the product; it loads the checker libraries by path:
 *   --lib oracle/_ref/libref4mc.so   the reference's own codec sources compiled where they lie (kind "reference"), or
 *   --lib oracle/liboracle.so        this repository's scalar restatement (kind "port")
 * and runs, per thread, the per-block work of the reference's file loops (native/4mc.c:301-329 compress + XXH32,
 * :637-661 XXH32 + decode) on 4 MiB blocks of the bench corpus (tools/corpus.c generator, or a file given with --data).
 *
 * One thread per PHYSICAL core (first hardware thread of every core, /sys/devices/system/cpu/cpuN/topology), pinned; its
 * buffers are allocated and first touched by the pinned thread itself, so they are NUMA-local; every thread works on its own
 * copies of its blocks.  All threads start together; the rate is bytes done by all threads / wall time of the measured
 * phase (>= --seconds per phase: a compress phase and a decompress phase, each bracketed by a barrier).
 *
 * Output: one JSON object on stdout.
 * Build: gcc -O2 -pthread tools/cpu_baseline.c tools/corpus.c -ldl -o tools/cpu_baseline
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define B (4u << 20)
void corpus_fill(void* dst, size_t n, uint64_t seed, uint64_t first_block);
void corpus_fill_logs(void* dst, size_t n, uint64_t seed, uint64_t first_block);

typedef int (*lz4c_fn)(const char*, char*, int, int);
typedef int (*lz4hc_fn)(const char*, char*, int, int, int);
typedef int (*lz4d_fn)(const char*, char*, int, int);
typedef size_t (*zc_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zd_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*xxh_fn)(const void*, size_t, unsigned);
/* the port's names and argument orders differ */
typedef int (*o_lz4c_fn)(const uint8_t*, uint8_t*, int, int);
typedef int (*o_lz4hc_fn)(const uint8_t*, uint8_t*, int, int, int);
typedef int64_t (*o_zc_fn)(const uint8_t*, size_t, uint8_t*, size_t, int);
typedef int64_t (*o_zd_fn)(const uint8_t*, size_t, uint8_t*, size_t);

static struct {
    int is_ref;
    lz4c_fn lz4c; lz4hc_fn lz4hc; lz4d_fn lz4d; zc_fn zc; zd_fn zd; xxh_fn xxh;
    o_lz4c_fn o_lz4c; o_lz4hc_fn o_lz4hc; o_lz4c_fn o_lz4d; o_zc_fn o_zc; o_zd_fn o_zd; xxh_fn o_xxh;
} L;

enum { C_LZ4, C_HC4, C_ZSTD };
static int g_codec, g_level, g_nblk, g_per_thread;
static double g_seconds;
static const uint8_t* g_corpus;
static pthread_barrier_t g_bar;
static volatile int g_stop;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int64_t do_compress(const uint8_t* s, uint8_t* d, int n, int cap)
{
    if (L.is_ref) {
        if (g_codec == C_LZ4) return L.lz4c((const char*)s, (char*)d, n, cap);
        if (g_codec == C_HC4) return L.lz4hc((const char*)s, (char*)d, n, cap, g_level);
        { size_t r = L.zc(d, (size_t)cap, s, (size_t)n, g_level); return r > (size_t)cap ? 0 : (int64_t)r; }   /* error codes are huge */
    }
    if (g_codec == C_LZ4) return L.o_lz4c(s, d, n, cap);
    if (g_codec == C_HC4) return L.o_lz4hc(s, d, n, cap, g_level);
    { int64_t r = L.o_zc(s, (size_t)n, d, (size_t)cap, g_level); return r < 0 ? 0 : r; }
}
static int64_t do_decompress(const uint8_t* s, int n, uint8_t* d, int cap)
{
    if (L.is_ref) return g_codec == C_ZSTD ? (int64_t)L.zd(d, (size_t)cap, s, (size_t)n) : L.lz4d((const char*)s, (char*)d, n, cap);
    return g_codec == C_ZSTD ? L.o_zd(s, (size_t)n, d, (size_t)cap) : L.o_lz4d(s, d, n, cap);
}
static unsigned do_xxh(const void* p, size_t n) { return L.is_ref ? L.xxh(p, n, 0) : L.o_xxh(p, n, 0); }

typedef struct { int tid, cpu; double c_bytes, d_bytes, c_wall, d_wall; uint64_t csum; int bad; } targ;

static void* worker(void* a)
{
    targ* t = (targ*)a;
    cpu_set_t set; CPU_ZERO(&set); CPU_SET(t->cpu, &set);
    pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    /* own, first-touched copies: NUMA-local */
    const int nb = g_per_thread;
    uint8_t* in = (uint8_t*)malloc((size_t)nb * B);
    uint8_t* comp = (uint8_t*)malloc((size_t)nb * (B + B / 128 + 1024));
    uint8_t* back = (uint8_t*)malloc(B);
    int* csz = (int*)calloc((size_t)nb, sizeof(int));
    for (int i = 0; i < nb; i++) memcpy(in + (size_t)i * B, g_corpus + (size_t)((t->tid * nb + i) % g_nblk) * B, B);
    memset(comp, 0, (size_t)nb * (B + B / 128 + 1024)); memset(back, 0, B);
    /* compress phase */
    pthread_barrier_wait(&g_bar);
    double t0 = now(), done = 0; int k = 0;
    do {
        uint8_t* out = comp + (size_t)k * (B + B / 128 + 1024);
        int64_t r = do_compress(in + (size_t)k * B, out, (int)B, (int)B - 1);            /* capacity n - 1: native/4mc.c:301 */
        const int stored = !(r > 0 && r < (int64_t)B);
        t->csum += do_xxh(stored ? in + (size_t)k * B : out, stored ? B : (size_t)r);        /* :311 / :323 */
        csz[k] = stored ? (int)B : (int)r;
        done += B; k = (k + 1) % nb;
    } while (!g_stop && (k != 0 || now() - t0 < g_seconds));
    t->c_wall = now() - t0; t->c_bytes = done;
    pthread_barrier_wait(&g_bar);
    /* every block has been compressed at least once (k wrapped): decompress phase */
    pthread_barrier_wait(&g_bar);
    t0 = now(); done = 0; k = 0;
    do {
        const uint8_t* src = comp + (size_t)k * (B + B / 128 + 1024);
        const int stored = csz[k] == (int)B;
        t->csum += do_xxh(stored ? in + (size_t)k * B : src, (size_t)csz[k]);                 /* :637 / :645 */
        if (stored) memcpy(back, in + (size_t)k * B, B);
        else if (do_decompress(src, csz[k], back, (int)B) != (int64_t)B) t->bad++;
        done += B; k = (k + 1) % nb;
    } while (k != 0 || now() - t0 < g_seconds);
    t->d_wall = now() - t0; t->d_bytes = done;
    if (memcmp(back, in + (size_t)(nb - 1) * B, B) != 0) t->bad++;
    free(in); free(comp); free(back); free(csz);
    return NULL;
}

static int physical_cpus(int* out, int max)
{
    int n = 0, ncpu = (int)sysconf(_SC_NPROCESSORS_ONLN);
    for (int c = 0; c < ncpu && n < max; c++) {
        char p[128]; snprintf(p, sizeof p, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        FILE* f = fopen(p, "r");
        int first = c;
        if (f) { if (fscanf(f, "%d", &first) != 1) first = c; fclose(f); }
        if (first == c) out[n++] = c;                       /* the first hardware thread of its core */
    }
    return n;
}

int main(int argc, char** argv)
{
    const char* lib = "oracle/_ref/libref4mc.so"; const char* codec = "lz4"; const char* data = NULL;
    int threads = 0, logs = 0; g_seconds = 5; g_nblk = 48; g_per_thread = 2;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--lib") && i + 1 < argc) lib = argv[++i];
        else if (!strcmp(argv[i], "--codec") && i + 1 < argc) codec = argv[++i];
        else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) g_seconds = atof(argv[++i]);
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--blocks") && i + 1 < argc) g_nblk = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--per-thread") && i + 1 < argc) g_per_thread = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--logs")) logs = 1;
        else if (!strcmp(argv[i], "--data") && i + 1 < argc) data = argv[++i];
        else { fprintf(stderr, "usage: cpu_baseline --lib so --codec lz4|hc4|zstd1|zstd3|zstd6|zstd12 [--seconds s] [--threads n] [--blocks n] [--per-thread n] [--logs] [--data file]\n"); return 2; }
    }
    if (!strcmp(codec, "lz4")) g_codec = C_LZ4;
    else if (!strcmp(codec, "hc4")) { g_codec = C_HC4; g_level = 4; }
    else if (!strncmp(codec, "zstd", 4)) { g_codec = C_ZSTD; g_level = atoi(codec + 4); }
    else return 2;
    void* h = dlopen(lib, RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    L.is_ref = dlsym(h, "LZ4_compress_default") != NULL;
    if (L.is_ref) {
        L.lz4c = (lz4c_fn)dlsym(h, "LZ4_compress_default"); L.lz4hc = (lz4hc_fn)dlsym(h, "LZ4_compress_HC"); L.lz4d = (lz4d_fn)dlsym(h, "LZ4_decompress_safe");
        L.zc = (zc_fn)dlsym(h, "ZSTD_compress"); L.zd = (zd_fn)dlsym(h, "ZSTD_decompress"); L.xxh = (xxh_fn)dlsym(h, "XXH32");
    } else {
        L.o_lz4c = (o_lz4c_fn)dlsym(h, "orc_lz4_compress_fast"); L.o_lz4hc = (o_lz4hc_fn)dlsym(h, "orc_lz4hc_compress"); L.o_lz4d = (o_lz4c_fn)dlsym(h, "orc_lz4_decompress_safe");
        L.o_zc = (o_zc_fn)dlsym(h, "orc_zstd_compress"); L.o_zd = (o_zd_fn)dlsym(h, "orc_zstd_decompress"); L.o_xxh = (xxh_fn)dlsym(h, "orc_xxh32");
        if (!L.o_lz4c || !L.o_xxh) { fprintf(stderr, "no codec entry points in %s\n", lib); return 1; }
    }
    uint8_t* corpus = (uint8_t*)malloc((size_t)g_nblk * B);
    if (data) {
        FILE* f = fopen(data, "rb");
        if (!f || fread(corpus, B, (size_t)g_nblk, f) != (size_t)g_nblk) { fprintf(stderr, "cannot read %d blocks from %s\n", g_nblk, data); return 1; }
        fclose(f);
    } else if (logs) corpus_fill_logs(corpus, (size_t)g_nblk * B, 0x4D43, 0);
    else corpus_fill(corpus, (size_t)g_nblk * B, 0x4D43, 0);
    g_corpus = corpus;

    int cpus[4096]; const int nphys = physical_cpus(cpus, 4096);
    const int logical = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (threads <= 0 || threads > nphys) threads = nphys;
    pthread_barrier_init(&g_bar, NULL, (unsigned)threads + 1);
    targ* ta = (targ*)calloc((size_t)threads, sizeof *ta); pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof *th);
    for (int i = 0; i < threads; i++) { ta[i].tid = i; ta[i].cpu = cpus[i]; pthread_create(&th[i], NULL, worker, &ta[i]); }
    pthread_barrier_wait(&g_bar); double c0 = now();
    pthread_barrier_wait(&g_bar); double c1 = now();
    pthread_barrier_wait(&g_bar); double d0 = now();
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    double d1 = now();
    double cb = 0, db = 0, cmin = 1e30, dmin = 1e30; int bad = 0;
    for (int i = 0; i < threads; i++) {
        cb += ta[i].c_bytes; db += ta[i].d_bytes; bad += ta[i].bad;
        const double cr = ta[i].c_bytes / ta[i].c_wall, dr = ta[i].d_bytes / ta[i].d_wall;
        if (cr < cmin) cmin = cr; if (dr < dmin) dmin = dr;
    }
    printf("{\"kind\": \"%s\", \"codec\": \"%s\", \"threads\": %d, \"cores_physical\": %d, \"cpus_logical\": %d, "
           "\"compress_GBps\": %.3f, \"decompress_GBps\": %.3f, \"compress_GBps_per_thread\": %.4f, \"decompress_GBps_per_thread\": %.4f, "
           "\"compress_GBps_slowest_thread\": %.4f, \"decompress_GBps_slowest_thread\": %.4f, "
           "\"compress_seconds\": %.2f, \"decompress_seconds\": %.2f, \"blocks_per_thread\": %d, \"corpus_blocks\": %d, \"corpus\": \"%s\", \"round_trip_failures\": %d}\n",
           L.is_ref ? "reference" : "port", codec, threads, nphys, logical,
           cb / (c1 - c0) / 1e9, db / (d1 - d0) / 1e9, cb / (c1 - c0) / 1e9 / threads, db / (d1 - d0) / 1e9 / threads,
           cmin / 1e9, dmin / 1e9, c1 - c0, d1 - d0, g_per_thread, g_nblk, data ? data : (logs ? "log corpus" : "S-mix"), bad);
    return bad ? 3 : 0;
}
