/*
 * tools/shard_rccl.c - C launcher of the several-ranks writer (4mc_amd/csrc/shard.c): one process per GPU, the ONE exchange
 * of the path - the per-block compressed sizes, 4 bytes per block - done with ncclAllGather (RCCL over xGMI).
 *
 *   RANK=r WORLD_SIZE=n LOCAL_RANK=r FOURMC_RDV=/dev/shm/some.id  tools/shard_rccl [-z] [-1..-4] <in> <out>
 *   RANK=r WORLD_SIZE=n LOCAL_RANK=r                              tools/shard_rccl -d <in.4mc|.4mz> <out>     (no exchange at all)
 *
 * The reference has no communication layer; this is what a deployment that shards one file over the GPUs of
 * a node links instead of the torch.distributed callback the tests use.  librccl.so is loaded at run time (dlopen), so that
 * neither this launcher nor libhadoop-4mc.so carries a link-time dependency on it; the rendezvous of the ncclUniqueId is a
 * file (rank 0 writes it, the others wait for it): no MPI needed.
 * RCCL refuses two ranks on one device, so on a one-GPU box only WORLD_SIZE=1 runs (the collective degenerates to a copy,
 * but communicator, stream and call are the real ones); the multi-rank byte layout is covered by tests/test_multirank_cpu.py.
 *
 * Build: gcc -O2 -D__HIP_PLATFORM_AMD__ tools/shard_rccl.c -I include -I /opt/rocm/include -L 4mc_amd/lib -lhadoop-4mc -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$ORIGIN/../4mc_amd/lib -Wl,-rpath,/opt/rocm/lib -ldl -o tools/shard_rccl
 */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include "fourmc.h"
#include "fourmc_gpu.h"

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_t_;
typedef int (*getid_fn)(ncclUniqueId_t*);
typedef int (*init_fn)(ncclComm_t_*, int, ncclUniqueId_t, int);
typedef int (*allgather_fn)(const void*, void*, size_t, int, ncclComm_t_, hipStream_t);
typedef int (*destroy_fn)(ncclComm_t_);
typedef const char* (*errstr_fn)(int);

static struct { ncclComm_t_ comm; allgather_fn allgather; errstr_fn errstr; hipStream_t stream; int world; } G;

/* fourmc_allgather_fn: `bytes` from every rank, rank order, into recv (host memory on both sides) */
static int gather_cb(void* ctx, const void* send, size_t bytes, void* recv)
{
    void *ds = NULL, *dr = NULL; int rc = -1, e;
    (void)ctx;
    if (hipMalloc(&ds, bytes ? bytes : 4) != hipSuccess || hipMalloc(&dr, (bytes ? bytes : 4) * (size_t)G.world) != hipSuccess) goto out;
    if (hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, G.stream) != hipSuccess) goto out;
    e = G.allgather(ds, dr, bytes, /* ncclUint8 */ 1, G.comm, G.stream);
    if (e != 0) { fprintf(stderr, "ncclAllGather: %s\n", G.errstr ? G.errstr(e) : "error"); goto out; }
    if (hipMemcpyAsync(recv, dr, bytes * (size_t)G.world, hipMemcpyDeviceToHost, G.stream) != hipSuccess) goto out;
    if (hipStreamSynchronize(G.stream) != hipSuccess) goto out;
    rc = 0;
out:
    if (ds) (void)hipFree(ds);
    if (dr) (void)hipFree(dr);
    return rc;
}

static int env_int(const char* k, int d) { const char* e = getenv(k); return e ? atoi(e) : d; }

int main(int argc, char** argv)
{
    int level = 1, zstd = 0, decode = 0, i;
    const char *in = NULL, *out = NULL;
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
    const char* rdv = getenv("FOURMC_RDV");
    void* h; getid_fn get_id; init_fn init; destroy_fn destroy;
    ncclUniqueId_t id;
    int rc;
    for (i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-z")) zstd = 1;
        else if (!strcmp(argv[i], "-d")) decode = 1;
        else if (argv[i][0] == '-' && argv[i][1] >= '1' && argv[i][1] <= '4' && !argv[i][2]) level = argv[i][1] - '0';
        else if (!in) in = argv[i];
        else if (!out) out = argv[i];
    }
    if (!in || !out) { fprintf(stderr, "usage: RANK= WORLD_SIZE= LOCAL_RANK= [FOURMC_RDV=file] shard_rccl [-z] [-1..-4] <in> <out>\n"); return 2; }
    if (fourmc_gpu_init(local) != FOURMC_OK) { fprintf(stderr, "GPU engine: %s\n", fourmc_gpu_last_error()); return 1; }
    if (decode) {   /* every rank finds its blocks through the footer index: nothing to gather */
        long long detail = 0;
        rc = fourmc_file_decompress_sharded(in, out, rank, world, &detail);
        if (rc != 0) fprintf(stderr, "rank %d: fourmc_file_decompress_sharded = %d (detail %lld, %s)\n", rank, rc, detail, fourmc_gpu_last_error());
        return rc ? 1 : 0;
    }
    if (hipSetDevice(local) != hipSuccess || hipStreamCreate(&G.stream) != hipSuccess) { fprintf(stderr, "HIP stream\n"); return 1; }
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "librccl.so: %s\n", dlerror()); return 1; }
    get_id = (getid_fn)dlsym(h, "ncclGetUniqueId"); init = (init_fn)dlsym(h, "ncclCommInitRank");
    G.allgather = (allgather_fn)dlsym(h, "ncclAllGather"); destroy = (destroy_fn)dlsym(h, "ncclCommDestroy"); G.errstr = (errstr_fn)dlsym(h, "ncclGetErrorString");
    if (!get_id || !init || !G.allgather || !destroy) { fprintf(stderr, "librccl.so lacks the entry points\n"); return 1; }
    G.world = world;
    if (rank == 0) {
        if (get_id(&id) != 0) { fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
        if (world > 1) {
            char tmp[4096]; FILE* f;
            if (!rdv) { fprintf(stderr, "FOURMC_RDV (rendezvous file) is needed for WORLD_SIZE > 1\n"); return 2; }
            snprintf(tmp, sizeof tmp, "%s.tmp", rdv);
            f = fopen(tmp, "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1 || fclose(f) != 0 || rename(tmp, rdv) != 0) { fprintf(stderr, "cannot write %s\n", rdv); return 1; }
        }
    } else {
        int tries; FILE* f = NULL;
        if (!rdv) { fprintf(stderr, "FOURMC_RDV (rendezvous file) is needed for WORLD_SIZE > 1\n"); return 2; }
        for (tries = 0; tries < 6000 && !(f = fopen(rdv, "rb")); tries++) { struct timespec ts = {0, 10 * 1000 * 1000}; nanosleep(&ts, NULL); }
        if (!f || fread(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "no rendezvous at %s\n", rdv); return 1; }
        fclose(f);
    }
    rc = init(&G.comm, world, id, rank);
    if (rc != 0) { fprintf(stderr, "ncclCommInitRank: %s\n", G.errstr ? G.errstr(rc) : "error"); return 1; }
    rc = fourmc_file_compress_sharded(in, out, level, zstd ? FOURMC_MAGIC_4MZ : FOURMC_MAGIC_4MC, rank, world, gather_cb, NULL);
    if (rc != 0) fprintf(stderr, "rank %d: fourmc_file_compress_sharded = %d (%s)\n", rank, rc, fourmc_gpu_last_error());
    destroy(G.comm);
    if (rank == 0 && rdv && world > 1) unlink(rdv);
    return rc ? 1 : 0;
}
