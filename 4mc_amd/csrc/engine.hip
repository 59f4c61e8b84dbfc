// 4mc_amd/csrc/engine.hip — host side of the C ABI in include/fourmc_gpu.h: device selection,
// batched launches, and the host-buffer staging used by the CLI / JNI entry points.
//
// There is NO CPU codec behind these calls: without a usable gfx950 device every entry point
// fails with FOURMC_ENODEV (the product must fail loudly rather than fall back).
#include <hip/hip_runtime.h>
#include <mutex>
#include <condition_variable>
#include <vector>
#include <algorithm>
#include <map>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fourmc_gpu.h"
#include "kernels.h"

namespace {

thread_local char g_err[512] = "";
std::mutex g_mu;                 // guards device selection and the host-staging arena
std::atomic<int> g_device{-1};
char g_arch[128] = "";

int fail_hip(hipError_t e, const char* what)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return FOURMC_EHIP;
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail_hip(e_, #x); } while (0)

// The HIP current device belongs to the THREAD: the file API runs engine calls on helper threads, which would otherwise
// work on device 0 whatever fourmc_gpu_init / FOURMC_DEVICE selected.  Every entry point passes through here.
int ensure_device()
{
    int d = g_device.load(std::memory_order_acquire);
    if (d < 0) { if (int r = fourmc_gpu_init(-1)) return r; d = g_device.load(std::memory_order_acquire); }
    // (the embedding application may have changed this thread's current device between two calls: ask, do not remember)
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != d) HIP_TRY(hipSetDevice(d));
    return FOURMC_OK;
}

// Host-staging arena: device buffers reused across host-buffer calls (grown on demand).
struct Arena {
    void* d_src = nullptr; size_t src_cap = 0;
    void* d_dst = nullptr; size_t dst_cap = 0;
    fourmc_block* d_blk = nullptr; size_t blk_cap = 0;
    hipStream_t stream = nullptr;
    int device = -1;
};
// Shard 0 serves every host-buffer call; with FOURMC_GPUS=N > 1 a batch of blocks is cut into N contiguous block ranges,
// one arena (device buffers + stream) per range, placed round robin on the visible devices starting at the selected one.
// N may exceed the number of devices: the surplus shards share devices (that is how the sharding is exercised on a
// one-GPU box).  One process, several devices: the per-range results come back to host memory, so the "gather of the
// per-rank block index" of the multi-process layout (bench.py, container.py: RCCL all_gather) is a plain array here.
constexpr int kMaxShards = 16;
Arena g_arenas[kMaxShards];
Arena& g_arena = g_arenas[0];

int host_shards()
{
    static int n = [] { const char* e = getenv("FOURMC_GPUS"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > kMaxShards ? kMaxShards : v; }();
    return n;
}

// Device workspaces (per-block scratch of the kernels: LZ4 decode records, LZ4 HC/MC tables, zstd literals / sequence
// areas / encoder tables).  ONE GROWABLE BUFFER PER STREAM: launches on a stream are ordered, so consecutive calls on the
// same stream can share a buffer, while calls on different streams (the JNI queue's own stream next to a device-API
// caller's, two device-API callers) never see each other's scratch.  A lease keeps the stream's entry locked from the
// moment the pointer is handed out until the caller's launches are enqueued; growing synchronizes THAT stream before the
// old buffer is freed, so no launch can still be using it.
struct StreamWs { std::mutex mu; void* p = nullptr; size_t cap = 0; };
std::mutex g_wsmu;
std::map<hipStream_t, StreamWs*> g_ws;
StreamWs* g_ws_last = nullptr;          // for fourmc_gpu_debug_read_workspace

class WsLease {
    StreamWs* w_ = nullptr;
public:
    WsLease() {}
    WsLease(const WsLease&) = delete;
    ~WsLease() { if (w_) w_->mu.unlock(); }
    int get(hipStream_t s, size_t need, void** out)
    {
        if (!w_) {
            {
                std::lock_guard<std::mutex> lk(g_wsmu);
                auto it = g_ws.find(s);
                if (it == g_ws.end()) it = g_ws.emplace(s, new StreamWs()).first;
                w_ = it->second; g_ws_last = w_;
            }
            w_->mu.lock();          // entries are never removed; the registry lock is not held while waiting here
        }
        if (need > w_->cap) {
            if (w_->p) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(w_->p)); w_->p = nullptr; w_->cap = 0; }
            const size_t want = need + need / 16 + 4096;
            const hipError_t me = hipMalloc(&w_->p, want);
            if (me == hipErrorOutOfMemory) {          // the one failure a caller may answer with smaller pieces
                (void)hipGetLastError(); w_->p = nullptr;
                snprintf(g_err, sizeof g_err, "hipMalloc(%zu bytes of workspace): out of memory", want);
                return FOURMC_ENOMEM;
            }
            HIP_TRY(me);
            w_->cap = want;
        }
        *out = w_->p;
        return FOURMC_OK;
    }
};

} // namespace (the export below is part of the boundary)
// Gives the device back every workspace the engine keeps per stream (a 64 GiB decode leaves 92 GB behind on its stream; they are kept
// because the next call of the same size reuses them).  Each stream is synchronized before its buffer goes; a lease in flight on
// another thread is waited for.  Returns FOURMC_OK or a device error.
extern "C" int fourmc_gpu_release_workspaces(void)
{
    std::vector<std::pair<hipStream_t, StreamWs*>> all;
    { std::lock_guard<std::mutex> lk(g_wsmu); for (auto& kv : g_ws) all.push_back(kv); }
    for (auto& kv : all) {
        std::lock_guard<std::mutex> lk(kv.second->mu);
        if (kv.second->p) {
            HIP_TRY(hipStreamSynchronize(kv.first)); HIP_TRY(hipFree(kv.second->p));
            kv.second->p = nullptr; kv.second->cap = 0;
        }
    }
    return FOURMC_OK;
}
namespace {

// The LZ4 decode workspace.  Path, piece size and bytes are resolved once (fourmc_lz4_decode_plan) and handed to the launcher; when
// the device cannot give the workspace the pieces halve FOR THIS CALL (the launch is then cut into more of them), and an automatic
// choice ends at the walk + window copier, which needs none.
int lease_lz4_decode(WsLease& ws, hipStream_t s, uint32_t n, void** work, fourmc_lz4_plan* plan)
{
    fourmc_lz4_plan prev; prev.path = -1; prev.batch = 0; prev.work_bytes = 0; prev.ok = 0;
    for (uint32_t shrink = 0;; shrink++) {
        *plan = fourmc_lz4_decode_plan(n, shrink);
        if (!plan->ok) { snprintf(g_err, sizeof g_err, "LZ4 decode workspace cannot be allocated"); return FOURMC_ENOMEM; }
        if (plan->work_bytes == 0) { *work = nullptr; return FOURMC_OK; }
        const int r = ws.get(s, plan->work_bytes, work);
        if (r == FOURMC_OK) return r;
        // only "out of memory" is answered with smaller pieces: any other device error is the caller's to see; and a plan that did
        // not change (a path without a shrinkable batch) cannot succeed on the next turn either (this loop spun forever)
        if (r != FOURMC_ENOMEM) return r;
        if (shrink && plan->path == prev.path && plan->batch == prev.batch && plan->work_bytes == prev.work_bytes) return r;
        prev = *plan;
    }
}

// A launch whose workspace grows with its blocks.  When the device cannot give the memory for all of them at once the launch is cut
// into pieces, halved until the allocation succeeds: the blocks are independent, so the results are the same and only the chip is
// filled less well (a 512-block batch of level-12 tables is 25 GiB; nothing retried with a smaller batch).
// `cap`: the most blocks a piece may have (0: no limit).  FOURMC_WS_FAIL_ABOVE=N makes leases of more than N bytes fail (test aid).
template <class Bytes, class Launch>
int in_pieces(hipStream_t s, uint32_t n, uint32_t cap, Bytes bytes, Launch launch)
{
    static const size_t fail_above = [] { const char* e = getenv("FOURMC_WS_FAIL_ABOVE"); return e ? size_t(strtoull(e, nullptr, 10)) : size_t(0); }();
    uint32_t piece = cap && cap < n ? cap : n;
    WsLease ws; void* work = nullptr;
    for (;;) {
        const size_t need = bytes(piece);
        int r = FOURMC_OK;
        if (fail_above && need > fail_above) { snprintf(g_err, sizeof g_err, "workspace of %zu bytes refused (FOURMC_WS_FAIL_ABOVE)", need); r = FOURMC_ENOMEM; }
        else r = ws.get(s, need, &work);
        if (r == FOURMC_OK) break;
        if (r != FOURMC_ENOMEM || piece <= 1) return r;
        piece = (piece + 1) / 2;
    }
    for (uint32_t b0 = 0; b0 < n; b0 += piece)
        if (int r = launch(b0, n - b0 < piece ? n - b0 : piece, work)) return r;
    return FOURMC_OK;
}

int arena_reserve(Arena& g_arena, size_t src_bytes, size_t dst_bytes, size_t nblk)
{
    if (!g_arena.stream) HIP_TRY(hipStreamCreateWithFlags(&g_arena.stream, hipStreamNonBlocking));
    auto grow = [](void** p, size_t* cap, size_t need) -> hipError_t {
        if (need <= *cap) return hipSuccess;
        if (*p) { hipError_t e = hipFree(*p); if (e != hipSuccess) return e; *p = nullptr; *cap = 0; }
        size_t want = need + need / 4 + 4096;
        hipError_t e = hipMalloc(p, want);
        if (e == hipSuccess) *cap = want;
        return e;
    };
    HIP_TRY(grow(&g_arena.d_src, &g_arena.src_cap, src_bytes + 64));
    HIP_TRY(grow(&g_arena.d_dst, &g_arena.dst_cap, dst_bytes + 64));
    HIP_TRY(grow(reinterpret_cast<void**>(&g_arena.d_blk), &g_arena.blk_cap, nblk * sizeof(fourmc_block)));
    return FOURMC_OK;
}

} // namespace

extern "C" {

const char* fourmc_gpu_last_error(void) { return g_err; }
const char* fourmc_gpu_arch(void) { return g_arch; }

int fourmc_gpu_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { fail_hip(e, "hipGetDeviceCount"); return FOURMC_ENODEV; }
    return n;
}

int fourmc_gpu_init(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = fourmc_gpu_device_count();
    if (n <= 0) {
        if (n == 0) snprintf(g_err, sizeof g_err, "no HIP device visible (4mc GPU engine has no CPU fallback)");
        return FOURMC_ENODEV;
    }
    if (device < 0) {
        // honour an already-selected device (one process per GPU: LOCAL_RANK / hipSetDevice upstream)
        const char* env = getenv("FOURMC_DEVICE");
        if (env) device = atoi(env);
        else if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) { snprintf(g_err, sizeof g_err, "device %d out of range (%d visible)", device, n); return FOURMC_EINVAL; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(g_arch, sizeof g_arch, "%s", prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof g_err, "device %d is %s; this build carries gfx950 code only", device, prop.gcnArchName);
        return FOURMC_ENODEV;
    }
    g_device.store(device, std::memory_order_release);
    return FOURMC_OK;
}

// ------------------------------------------------------------------------ device-resident API
int fourmc_gpu_lz4_decompress(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, void* stream)
{
    if (int r = ensure_device()) return r;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WsLease ws; void* work = nullptr;
    fourmc_lz4_plan plan;
    if (int r = lease_lz4_decode(ws, s, n, &work, &plan)) return r;
    HIP_TRY(fourmc_launch_lz4_decode(d_src, d_dst, d_blocks, n, 0, &plan, work, s));
    return FOURMC_OK;
}

// LZ4 fast encoder of the launches: 0 the reference parse, byte for byte (default); 1 the ratio-tolerance encoder
// (lz4_par_encode.hip: valid LZ4 blocks, not the reference's bytes).  FOURMC_LZ4_ENCODE = exact | parallel.
static std::atomic<int> g_lz4_encode_mode{-1};
void fourmc_gpu_set_lz4_encode_mode(int mode) { g_lz4_encode_mode.store(mode == 1 ? 1 : 0, std::memory_order_release); }
int fourmc_gpu_get_lz4_encode_mode(void)
{
    int m = g_lz4_encode_mode.load(std::memory_order_acquire);
    if (m < 0) {
        const char* e = getenv("FOURMC_LZ4_ENCODE");
        m = (e && (!strcmp(e, "parallel") || !strcmp(e, "par") || !strcmp(e, "1"))) ? 1 : 0;
        g_lz4_encode_mode.store(m, std::memory_order_release);
    }
    return m;
}

static int lz4_fast_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, int container_mode, hipStream_t s)
{
    if (fourmc_gpu_get_lz4_encode_mode() == 1) {
        // pieces bound the workspace (4 MiB per block)
        return in_pieces(s, n, 4096, fourmc_lz4_par_work_bytes, [&](uint32_t b0, uint32_t m, void* work) -> int {
            HIP_TRY(fourmc_launch_lz4_encode_par(d_src, d_dst, d_blocks + b0, m, container_mode, work, s)); return FOURMC_OK; });
    }
    HIP_TRY(fourmc_launch_lz4_encode_fast(d_src, d_dst, d_blocks, n, container_mode, s));
    return FOURMC_OK;
}

int fourmc_gpu_lz4_compress_fast(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, void* stream)
{
    if (int r = ensure_device()) return r;
    return lz4_fast_encode(d_src, d_dst, d_blocks, n, 0, static_cast<hipStream_t>(stream));
}

int fourmc_gpu_lz4_compress_hc(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, int level, void* stream)
{
    if (int r = ensure_device()) return r;
    if (level < 1 || level > 8) { snprintf(g_err, sizeof g_err, "LZ4 HC level %d not on the device (hash-chain levels 1..8 are)", level); return FOURMC_EUNSUP; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    return in_pieces(s, n, 0, fourmc_lz4hc_work_bytes, [&](uint32_t b0, uint32_t m, void* work) -> int {
        HIP_TRY(fourmc_launch_lz4hc_encode(d_src, d_dst, d_blocks + b0, m, work, level, 0, s)); return FOURMC_OK; });
}

int fourmc_gpu_lz4_compress_mc(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, void* stream)
{
    if (int r = ensure_device()) return r;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return in_pieces(s, n, 0, fourmc_lz4hc_work_bytes, [&](uint32_t b0, uint32_t m, void* work) -> int {      // same layout as HC
        HIP_TRY(fourmc_launch_lz4mc_encode(d_src, d_dst, d_blocks + b0, m, work, 0, s)); return FOURMC_OK; });
}

#ifdef FOURMC_RESEARCH      /* debug / profiling exports: the research side build only (make research), never the product's ABI */
/* profiling aid: copies `bytes` of the shared per-block workspace at `offset` to the host (phase cycle counters
 * the zstd kernels leave behind their literal buffers) */
int fourmc_gpu_debug_read_workspace(void* host, size_t offset, size_t bytes)
{
    if (int r = ensure_device()) return r;
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_wsmu);                 // the workspace of the stream that was served last
    if (!g_ws_last || !g_ws_last->p || offset + bytes > g_ws_last->cap) { snprintf(g_err, sizeof g_err, "workspace range out of bounds"); return FOURMC_EINVAL; }
    HIP_TRY(hipMemcpy(host, static_cast<char*>(g_ws_last->p) + offset, bytes, hipMemcpyDeviceToHost));
    return FOURMC_OK;
}

/* test aid: run ONLY the parser of the block-parallel LZ4 decoder and copy the first `bytes` of block 0's workspace slot
 * (header, window descriptors, token positions: lz4par.h) to the host */
int fourmc_gpu_debug_lz4_parse(const void* d_src, const void* d_dst, fourmc_block* d_blocks, uint32_t n, int container_mode,
                               void* host, size_t bytes, size_t* layout)
{
    if (int r = ensure_device()) return r;
    if (layout) { layout[0] = fourmc_lz4_parse_work_bytes(1); layout[1] = 64; layout[2] = fourmc_lz4_decode_tok_offset(); }
    if (n == 0) return FOURMC_OK;
    WsLease ws; void* work = nullptr;
    if (int r = ws.get(nullptr, fourmc_lz4_parse_work_bytes(n), &work)) return r;
    HIP_TRY(fourmc_launch_lz4_parse(d_src, d_dst, d_blocks, n, container_mode, work, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    const size_t have = fourmc_lz4_parse_work_bytes(n);
    if (host && bytes) HIP_TRY(hipMemcpy(host, work, bytes < have ? bytes : have, hipMemcpyDeviceToHost));
    return FOURMC_OK;
}

#endif

int fourmc_gpu_zstd_decompress(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, void* stream)
{
    if (int r = ensure_device()) return r;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return in_pieces(s, n, 0, fourmc_zstd_scratch_bytes, [&](uint32_t b0, uint32_t m, void* scratch) -> int {
        HIP_TRY(fourmc_launch_zstd_decode(d_src, d_dst, d_blocks + b0, m, scratch, 0, s)); return FOURMC_OK; });
}

// zstd levels on the device: 1..12 (fast, dfast, greedy, lazy, lazy2 and - for the short last block of a file - btlazy2 / btopt: every
// strategy clevels.h names for them); 4mz itself uses 1, 3, 6, 12.  Levels 13 and above (btlazy2 on full blocks, btultra) and levels
// below 1 are refused rather than answered with bytes the reference would not emit
static int zstd_level_ok(const fourmc_block*, uint32_t, int level, hipStream_t)
{
    if (fourmc_zstd_enc_level_ok(level)) return FOURMC_OK;
    snprintf(g_err, sizeof g_err, "ZSTD level %d not on the device (levels 1..12 are)", level);
    return FOURMC_EUNSUP;
}

static int zstd_enc_serial() { const char* e = getenv("FOURMC_ZSTD_SERIAL"); return e && *e == '1'; }

int fourmc_gpu_zstd_compress(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n, int level, void* stream)
{
    if (int r = ensure_device()) return r;
    if (int r = zstd_level_ok(d_blocks, n, level, static_cast<hipStream_t>(stream))) return r;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return in_pieces(s, n, 0, [&](uint32_t m) { return fourmc_zstd_enc_work_bytes(m, level); }, [&](uint32_t b0, uint32_t m, void* work) -> int {
        HIP_TRY(fourmc_launch_zstd_encode(d_src, d_dst, d_blocks + b0, m, work, 0, level, zstd_enc_serial(), s)); return FOURMC_OK; });
}

int fourmc_gpu_xxh32(const void* d_src, fourmc_block* d_blocks, uint32_t n, uint32_t seed, void* stream)
{
    if (int r = ensure_device()) return r;
    HIP_TRY(fourmc_launch_xxh32(d_src, d_blocks, n, seed, FOURMC_HASH_SRC, static_cast<hipStream_t>(stream)));
    return FOURMC_OK;
}

int fourmc_gpu_4mc_encode_blocks(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                 int codec, int level, void* stream)
{
    if (int r = ensure_device()) return r;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (codec == FOURMC_CODEC_LZ4_HC) {
        if (level < 1 || level > 8) { snprintf(g_err, sizeof g_err, "LZ4 HC level %d not on the device", level); return FOURMC_EUNSUP; }
        if (int r = in_pieces(s, n, 0, fourmc_lz4hc_work_bytes, [&](uint32_t b0, uint32_t m, void* work) -> int {
                HIP_TRY(fourmc_launch_lz4hc_encode(d_src, d_dst, d_blocks + b0, m, work, level, 1, s)); return FOURMC_OK; })) return r;
        HIP_TRY(fourmc_launch_xxh32(d_dst, d_blocks, n, 0, FOURMC_HASH_DST_RESULT, s));
        return FOURMC_OK;
    }
    if (codec == FOURMC_CODEC_LZ4_MC) {
        if (int r = in_pieces(s, n, 0, fourmc_lz4hc_work_bytes, [&](uint32_t b0, uint32_t m, void* work) -> int {
                HIP_TRY(fourmc_launch_lz4mc_encode(d_src, d_dst, d_blocks + b0, m, work, 1, s)); return FOURMC_OK; })) return r;
        HIP_TRY(fourmc_launch_xxh32(d_dst, d_blocks, n, 0, FOURMC_HASH_DST_RESULT, s));
        return FOURMC_OK;
    }
    if (codec == FOURMC_CODEC_ZSTD) {
        if (int r = zstd_level_ok(d_blocks, n, level, s)) return r;
        if (int r = in_pieces(s, n, 0, [&](uint32_t m) { return fourmc_zstd_enc_work_bytes(m, level); }, [&](uint32_t b0, uint32_t m, void* work) -> int {
                HIP_TRY(fourmc_launch_zstd_encode(d_src, d_dst, d_blocks + b0, m, work, 1, level, zstd_enc_serial(), s)); return FOURMC_OK; })) return r;
        HIP_TRY(fourmc_launch_xxh32(d_dst, d_blocks, n, 0, FOURMC_HASH_DST_RESULT, s));
        return FOURMC_OK;
    }
    if (codec != FOURMC_CODEC_LZ4_FAST) {
        snprintf(g_err, sizeof g_err, "codec %d not implemented on the device yet", codec);
        return FOURMC_EUNSUP;
    }
    if (int r = lz4_fast_encode(d_src, d_dst, d_blocks, n, 1, s)) return r;
    HIP_TRY(fourmc_launch_xxh32(d_dst, d_blocks, n, 0, FOURMC_HASH_DST_RESULT, s));
    return FOURMC_OK;
}

int fourmc_gpu_4mc_decode_blocks(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                 int codec, void* stream)
{
    if (int r = ensure_device()) return r;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (codec == FOURMC_CODEC_ZSTD) {              // .4mz: every level decodes with the same kernel
        HIP_TRY(fourmc_launch_xxh32(d_src, d_blocks, n, 0, FOURMC_VERIFY_SRC, s));
        return in_pieces(s, n, 0, fourmc_zstd_scratch_bytes, [&](uint32_t b0, uint32_t m, void* scratch) -> int {
            HIP_TRY(fourmc_launch_zstd_decode(d_src, d_dst, d_blocks + b0, m, scratch, 1, s)); return FOURMC_OK; });
    }
    if (codec != FOURMC_CODEC_LZ4_FAST && codec != FOURMC_CODEC_LZ4_MC && codec != FOURMC_CODEC_LZ4_HC) {
        snprintf(g_err, sizeof g_err, "codec %d not implemented on the device yet", codec);
        return FOURMC_EUNSUP;
    }
    WsLease ws; void* work = nullptr;
    fourmc_lz4_plan plan;
    if (int r = lease_lz4_decode(ws, s, n, &work, &plan)) return r;
    HIP_TRY(fourmc_launch_xxh32(d_src, d_blocks, n, 0, FOURMC_VERIFY_SRC, s));
    HIP_TRY(fourmc_launch_lz4_decode(d_src, d_dst, d_blocks, n, 1, &plan, work, s));
    return FOURMC_OK;
}

int fourmc_gpu_4mc_pack_image(const void* d_staging, void* d_image, const fourmc_block* d_blocks,
                              const uint64_t* d_image_off, uint32_t n, void* stream)
{
    if (int r = ensure_device()) return r;
    HIP_TRY(fourmc_launch_pack_image(d_staging, d_image, d_blocks, d_image_off, n, static_cast<hipStream_t>(stream)));
    return FOURMC_OK;
}

// ------------------------------------------------------------------------ host-buffer API
int fourmc_LZ4_compressBound(int n) { return (unsigned)n > 0x7E000000u ? 0 : n + n / 255 + 16; }

// d_src / d_dst: the bases the descriptors' offsets are relative to (they may point below an arena's allocation when only
// a slice of the host buffers was staged)
static int launch_host_op(int op, int codec, int level, const void* d_src, void* d_dst, fourmc_block* d_blk, uint32_t n, hipStream_t s)
{
    switch (op) {
        case 0: return fourmc_gpu_4mc_encode_blocks(d_src, d_dst, d_blk, n, codec, level, s);
        case 1: return fourmc_gpu_4mc_decode_blocks(d_src, d_dst, d_blk, n, codec, s);
        case 2: return fourmc_gpu_lz4_compress_fast(d_src, d_dst, d_blk, n, s);
        case 3: return fourmc_gpu_lz4_decompress(d_src, d_dst, d_blk, n, s);
        case 5: return fourmc_gpu_zstd_decompress(d_src, d_dst, d_blk, n, s);
        case 8: return fourmc_gpu_zstd_compress(d_src, d_dst, d_blk, n, level, s);
        case 7: return fourmc_gpu_lz4_compress_mc(d_src, d_dst, d_blk, n, s);
        case 6: return fourmc_gpu_lz4_compress_hc(d_src, d_dst, d_blk, n, level, s);
        default: return fourmc_gpu_xxh32(d_src, d_blk, n, (uint32_t)level, s);
    }
}

// FOURMC_GPUS > 1: contiguous block ranges, one per shard, all in flight together (native/4mc.c:280-333 is the serial
// loop this replaces; the ranges are independent, nothing crosses between devices)
static int host_roundtrip_sharded(const void* src, void* dst, fourmc_block* blocks, uint32_t n, int op, int codec, int level, int shards)
{
    int ndev = fourmc_gpu_device_count();
    if (ndev <= 0) return FOURMC_ENODEV;
    const int dev0 = g_device.load(std::memory_order_acquire);
    const uint32_t per = (n + uint32_t(shards) - 1) / uint32_t(shards);
    struct Range { uint32_t lo, hi; uint64_t smin, smax, dmin, dmax; };
    std::vector<Range> rg;
    for (int k = 0; k < shards; k++) {
        Range r; r.lo = std::min(n, uint32_t(k) * per); r.hi = std::min(n, r.lo + per);
        if (r.lo >= r.hi) break;
        r.smin = r.dmin = ~0ull; r.smax = r.dmax = 0;
        for (uint32_t b = r.lo; b < r.hi; b++) {
            r.smin = std::min<uint64_t>(r.smin, blocks[b].src_off); r.smax = std::max<uint64_t>(r.smax, blocks[b].src_off + blocks[b].src_len);
            r.dmin = std::min<uint64_t>(r.dmin, blocks[b].dst_off); r.dmax = std::max<uint64_t>(r.dmax, blocks[b].dst_off + blocks[b].dst_cap);
        }
        rg.push_back(r);
    }
    int rc = FOURMC_OK;
    for (size_t k = 0; k < rg.size() && rc == FOURMC_OK; k++) {                      // stage + launch every range
        Arena& a = g_arenas[k]; const Range& r = rg[k];
        a.device = (dev0 + int(k)) % ndev;
        // no return from inside this loop: earlier ranges are in flight against the caller's buffers and have to be waited for
        // below, and the thread's device has to be put back
        auto step = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == FOURMC_OK) rc = fail_hip(e, what); return rc == FOURMC_OK; };
        if (!step(hipSetDevice(a.device), "hipSetDevice")) break;
        if ((rc = arena_reserve(a, size_t(r.smax - r.smin), size_t(r.dmax - r.dmin), r.hi - r.lo))) break;
        if (!step(hipMemcpyAsync(a.d_src, static_cast<const char*>(src) + r.smin, size_t(r.smax - r.smin), hipMemcpyHostToDevice, a.stream), "hipMemcpyAsync")) break;
        if (!step(hipMemcpyAsync(a.d_blk, blocks + r.lo, (r.hi - r.lo) * sizeof(fourmc_block), hipMemcpyHostToDevice, a.stream), "hipMemcpyAsync")) break;
        rc = launch_host_op(op, codec, level, static_cast<char*>(a.d_src) - r.smin, static_cast<char*>(a.d_dst) - r.dmin, a.d_blk, r.hi - r.lo, a.stream);
        if (rc) break;
        if (!step(hipMemcpyAsync(blocks + r.lo, a.d_blk, (r.hi - r.lo) * sizeof(fourmc_block), hipMemcpyDeviceToHost, a.stream), "hipMemcpyAsync")) break;
    }
    for (size_t k = 0; k < rg.size(); k++) {                                          // results, then what each block produced
        Arena& a = g_arenas[k]; const Range& r = rg[k];
        if (!a.stream) continue;
        (void)hipSetDevice(a.device);
        hipError_t e = hipStreamSynchronize(a.stream);
        if (e != hipSuccess && rc == FOURMC_OK) rc = fail_hip(e, "hipStreamSynchronize");
        if (rc != FOURMC_OK || op == 4) continue;
        for (uint32_t b = r.lo; b < r.hi; b++)
            if (blocks[b].result > 0) {
                e = hipMemcpyAsync(static_cast<char*>(dst) + blocks[b].dst_off, static_cast<char*>(a.d_dst) + (blocks[b].dst_off - r.dmin),
                                   (size_t)blocks[b].result, hipMemcpyDeviceToHost, a.stream);
                if (e != hipSuccess && rc == FOURMC_OK) rc = fail_hip(e, "hipMemcpyAsync");
            }
    }
    for (size_t k = 0; k < rg.size(); k++) {
        Arena& a = g_arenas[k];
        if (!a.stream) continue;
        (void)hipSetDevice(a.device);
        hipError_t e = hipStreamSynchronize(a.stream);
        if (e != hipSuccess && rc == FOURMC_OK) rc = fail_hip(e, "hipStreamSynchronize");
    }
    (void)hipSetDevice(dev0);
    return rc;
}

static int host_roundtrip_many(const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                               fourmc_block* blocks, uint32_t n, int op, int codec, int level)
{
    if (int r = ensure_device()) return r;
    std::lock_guard<std::mutex> lk(g_mu);
    if (host_shards() > 1 && n > 1) return host_roundtrip_sharded(src, dst, blocks, n, op, codec, level, host_shards());
    if (int r = arena_reserve(g_arena, src_bytes, dst_bytes, n)) return r;
    hipStream_t s = g_arena.stream;
    HIP_TRY(hipMemcpyAsync(g_arena.d_src, src, src_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g_arena.d_blk, blocks, n * sizeof(fourmc_block), hipMemcpyHostToDevice, s));
    if (int r = launch_host_op(op, codec, level, g_arena.d_src, g_arena.d_dst, g_arena.d_blk, n, s)) return r;
    HIP_TRY(hipMemcpyAsync(blocks, g_arena.d_blk, n * sizeof(fourmc_block), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (op != 4) {
        // copy back only what each block produced
        for (uint32_t b = 0; b < n; b++) {
            if (blocks[b].result > 0)
                HIP_TRY(hipMemcpyAsync(static_cast<char*>(dst) + blocks[b].dst_off,
                                       static_cast<char*>(g_arena.d_dst) + blocks[b].dst_off,
                                       (size_t)blocks[b].result, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
    }
    return FOURMC_OK;
}

// ---- one-block calls (the JNI methods, the LZ4_* / ZSTD_* twins).  One call is one block and returns synchronously, so
// a lone caller cannot be batched - but Hadoop runs many compressor objects on many threads.  Calls that arrive while a
// launch is in flight queue up; the caller that finds nobody serving becomes the server and takes everything queued for
// the same operation into ONE staging copy and ONE launch.  No timer: a lone caller pays nothing, concurrency batches itself.
struct OneReq {
    const void* src; size_t src_bytes; void* dst; size_t dst_bytes; fourmc_block* blk;
    int op, codec, level, rc; bool done; OneReq* next;
    char err[256];                      // fourmc_gpu_last_error() of the launch that served the request
};
static std::mutex g_qmu;
static std::condition_variable g_qcv;
static OneReq* g_qhead = nullptr; static OneReq* g_qtail = nullptr;
static bool g_qserving = false;
static unsigned long long g_one_calls = 0, g_one_launches = 0;

static int serve_group(std::vector<OneReq*>& g)
{
    if (int r = ensure_device()) return r;
    std::lock_guard<std::mutex> lk(g_mu);
    const uint32_t n = (uint32_t)g.size();
    std::vector<fourmc_block> blocks(n);
    size_t so = 0, dof = 0;
    for (uint32_t i = 0; i < n; i++) {
        blocks[i] = *g[i]->blk;
        blocks[i].src_off = so; blocks[i].dst_off = dof;
        so += (g[i]->src_bytes + 63) & ~size_t(63); dof += (g[i]->dst_bytes + 63) & ~size_t(63);
    }
    if (int r = arena_reserve(g_arena, so, dof, n)) return r;
    hipStream_t s = g_arena.stream;
    for (uint32_t i = 0; i < n; i++)
        if (g[i]->src_bytes) HIP_TRY(hipMemcpyAsync(static_cast<char*>(g_arena.d_src) + blocks[i].src_off, g[i]->src, g[i]->src_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g_arena.d_blk, blocks.data(), n * sizeof(fourmc_block), hipMemcpyHostToDevice, s));
    if (int r = launch_host_op(g[0]->op, g[0]->codec, g[0]->level, g_arena.d_src, g_arena.d_dst, g_arena.d_blk, n, s)) return r;
    HIP_TRY(hipMemcpyAsync(blocks.data(), g_arena.d_blk, n * sizeof(fourmc_block), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n; i++) {
        g[i]->blk->result = blocks[i].result; g[i]->blk->xxh32 = blocks[i].xxh32;
        if (g[i]->op != 4 && blocks[i].result > 0)
            HIP_TRY(hipMemcpyAsync(g[i]->dst, static_cast<char*>(g_arena.d_dst) + blocks[i].dst_off, (size_t)blocks[i].result, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return FOURMC_OK;
}

static int host_one(const void* src, size_t src_bytes, void* dst, size_t dst_bytes, fourmc_block* blk, int op, int codec, int level)
{
    OneReq me = {src, src_bytes, dst, dst_bytes, blk, op, codec, level, FOURMC_OK, false, nullptr, {0}};
    std::unique_lock<std::mutex> lk(g_qmu);
    g_one_calls++;
    if (g_qtail) g_qtail->next = &me; else g_qhead = &me;
    g_qtail = &me;
    // Whoever finds nobody serving serves ONE group (the oldest request's kind) and hands the role on: a caller never
    // keeps serving other threads' requests after its own has been answered.
    for (;;) {
        if (me.done) { if (me.rc) snprintf(g_err, sizeof g_err, "%s", me.err); return me.rc; }
        if (g_qserving) { g_qcv.wait(lk, [&] { return me.done || !g_qserving; }); continue; }
        g_qserving = true;
        {
        // everything queued for the same operation as the oldest request, in arrival order
        std::vector<OneReq*> grp;
        OneReq* keep_head = nullptr; OneReq* keep_tail = nullptr;
        const OneReq* first = g_qhead;
        for (OneReq* r = g_qhead; r; ) {
            OneReq* nx = r->next; r->next = nullptr;
            if (r->op == first->op && r->codec == first->codec && r->level == first->level && grp.size() < 4096) grp.push_back(r);
            else { if (keep_tail) keep_tail->next = r; else keep_head = r; keep_tail = r; }
            r = nx;
        }
        g_qhead = keep_head; g_qtail = keep_tail;
        g_one_launches++;
        lk.unlock();
        const int rc = serve_group(grp);
        char err[256]; snprintf(err, sizeof err, "%s", g_err);
        lk.lock();
        for (OneReq* r : grp) { r->rc = rc; if (rc) memcpy(r->err, err, sizeof err); r->done = true; }
        }
        g_qserving = false;
        g_qcv.notify_all();
    }
}

static int host_roundtrip(const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                          fourmc_block* blocks, uint32_t n, int op, int codec, int level)
{
    if (n == 1 && op >= 2) return host_one(src, src_bytes, dst, dst_bytes, blocks, op, codec, level);
    return host_roundtrip_many(src, src_bytes, dst, dst_bytes, blocks, n, op, codec, level);
}

void fourmc_gpu_one_block_stats(unsigned long long* calls, unsigned long long* launches)
{
    std::lock_guard<std::mutex> lk(g_qmu);
    if (calls) *calls = g_one_calls;
    if (launches) *launches = g_one_launches;
}
#ifdef FOURMC_RESEARCH
void fourmc_debug_one_block_counters(unsigned long long* calls, unsigned long long* launches) { fourmc_gpu_one_block_stats(calls, launches); }
#endif

/* page-locked host buffers for the file API's staging (H2D / D2H at PCIe rate instead of the pageable path's bounce copies);
 * NULL when no device is usable or the allocation fails - the caller then falls back to malloc() */
void* fourmc_host_alloc(size_t bytes)
{
    if (ensure_device() != FOURMC_OK) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void fourmc_host_free(void* p) { if (p) (void)hipHostFree(p); }

int fourmc_host_4mc_encode(const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                           fourmc_block* blocks, uint32_t n, int codec, int level)
{ return host_roundtrip(src, src_bytes, dst, dst_bytes, blocks, n, 0, codec, level); }

int fourmc_host_4mc_decode(const void* src, size_t src_bytes, void* dst, size_t dst_bytes,
                           fourmc_block* blocks, uint32_t n, int codec)
{ return host_roundtrip(src, src_bytes, dst, dst_bytes, blocks, n, 1, codec, 0); }

// Encode `n` blocks and hand back the finished piece of the .4mc / .4mz file: "12-byte header + payload" per block, back
// to back (native/4mc.c:309-315), laid down in HBM by pack.hip and copied to `image` in ONE transfer - `image` may be the
// file itself (a shared mapping): nothing passes through a staging buffer or stdio.  image_off[b] = where block b's header
// sits in the piece (the footer index is built from these), *image_bytes = its length.
int fourmc_host_4mc_encode_image(const void* src, size_t src_bytes, fourmc_block* blocks, uint32_t n, int codec, int level,
                                 void* image, size_t image_cap, uint64_t* image_off, size_t* image_bytes)
{
    if (int r = ensure_device()) return r;
    if (n == 0) { *image_bytes = 0; return FOURMC_OK; }
    std::lock_guard<std::mutex> lk(g_mu);
    static void* d_img = nullptr; static size_t img_cap = 0; static int img_dev = -1;          // guarded by g_mu, like the arena; keyed by the device
    static uint64_t* d_ioff = nullptr; static size_t ioff_cap = 0; static int ioff_dev = -1;
    size_t dst_bytes = 0;
    for (uint32_t b = 0; b < n; b++) dst_bytes = std::max<size_t>(dst_bytes, blocks[b].dst_off + blocks[b].dst_cap);
    if (int r = arena_reserve(g_arena, src_bytes, dst_bytes, n)) return r;
    hipStream_t s = g_arena.stream;
    // every exit behind the first asynchronous copy waits for the stream: the copies read and write the CALLER's memory (the
    // descriptor array, the mapped files), and the next user of the arena must not meet them still in flight
    int rc = FOURMC_OK; uint64_t pos = 0;
    auto step = [&](hipError_t e, const char* what) { if (rc == FOURMC_OK && e != hipSuccess) rc = fail_hip(e, what); return rc == FOURMC_OK; };
    do {
        if (!step(hipMemcpyAsync(g_arena.d_src, src, src_bytes, hipMemcpyHostToDevice, s), "H2D source")) break;
        if (!step(hipMemcpyAsync(g_arena.d_blk, blocks, n * sizeof(fourmc_block), hipMemcpyHostToDevice, s), "H2D descriptors")) break;
        if (int r = launch_host_op(0, codec, level, g_arena.d_src, g_arena.d_dst, g_arena.d_blk, n, s)) { rc = r; break; }
        if (!step(hipMemcpyAsync(blocks, g_arena.d_blk, n * sizeof(fourmc_block), hipMemcpyDeviceToHost, s), "D2H descriptors")) break;
        if (!step(hipStreamSynchronize(s), "hipStreamSynchronize")) break;
        for (uint32_t b = 0; b < n && rc == FOURMC_OK; b++) {
            if (blocks[b].result <= 0 || uint32_t(blocks[b].result) > blocks[b].src_len) { snprintf(g_err, sizeof g_err, "block %u: encoder returned %d", b, blocks[b].result); rc = FOURMC_EINVAL; break; }
            image_off[b] = pos; pos += 12ull + uint32_t(blocks[b].result);
        }
        if (rc != FOURMC_OK) break;
        if (pos > image_cap) { snprintf(g_err, sizeof g_err, "image capacity %zu below %llu", image_cap, (unsigned long long)pos); rc = FOURMC_EINVAL; break; }
        if (pos > img_cap || img_dev != g_device.load()) {
            if (d_img) step(hipFree(d_img), "hipFree"); d_img = nullptr; img_cap = 0;
            if (!step(hipMalloc(&d_img, pos + pos / 8 + 4096), "hipMalloc image")) break;
            img_cap = pos + pos / 8 + 4096; img_dev = g_device.load();
        }
        if (n > ioff_cap || ioff_dev != g_device.load()) {
            if (d_ioff) step(hipFree(d_ioff), "hipFree"); d_ioff = nullptr; ioff_cap = 0;
            if (!step(hipMalloc(&d_ioff, (size_t(n) + 64) * 8), "hipMalloc offsets")) break;
            ioff_cap = size_t(n) + 64; ioff_dev = g_device.load();
        }
        if (!step(hipMemcpyAsync(d_ioff, image_off, size_t(n) * 8, hipMemcpyHostToDevice, s), "H2D offsets")) break;
        if (!step(fourmc_launch_pack_image(g_arena.d_dst, d_img, g_arena.d_blk, d_ioff, n, s), "pack")) break;
        if (!step(hipMemcpyAsync(image, d_img, pos, hipMemcpyDeviceToHost, s), "D2H image")) break;
    } while (0);
    { const hipError_t e = hipStreamSynchronize(s); if (rc == FOURMC_OK && e != hipSuccess) rc = fail_hip(e, "hipStreamSynchronize"); }
    if (rc != FOURMC_OK) return rc;
    *image_bytes = pos;
    return FOURMC_OK;
}

int fourmc_LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity)
{
    if (srcSize < 0 || dstCapacity < 0) return 0;
    fourmc_block b; memset(&b, 0, sizeof b);
    b.src_len = (uint32_t)srcSize; b.dst_cap = (uint32_t)dstCapacity;
    size_t dst_bytes = (size_t)dstCapacity;
    int r = host_roundtrip(src, (size_t)srcSize, dst, dst_bytes, &b, 1, 2, FOURMC_CODEC_LZ4_FAST, 0);
    if (r) { fprintf(stderr, "4mc-gpu: %s\n", g_err); return 0; }
    return b.result;
}

static int host_mc(const char* src, char* dst, int srcSize, uint32_t cap_field, size_t dst_bytes)
{
    if (srcSize < 0) return 0;
    fourmc_block b; memset(&b, 0, sizeof b);
    b.src_len = (uint32_t)srcSize; b.dst_cap = cap_field;
    int r = host_roundtrip(src, (size_t)srcSize, dst, dst_bytes, &b, 1, 7, FOURMC_CODEC_LZ4_MC, 0);
    if (r) { fprintf(stderr, "4mc-gpu: %s\n", g_err); return 0; }
    return b.result;
}

/* LZ4_compressMC writes without an output limit: dst must hold LZ4_compressBound(srcSize) (lz4mc.h) */
int fourmc_LZ4_compressMC(const char* src, char* dst, int srcSize)
{ return host_mc(src, dst, srcSize, 0xFFFFFFFFu, (size_t)fourmc_LZ4_compressBound(srcSize)); }

int fourmc_LZ4_compressMC_limitedOutput(const char* src, char* dst, int srcSize, int maxOutputSize)
{ return maxOutputSize < 0 ? 0 : host_mc(src, dst, srcSize, (uint32_t)maxOutputSize, (size_t)maxOutputSize); }

int fourmc_LZ4_compress_HC(const char* src, char* dst, int srcSize, int dstCapacity, int compressionLevel)
{
    if (srcSize < 0 || dstCapacity < 0) return 0;
    fourmc_block b; memset(&b, 0, sizeof b);
    b.src_len = (uint32_t)srcSize; b.dst_cap = (uint32_t)dstCapacity;
    int r = host_roundtrip(src, (size_t)srcSize, dst, (size_t)dstCapacity, &b, 1, 6, FOURMC_CODEC_LZ4_HC, compressionLevel);
    if (r) { fprintf(stderr, "4mc-gpu: %s\n", g_err); return 0; }
    return b.result;
}

int fourmc_LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity)
{
    if (compressedSize < 0 || dstCapacity < 0) return -1;
    fourmc_block b; memset(&b, 0, sizeof b);
    b.src_len = (uint32_t)compressedSize; b.dst_cap = (uint32_t)dstCapacity;
    int r = host_roundtrip(src, (size_t)compressedSize, dst, (size_t)dstCapacity, &b, 1, 3, FOURMC_CODEC_LZ4_FAST, 0);
    if (r) { fprintf(stderr, "4mc-gpu: %s\n", g_err); return -1; }
    return b.result;
}

size_t fourmc_ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize)
{
    const size_t kCorrupt = (size_t)-20;           /* ZSTD_error_corruption_detected: ZSTD_isError() is true */
    if (compressedSize > 0x7FFFFFFFu || dstCapacity > 0x7FFFFFFFu) return (size_t)-72;   /* srcSize_wrong */
    fourmc_block b; memset(&b, 0, sizeof b);
    b.src_len = (uint32_t)compressedSize; b.dst_cap = (uint32_t)dstCapacity;
    int r = host_roundtrip(src, compressedSize, dst, dstCapacity, &b, 1, 5, FOURMC_CODEC_ZSTD, 0);
    if (r) { fprintf(stderr, "4mc-gpu: %s\n", g_err); return (size_t)-1; }
    return b.result < 0 ? kCorrupt : (size_t)b.result;
}

size_t fourmc_ZSTD_compressBound(size_t n)
{ return n + (n >> 8) + (n < (128u << 10) ? ((128u << 10) - n) >> 11 : 0); }

size_t fourmc_ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel)
{
    if (srcSize > 0x7FFFFFFFu) return (size_t)-72;                           /* srcSize_wrong */
    if (dstCapacity > 0x7FFFFFFFu) dstCapacity = 0x7FFFFFFFu;                /* Java passes 1 GiB (jniZstdCompressor.c:93) */
    const size_t bound = fourmc_ZSTD_compressBound(srcSize);
    const size_t cap = dstCapacity < bound ? dstCapacity : bound;            /* the frame never exceeds the bound */
    fourmc_block b; memset(&b, 0, sizeof b);
    b.src_len = (uint32_t)srcSize; b.dst_cap = (uint32_t)cap;
    int r = host_roundtrip(src, srcSize, dst, cap, &b, 1, 8, FOURMC_CODEC_ZSTD, compressionLevel);
    if (r) { fprintf(stderr, "4mc-gpu: %s\n", g_err); return (size_t)-1; }
    return b.result < 0 ? (size_t)(ptrdiff_t)b.result : (size_t)b.result;  /* -(error number), ZSTD_isError() is true */
}

} // extern "C"
