// 4mc_amd/csrc/lz4mc_encode.hip — K4: batched 4mc "Medium" LZ4 block encode on gfx950,
// BYTE-IDENTICAL to LZ4_compressMC / LZ4_compressMC_limitedOutput of the reference
// (4mc's own encoder, native/lz4/lz4mc.c; reached by `4mc -2`, native/4mc.c:246-247, and by
// Lz4Compressor.compressBytesDirectMC, native/jniCompressor.c:124).
//
//   parse    LZ4MC_compress_generic          lz4mc.c:518-579  (greedy, probe stride grows on misses)
//   search   LZ4MC_InsertAndFindBestMatch    lz4mc.c:435-462  (hash chain, 4 attempts)
//   tables   LZ4MC_Insert                    lz4mc.c:387-404  (first pending + last skipped position)
//   emit     LZ4MC_encodeSequence            lz4mc.c:466-505
//
// One wavefront per block; hash heads (32768 x u32) and chain (65536 x u16) sit in the block's
// HBM workspace slot (same layout as the HC kernel).  Per probe the two table insertions and the
// head lookup are issued together (bucket collisions between them are resolved in registers),
// the <= 4 chain candidates are measured one per lane, long matches are finished by the whole
// wavefront, emission is wave-wide.  The reference's one-byte pre-filter at `ml` cannot reject a
// winning candidate, so "longest, earliest wins ties" reproduces its choice exactly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fourmc_gpu.h"
#include "kernels.h"
#include "devenc.h"

namespace {

constexpr int      kHashLog = 15;
constexpr uint32_t kMaxDist = 65535;
constexpr int      kMfLimit = 12, kLastLit = 5, kAttempts = 4;
constexpr size_t   kWorkBytes = (size_t(4) << kHashLog) + 2 * 65536;

__device__ __forceinline__ uint32_t mc_hash(uint32_t v) { return (v * 2654435761u) >> (32 - kHashLog); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }

__device__ __forceinline__ uint32_t wave_count_fwd(const uint8_t* s, uint32_t a, uint32_t b, uint32_t lim, int lane)
{
    uint32_t n = 0;
    for (;;) {
        if (a + 1024 <= lim) {
            const U16B x = *reinterpret_cast<const U16B*>(s + a + 16 * lane);
            const U16B y = *reinterpret_cast<const U16B*>(s + b + 16 * lane);
            const uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
            const uint32_t eq = d0 ? uint32_t(__builtin_ctzll(d0) >> 3) : (d1 ? 8u + uint32_t(__builtin_ctzll(d1) >> 3) : 16u);
            const unsigned long long bad = __ballot(eq < 16);
            if (bad) { const int l = __builtin_ctzll(bad); return n + 16 * l + __builtin_amdgcn_readlane(eq, l); }
            n += 1024; a += 1024; b += 1024;
        } else {
            const uint32_t i = a + lane;
            const bool same = (i < lim) && s[i] == s[b + lane];
            const unsigned long long bad = ~__ballot(same);
            if (bad) return n + __builtin_ctzll(bad);
            n += 64; a += 64; b += 64;
        }
    }
}

// cap < 0: no limit (LZ4_compressMC), else LZ4_compressMC_limitedOutput
__device__ int lz4mc_encode_block(const uint8_t* src, uint8_t* dst, int n, int cap, uint8_t* work, int lane)
{
    uint32_t* heads = reinterpret_cast<uint32_t*>(work);
    uint16_t* chain = reinterpret_cast<uint16_t*>(work + (size_t(4) << kHashLog));
    {   // LZ4_initMC: heads = 0, chain = 0xFFFF
        uint4* w = reinterpret_cast<uint4*>(work);
        const uint32_t nh = uint32_t((size_t(4) << kHashLog) / 16), nt = uint32_t(kWorkBytes / 16);
        for (uint32_t i = lane; i < nt; i += 64) w[i] = i < nh ? make_uint4(0, 0, 0, 0) : make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    const bool limited = cap >= 0;
    const uint32_t ucap = limited ? uint32_t(cap) : 0u;
    const int mflimit = n - kMfLimit;
    const uint32_t matchlimit = uint32_t(n > kLastLit ? n - kLastLit : 0);
    uint32_t ip = 1, anchor = 0, ntu = 1, op = 0, tries = 64, step = 1;

    while (int(ip) < mflimit) {
        // ---- insert (at most two positions) + head lookup, one round trip
        const bool ins0 = ntu < ip, ins1 = ntu + 1 < ip;               // positions ntu and ip-1
        const uint32_t p0 = ntu, p1 = ip - 1;
        const uint32_t h0 = ins0 ? mc_hash(ld4(src + p0)) : 0xFFFFFFFFu;
        const uint32_t h1 = ins1 ? mc_hash(ld4(src + p1)) : 0xFFFFFFFEu;
        const uint32_t word = ld4(src + ip);
        const uint32_t h = mc_hash(word);
        const uint32_t old0 = ins0 ? heads[h0] : 0u, old1 = ins1 ? heads[h1] : 0u, oldh = heads[h];
        if (ins0) {
            uint32_t d = p0 - old0; if (d > kMaxDist) d = kMaxDist;
            if (lane == 0) { chain[p0 & 0xFFFF] = uint16_t(d); heads[h0] = p0; }
        }
        if (ins1) {
            uint32_t d = p1 - (h1 == h0 ? p0 : old1); if (d > kMaxDist) d = kMaxDist;
            if (lane == 0) { chain[p1 & 0xFFFF] = uint16_t(d); heads[h1] = p1; }
        }
        if (ins0) ntu = ip;
        int ref = int(uni(h == h1 ? p1 : (h == h0 ? p0 : oldh)));
        // ---- walk the chain (<= 4 candidates), one per lane
        int cand = 0, nc = 0;
        while (uint32_t(int(ip) - ref) <= kMaxDist && nc < kAttempts) {
            if (lane == nc) cand = ref;
            nc++;
            ref -= int(uni(uint32_t(chain[uint32_t(ref) & 0xFFFF])));
        }
        const bool live = lane < nc && ld4(src + (lane < nc ? cand : 0)) == word;
        uint32_t fl = 0; bool more = false;
        if (live) {
            uint32_t a = ip + 4, b = uint32_t(cand) + 4;
            more = true;
            for (int it = 0; it < 4; it++) {
                if (a + 8 > matchlimit) { while (a < matchlimit && src[a] == src[b]) { a++; b++; fl++; } more = false; break; }
                const uint64_t x = ld8(src + a) ^ ld8(src + b);
                if (x) { fl += uint32_t(__builtin_ctzll(x) >> 3); more = false; break; }
                a += 8; b += 8; fl += 8;
            }
        }
        for (unsigned long long todo = __ballot(more); todo; todo &= todo - 1) {
            const int l = __builtin_ctzll(todo);
            const uint32_t mm = uint32_t(__builtin_amdgcn_readlane(cand, l));
            const uint32_t extra = wave_count_fwd(src, ip + 4 + 32, mm + 4 + 32, matchlimit, lane);
            if (lane == l) fl += extra;
        }
        uint32_t key = live ? (((4 + fl) << 6) | uint32_t(63 - lane)) : 0u;
        // lanes 0..3 hold the candidates: maximum by two DPP row shifts, read from lane 3
        key = max(key, uint32_t(__builtin_amdgcn_update_dpp(0, int(key), 0x111, 0xf, 0xf, false)));
        key = max(key, uint32_t(__builtin_amdgcn_update_dpp(0, int(key), 0x112, 0xf, 0xf, false)));
        key = uint32_t(__builtin_amdgcn_readlane(int(key), 3));
        if (!key) { ip += step; step = tries++ >> 6; continue; }
        const uint32_t ml = key >> 6;
        const uint32_t best = uint32_t(__builtin_amdgcn_readlane(cand, 63 - int(key & 63)));
        {   // ---- LZ4MC_encodeSequence
            uint32_t len = ip - anchor;
            const uint32_t token_pos = op++;
            if (limited && op + len + (2 + 1 + kLastLit) + (len >> 8) > ucap) return 0;
            uint32_t tok;
            if (len >= 15) { tok = 0xF0; op += emit_len(dst + op, len - 15, lane); } else tok = len << 4;
            copy_bytes(dst + op, src + anchor, len, lane);
            op += len;
            const uint32_t off = ip - best;
            if (lane == 0) { dst[op] = uint8_t(off); dst[op + 1] = uint8_t(off >> 8); }
            op += 2;
            len = ml - 4;
            if (limited && op + (1 + kLastLit) + (len >> 8) > ucap) return 0;
            if (len >= 15) { tok += 15; op += emit_len(dst + op, len - 15, lane); } else tok += len;
            if (lane == 0) dst[token_pos] = uint8_t(tok);
            ip += ml; anchor = ip;
        }
        step = 1; tries = 64;
    }
    {   // ---- last literals (lz4mc.c:565-573)
        const uint32_t run = uint32_t(n) - anchor;
        if (limited && op + run + 1 + (run + 255 - 15) / 255 > ucap) return 0;
        if (run >= 15) { if (lane == 0) dst[op] = 0xF0; op++; op += emit_len(dst + op, run - 15, lane); }
        else { if (lane == 0) dst[op] = uint8_t(run << 4); op++; }
        copy_bytes(dst + op, src + anchor, run, lane);
        op += run;
    }
    return int(op);
}

// container_mode 0: dst_cap == 0xFFFFFFFF selects LZ4_compressMC (no limit), else _limitedOutput(dst_cap);
// container_mode 1: native/4mc.c:246-247,:301 (limitedOutput with n-1, stored fallback)
__global__ __launch_bounds__(64)
void lz4mc_encode_kernel(const uint8_t* __restrict__ src_base, uint8_t* dst_base, fourmc_block* blocks,
                         uint32_t nblocks, uint8_t* work_base, int container_mode)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const fourmc_block blk = uniform_block(blocks[b]);
    const uint8_t* src = src_base + blk.src_off;
    uint8_t* dst = dst_base + blk.dst_off;
    const int n = int(blk.src_len);
    const int cap = container_mode ? n - 1 : (blk.dst_cap == 0xFFFFFFFFu ? -1 : int(blk.dst_cap));
    int r = lz4mc_encode_block(src, dst, n, cap, work_base + size_t(b) * kWorkBytes, threadIdx.x);
    if (container_mode && r <= 0) { copy_bytes(dst, src, uint32_t(n), threadIdx.x); r = n; }
    if (threadIdx.x == 0) blocks[b].result = r;
}

} // namespace

extern "C" hipError_t fourmc_launch_lz4mc_encode(const void* d_src, void* d_dst, fourmc_block* d_blocks, uint32_t n,
                                                 void* d_work, int container_mode, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(lz4mc_encode_kernel, dim3(n), dim3(64), 0, stream,
                       static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), d_blocks, n,
                       static_cast<uint8_t*>(d_work), container_mode);
    return hipGetLastError();
}
