// Microbenchmark (round 6): what do the LDS operations of the group executor (lz4_ring.hip) cost per wave-instruction, as a function of the
// ACTIVE LANES and of how many lanes share a dword?  ds_or_b32 (no return) against ds_write_b32 / ds_write_b8 / ds_read_b32, one wave
// and 16 waves per CU; and what one s_barrier round trip of an 8-wave workgroup costs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic.hip -o tools/ubench/lds_atomic && tools/ubench/lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
enum { OR32, WR32, WR8, RD32, OR32X4 };
// active: lanes below `active` take part; share: `share` consecutive lanes address the same dword
template <int OP>
__global__ void k(long long* cyc, uint32_t* sink, int active, int share, int iters)
{
    __shared__ uint32_t lds[16384];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const uint32_t a = ((wave * 1024u + (lane / uint32_t(share)) * 1u) & 16383u) * 4u + (OP == WR8 ? (lane % uint32_t(share)) & 3u : 0u);
    uint32_t acc = lane;
    const bool on = int(lane) < active;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t aa = a + uint32_t(u) * 256u;
                if (OP == OR32) asm volatile("ds_or_b32 %0, %1" :: "v"(aa), "v"(acc) : "memory");
                if (OP == WR32) asm volatile("ds_write_b32 %0, %1" :: "v"(aa), "v"(acc) : "memory");
                if (OP == WR8)  asm volatile("ds_write_b8 %0, %1" :: "v"(aa), "v"(acc) : "memory");
                if (OP == RD32) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(aa) : "memory"); acc += v & 1u; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0xdeadbeef) sink[0] = acc;
}
__global__ void kbar(long long* cyc, int iters)
{
    __shared__ uint32_t flag[4];
    if (threadIdx.x == 0) flag[0] = 0;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        if ((threadIdx.x & 63) == 0) flag[i & 3] = i;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        acc += flag[i & 3];
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0 + (acc == 1 ? 1 : 0);
}
template <int OP> static int run(const char* name, long long* cyc, uint32_t* sink)
{
    const int iters = 2000;
    for (int waves : {1, 8}) for (int blocks : {1, 512}) {
        printf("%-13s %d wave(s)/WG, %3d WG:", name, waves, blocks);
        for (int share : {1, 2, 4, 64}) for (int active : {1, 4, 16, 64}) {
            if (share == 64 && active < 64) continue;
            if (share != 1 && active != 64) continue;
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves), 0, 0, cyc, sink, active, share, iters);
            CK(hipDeviceSynchronize());
            long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("  a%d/s%d %.1f", active, share, double(h) / (iters * 8.0));
        }
        printf("   clk per wave-instruction (wave 0's clock)\n");
    }
    return 0;
}
int main()
{
    long long* cyc; uint32_t* sink;
    CK(hipMalloc(&cyc, 8 * 512)); CK(hipMalloc(&sink, 64));
    if (run<OR32>("ds_or_b32", cyc, sink) || run<WR32>("ds_write_b32", cyc, sink) || run<WR8>("ds_write_b8", cyc, sink) || run<RD32>("ds_read_b32", cyc, sink)) return 1;
    for (int blocks : {1, 512}) {
        hipLaunchKernelGGL(kbar, dim3(blocks), dim3(512), 0, 0, cyc, 4000);
        CK(hipDeviceSynchronize());
        long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("barrier round (8 waves: flag write, s_barrier, flag read), %d WG: %.1f clk\n", blocks, double(h) / 4000.0);
    }
    return 0;
}
