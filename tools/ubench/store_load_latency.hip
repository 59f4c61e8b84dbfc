// Microbenchmark: how long does a dependent global load take when byte stores are outstanding?
// (gfx9 has ONE vmcnt for loads and stores; s_waitcnt for a load also drains older stores.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k(uint8_t* out, const uint32_t* chase, uint32_t* res, int iters, long long* cyc)
{
    const int lane = threadIdx.x;
    uint8_t* o = out + size_t(blockIdx.x) * (8u << 20);
    uint32_t idx = 0, acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 1) { if (lane < 8) o[size_t(i) * 9 + lane] = uint8_t(acc + lane); }            // byte stores marching through fresh lines
        if (MODE == 2) { if (lane < 8) o[(i & 7) * 8 + lane] = uint8_t(acc + lane); }                // byte stores to one hot line
        if (MODE == 3) { if ((i & 63) == 63) reinterpret_cast<uint4*>(o)[size_t(i >> 6) * 64 + lane] = make_uint4(acc, i, 0, 0); }   // 1 KiB full-line store every 64 iterations
        idx = chase[(idx + lane * 0) & 0xFFFF];                                                       // dependent L2-resident load (uniform)
        acc += idx;
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { res[blockIdx.x] = acc; cyc[blockIdx.x] = t1 - t0; }
}

int main()
{
    uint8_t* out; uint32_t *chase, *res; long long* cyc;
    const int nb = 2048, iters = 20000;
    CK(hipMalloc(&out, size_t(nb) * (8u << 20) > (size_t(16) << 30) ? (size_t(16) << 30) : size_t(nb) * (8u << 20)));
    CK(hipMalloc(&chase, 65536 * 4)); CK(hipMalloc(&res, nb * 4)); CK(hipMalloc(&cyc, nb * 8));
    uint32_t* h = (uint32_t*)malloc(65536 * 4);
    for (int i = 0; i < 65536; i++) h[i] = (uint32_t)((i * 40503u + 17) & 0xFFFF);
    CK(hipMemcpy(chase, h, 65536 * 4, hipMemcpyHostToDevice));
    const char* names[4] = {"load only", "8 byte-stores (fresh lines) + load", "8 byte-stores (hot line) + load", "1 KiB store / 64 iters + load"};
    for (int blocks : {1, 2048}) for (int mode = 0; mode < 4; mode++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, chase, res, iters, cyc);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, chase, res, iters, cyc);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, chase, res, iters, cyc);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, chase, res, iters, cyc);
            hipEventRecord(b); CK(hipDeviceSynchronize());
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        long long c0; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
        printf("blocks %4d  %-40s : %8.3f ms  = %7.1f ns / iteration, %lld clk/iter (s_memtime)\n", blocks, names[mode], ms, ms * 1e6 / iters, c0 / iters);
    }
    return 0;
}
