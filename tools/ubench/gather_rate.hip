// Microbenchmark: what a FULL chip of one-wave workgroups gets out of the memory system for the load shapes of the LZ77 kernels
// (K1s executor: literals and far matches; K2p: candidate gathers).  Every wave owns a window of W bytes (its block's last 64 KiB, or
// its segment) and issues N independent loads per s_waitcnt; reported: wave-load-instructions per microsecond chip-wide, useful
// bytes per second, for
//   seq16   lane l reads 16 bytes at 16 l            (a coalesced 1 KiB)
//   near16  lane l reads 16 bytes at ~9 l + jitter   (literal strings of consecutive sequences: unaligned, neighbours share lines)
//   rand16  lane l reads 16 unaligned bytes at a pseudo-random place of the window (far matches, candidates)
//   rand4   the same, 4 bytes
//   rand32  two 16-byte loads at a random place (a 32-byte string)
// with the windows of all waves together either small (32 MiB: they live in the L2s) or as large as in the real launches
// (waves x 64 KiB = 512 MiB at 8192 waves: beyond the MALL).
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t x; };

template <int MODE>
__global__ __launch_bounds__(64) void k(const uint8_t* base, uint32_t wbytes, uint32_t wstride, int iters, uint32_t* sink)
{
    const int lane = threadIdx.x;
    const uint8_t* w = base + size_t(blockIdx.x) * wstride;
    uint32_t acc = 0, r = blockIdx.x * 2654435761u + lane * 40503u + 12345u;
    const uint32_t mask = wbytes - 64u;                   // wbytes is a power of two
    for (int i = 0; i < iters; i++) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            r = r * 1664525u + 1013904223u;
            const uint32_t rnd = (r >> 8) & mask;
            // a moving origin common to the wave (the cursor), so that seq / near patterns walk through the window
            const uint32_t org = (uint32_t(i * 8 + j) * 1024u) & mask;
            uint32_t a;
            if (MODE == 0) a = (org + 16u * lane) & mask;
            else if (MODE == 1) a = (org + 9u * lane + ((r >> 28) & 3u)) & mask;
            else a = rnd;
            if (MODE == 3) v[j] = reinterpret_cast<const U4B*>(w + a)->x;
            else if (MODE == 4) { const U16B t = *reinterpret_cast<const U16B*>(w + a); const U16B u = *reinterpret_cast<const U16B*>(w + a + 16); v[j] = t.x ^ t.w ^ u.y ^ u.w; }
            else { const U16B t = *reinterpret_cast<const U16B*>(w + a); v[j] = t.x ^ t.y ^ t.z ^ t.w; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main()
{
    const int iters = 400;
    const uint32_t wbytes = 65536;
    uint8_t* buf; uint32_t* sink;
    const size_t total = size_t(8192) * wbytes + 4096;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 8192 * 4));
    CK(hipMemset(buf, 7, total));
    const char* names[5] = {"seq16 ", "near16", "rand16", "rand4 ", "rand32"};
    const int bytes_per_lane[5] = {16, 16, 16, 4, 32};
    const int instr_per_load[5] = {1, 1, 1, 1, 2};
    printf("waves  windows        pattern   ms      wave-loads/us   useful GB/s   lane-loads/us per CU\n");
    for (int waves : {2048, 8192}) for (int spread = 0; spread < 2; spread++) {
        // spread 0: all waves' windows inside 32 MiB (stride 4 KiB: overlapping windows, L2-resident); 1: a window of its own per wave
        const uint32_t stride = spread ? wbytes : 4096u;
        for (int mode = 0; mode < 5; mode++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, buf, wbytes, stride, iters, sink);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, buf, wbytes, stride, iters, sink);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, buf, wbytes, stride, iters, sink);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, buf, wbytes, stride, iters, sink);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, buf, wbytes, stride, iters, sink);
                hipEventRecord(b); CK(hipDeviceSynchronize());
                hipEventElapsedTime(&ms, a, b);
            }
            const double loads = double(waves) * iters * 8 * instr_per_load[mode];
            printf("%5d  %-13s  %s  %7.3f  %10.1f  %12.1f  %10.2f\n", waves, spread ? "own 64 KiB" : "in 32 MiB", names[mode], ms,
                   loads / (ms * 1e3), double(waves) * iters * 8 * 64 * bytes_per_lane[mode] / (ms * 1e6), loads * 64 / (ms * 1e3) / 256);
        }
    }
    return 0;
}
