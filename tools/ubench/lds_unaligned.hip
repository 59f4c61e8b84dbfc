// Microbenchmark: cost and correctness of byte-unaligned LDS accesses on gfx950 (the LZ77 copy primitive of the
// LDS-window decoders).  For each instruction and each misalignment 0..7: checks the data and reports cycles per
// wave-instruction (16 independent ops per s_waitcnt, W waves per workgroup, one workgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { RD32, RD64, WR32, WR64, RD8, WR8, RD16, WR16, RD128, NOPS };
static const char* kNames[NOPS] = {"ds_read_b32", "ds_read_b64", "ds_write_b32", "ds_write_b64", "ds_read_u8", "ds_write_b8", "ds_read_u16", "ds_write_b16", "ds_read_b128"};

template <int OP>
__global__ void k(int mis, int stride, int iters, long long* cyc, uint32_t* bad)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t buf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536; i += blockDim.x) buf[i] = uint8_t(i * 31 + (i >> 8));
    __syncthreads();
    // pseudo-random but conflict-light addresses: lane * stride + mis inside the wave's own 4 KiB
    const uint32_t a = uint32_t(wave) * 4096u + uint32_t(lane) * uint32_t(stride) + uint32_t(mis);
    uint32_t nbad = 0;
    // correctness
    if (OP == RD32) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        uint32_t e = 0; for (int j = 0; j < 4; j++) e |= uint32_t(uint8_t((a + j) * 31 + ((a + j) >> 8))) << (8 * j); nbad += v != e; }
    if (OP == RD64) { uint64_t v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        uint64_t e = 0; for (int j = 0; j < 8; j++) e |= uint64_t(uint8_t((a + j) * 31 + ((a + j) >> 8))) << (8 * j); nbad += v != e; }
    if (OP == WR32 || OP == WR64 || OP == WR8 || OP == WR16) {
        const uint64_t val = 0x1122334455667788ull ^ (uint64_t(lane) * 0x0101010101010101ull);
        const int nbytes = OP == WR32 ? 4 : OP == WR64 ? 8 : OP == WR8 ? 1 : 2;
        const uint32_t w = a + 32768u;    // upper half: written
        // neighbours must stay intact
        uint8_t before = buf[w - 1], after = buf[w + nbytes];
        if (OP == WR32) asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(w), "v"(uint32_t(val)) : "memory");
        if (OP == WR64) asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(w), "v"(val) : "memory");
        if (OP == WR8)  asm volatile("ds_write_b8 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(w), "v"(uint32_t(val)) : "memory");
        if (OP == WR16) asm volatile("ds_write_b16 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(w), "v"(uint32_t(val)) : "memory");
        for (int j = 0; j < nbytes; j++) nbad += buf[w + j] != uint8_t(val >> (8 * j));
        if (stride > nbytes + 1) nbad += (buf[w - 1] != before) + (buf[w + nbytes] != after);
    }
    __syncthreads();
    uint32_t acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint32_t aa = a + ((u & 3) * 1024u);
            if (OP == RD32) { uint32_t v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(aa) : "memory"); acc += v; }
            if (OP == RD8)  { uint32_t v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(aa) : "memory"); acc += v; }
            if (OP == RD16) { uint32_t v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(aa) : "memory"); acc += v; }
            if (OP == RD64) { uint64_t v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(aa) : "memory"); acc += uint32_t(v); }
            if (OP == RD128) { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(aa & ~15u) : "memory"); acc += v.x; }
            if (OP == WR32) asm volatile("ds_write_b32 %0, %1" :: "v"(aa + 32768u), "v"(acc) : "memory");
            if (OP == WR64) asm volatile("ds_write_b64 %0, %1" :: "v"(aa + 32768u), "v"(uint64_t(acc)) : "memory");
            if (OP == WR8)  asm volatile("ds_write_b8 %0, %1" :: "v"(aa + 32768u), "v"(acc) : "memory");
            if (OP == WR16) asm volatile("ds_write_b16 %0, %1" :: "v"(aa + 32768u), "v"(acc) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = t1 - t0;
    if (nbad || acc == 0x12345) atomicAdd(bad, nbad);
}

// dependent chain: read -> (address) -> read ...; and write -> read of the same bytes (store-to-load through the LDS)
__global__ void chain(int mis, int iters, long long* cyc, uint32_t* sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[8192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) buf[i] = 0;
    __syncthreads();
    uint32_t a = uint32_t(lane) * 24u + uint32_t(mis);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        uint32_t v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        a += v;     // v == 0: stays, but dependent
    }
    long long t1 = __builtin_readcyclecounter();
    uint64_t x = lane;
    for (int i = 0; i < iters; i++) {     // LZ77 hop: read 8 bytes, write them 100 bytes further, next read depends on that write's data
        uint64_t v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        asm volatile("ds_write_b64 %0, %1" :: "v"(a + 2048u + uint32_t(v & 0)), "v"(v + x) : "memory");
        a = (a + 2048u + uint32_t(v & 0)) & 4095u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; sink[0] = a; }
}

template <int OP> int run(long long* cyc, uint32_t* bad)
{
    const int iters = 2000;
    for (int stride : {12, 36}) for (int waves : {1, 4, 16}) {
        printf("%-13s stride %2d waves %2d :", kNames[OP], stride, waves);
        for (int mis = 0; mis < 8; mis++) {
            CK(hipMemset(bad, 0, 4));
            hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 65536, 0, mis, stride, iters, cyc, bad);
            CK(hipDeviceSynchronize());
            long long c[16]; uint32_t nb; CK(hipMemcpy(c, cyc, 8 * waves, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
            long long mx = 0; for (int w = 0; w < waves; w++) mx = c[w] > mx ? c[w] : mx;
            // cycles per wave-instruction as seen by the CU: total instructions = waves * iters * 16
            printf(" %6.1f%s", double(mx) / (double(iters) * 16 * waves), nb ? "!" : "");
        }
        printf("   clk / wave-instr (CU view), mis 0..7; '!' = wrong data\n");
    }
    return 0;
}

int main()
{
    long long* cyc; uint32_t* bad;
    CK(hipMalloc(&cyc, 8 * 64)); CK(hipMalloc(&bad, 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<RD32>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    run<RD32>(cyc, bad); run<RD64>(cyc, bad); run<RD128>(cyc, bad); run<RD8>(cyc, bad); run<RD16>(cyc, bad);
    run<WR32>(cyc, bad); run<WR64>(cyc, bad); run<WR8>(cyc, bad); run<WR16>(cyc, bad);
    for (int mis : {0, 1, 4}) {
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, mis, 4000, cyc, bad);
        CK(hipDeviceSynchronize());
        long long c[2]; CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
        printf("chain mis %d: dependent ds_read_b32 %.1f clk; read_b64 -> write_b64 -> dependent read hop %.1f clk\n", mis, c[0] / 4000.0, c[1] / 4000.0);
    }
    return 0;
}
