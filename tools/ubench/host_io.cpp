// tools/ubench/host_io.cpp - what the file API's host side costs on the GPU box: page-locked allocation, H2D / D2H from and
// into pageable and file-mapped memory, single- and multi-threaded tmpfs writes.  hipcc -O2 host_io.cpp -o host_io -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, bytes, ...) do { double t0 = now(); __VA_ARGS__; double t = now() - t0; printf("%-46s %8.3f s  %7.2f GB/s\n", label, t, (bytes) / t / 1e9); fflush(stdout); } while (0)
int main(int argc, char** argv)
{
    const size_t G = size_t(1) << 30;
    const char* dir = argc > 1 ? argv[1] : "/dev/shm";
    char path[256]; snprintf(path, sizeof path, "%s/host_io.bin", dir);
    double t0 = now(); hipFree(0); printf("%-46s %8.3f s\n", "hipInit (first call)", now() - t0);
    void* d; hipMalloc(&d, 2 * G);
    void* pinned = nullptr; T("hipHostMalloc 1 GiB", G, hipHostMalloc(&pinned, G, hipHostMallocDefault));
    T("  touch pinned (memset)", G, memset(pinned, 1, G));
    T("H2D 1 GiB from pinned", G, hipMemcpy(d, pinned, G, hipMemcpyHostToDevice));
    T("D2H 1 GiB into pinned", G, hipMemcpy(pinned, d, G, hipMemcpyDeviceToHost));
    char* pg = (char*)malloc(G); T("  touch malloc (memset)", G, memset(pg, 1, G));
    T("H2D 1 GiB from pageable", G, hipMemcpy(d, pg, G, hipMemcpyHostToDevice));
    T("D2H 1 GiB into pageable", G, hipMemcpy(pg, d, G, hipMemcpyDeviceToHost));
    T("hipHostRegister 1 GiB (pageable, touched)", G, hipHostRegister(pg, G, hipHostRegisterDefault));
    T("H2D 1 GiB from registered", G, hipMemcpy(d, pg, G, hipMemcpyHostToDevice));
    T("hipHostUnregister", G, hipHostUnregister(pg));
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    T("fwrite-like write() 1 GiB to file", G, { size_t o = 0; while (o < G) { ssize_t w = write(fd, pg + o, G - o < (64u << 20) ? G - o : (64u << 20)); if (w <= 0) break; o += w; } });
    T("read() 1 GiB from file", G, { lseek(fd, 0, SEEK_SET); size_t o = 0; while (o < G) { ssize_t r = read(fd, pg + o, G - o); if (r <= 0) break; o += r; } });
    for (int nt : {2, 4, 8}) {
        char lbl[64]; snprintf(lbl, sizeof lbl, "pwrite 1 GiB, %d threads (rewrite)", nt);
        T(lbl, G, { std::vector<std::thread> th; for (int k = 0; k < nt; k++) th.emplace_back([&, k] { size_t lo = G / nt * k, hi = G / nt * (k + 1); while (lo < hi) { ssize_t w = pwrite(fd, pg + lo, hi - lo < (32u << 20) ? hi - lo : (32u << 20), lo); if (w <= 0) break; lo += w; } }); for (auto& t : th) t.join(); });
    }
    ftruncate(fd, 0);
    for (int nt : {4}) {
        T("pwrite 1 GiB, 4 threads (fresh file)", G, { std::vector<std::thread> th; for (int k = 0; k < nt; k++) th.emplace_back([&, k] { size_t lo = G / nt * k, hi = G / nt * (k + 1); while (lo < hi) { ssize_t w = pwrite(fd, pg + lo, hi - lo < (32u << 20) ? hi - lo : (32u << 20), lo); if (w <= 0) break; lo += w; } }); for (auto& t : th) t.join(); });
    }
    char* mi = (char*)mmap(nullptr, G, PROT_READ, MAP_PRIVATE, fd, 0);
    T("H2D 1 GiB from mmap'd file (page cache)", G, hipMemcpy(d, mi, G, hipMemcpyHostToDevice));
    munmap(mi, G);
    ftruncate(fd, 0); ftruncate(fd, G);
    char* mo = (char*)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    T("D2H 1 GiB into mmap'd fresh file", G, hipMemcpy(mo, d, G, hipMemcpyDeviceToHost));
    T("D2H 1 GiB into mmap'd file again", G, hipMemcpy(mo, d, G, hipMemcpyDeviceToHost));
    munmap(mo, G); close(fd); unlink(path);
    return 0;
}
