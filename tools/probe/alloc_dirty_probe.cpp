// Is hipMalloc slower for device memory that was used and freed a moment ago, and does a slow hipMalloc in one thread hold up
// the host-to-device copies of another?     hipcc -O2 -o alloc_dirty_probe alloc_dirty_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0)); CK(hipFree(0));
    const size_t blk = (size_t)8 << 30;
    auto time_allocs = [&](const char *what, int n) {
        std::vector<void *> ps;
        double worst = 0, total = 0;
        for (int i = 0; i < n; ++i) { void *p = nullptr; double t = now(); if (hipMalloc(&p, blk) != hipSuccess) break; double d = now() - t; total += d; if (d > worst) worst = d; ps.push_back(p); }
        printf("%s: %zu x hipMalloc(8 GiB): total %.1f ms, slowest %.1f ms\n", what, ps.size(), total * 1e3, worst * 1e3);
        return ps;
    };
    auto ps = time_allocs("never used", 8);
    for (void *p : ps) CK(hipMemset(p, 1, blk));                 // touch everything
    CK(hipDeviceSynchronize());
    double t = now();
    for (void *p : ps) CK(hipFree(p));
    printf("hipFree x %zu: %.1f ms\n", ps.size(), (now() - t) * 1e3);
    ps = time_allocs("right after use + free", 8);
    for (void *p : ps) CK(hipFree(p));
    // fill most of the device, free, allocate again
    {
        std::vector<void *> big;
        for (int i = 0; i < 30; ++i) { void *p = nullptr; if (hipMalloc(&p, blk) != hipSuccess) break; (void)hipMemset(p, 2, blk); big.push_back(p); }
        CK(hipDeviceSynchronize());
        printf("filled %zu x 8 GiB\n", big.size());
        for (void *p : big) CK(hipFree(p));
    }
    ps = time_allocs("after filling the device and freeing it", 8);
    for (void *p : ps) CK(hipFree(p));
    {   // copies alone, then copies while another thread allocates "used" memory
        size_t n = (size_t)256 << 20;
        void *h = nullptr, *d = nullptr;
        CK(hipHostMalloc(&h, n, hipHostMallocDefault)); CK(hipMalloc(&d, n));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        auto copies = [&](int reps) { double t0 = now(); for (int i = 0; i < reps; ++i) (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); return reps * (double)n / (now() - t0) / 1e9; };
        copies(2);
        printf("H2D alone: %.1f GB/s\n", copies(16));
        double alloc_ms = 0; int allocs = 0;
        std::thread th([&] { (void)hipSetDevice(0); std::vector<void *> q; for (int i = 0; i < 8; ++i) { void *p = nullptr; double t0 = now(); if (hipMalloc(&p, blk) != hipSuccess) break; alloc_ms += (now() - t0) * 1e3; ++allocs; q.push_back(p); } for (void *p : q) (void)hipFree(p); });
        double r = copies(64);
        th.join();
        printf("H2D while another thread allocates 8 x 8 GiB: %.1f GB/s (%d allocations, %.1f ms each)\n", r, allocs, allocs ? alloc_ms / allocs : 0.0);
    }
    return 0;
}
