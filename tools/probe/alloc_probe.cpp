// How long does device memory take to allocate on this box, by size and by API, and does allocating in one thread slow down
// host-to-device copies issued by another?   hipcc -O2 -o alloc_probe alloc_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(0));
    for (size_t gb : {1, 2, 4, 8, 16}) {
        void *p = nullptr;
        double t = now();
        CK(hipMalloc(&p, gb << 30));
        double t1 = now();
        CK(hipFree(p));
        double t2 = now();
        printf("hipMalloc %2zu GiB: %.1f ms (%.1f ms/GiB), hipFree %.1f ms\n", gb, (t1 - t) * 1e3, (t1 - t) * 1e3 / gb, (t2 - t1) * 1e3);
    }
    {   // second allocation of the same size after a free: is freed memory handed back faster?
        void *p = nullptr;
        CK(hipMalloc(&p, (size_t)8 << 30)); CK(hipFree(p));
        double t = now();
        CK(hipMalloc(&p, (size_t)8 << 30));
        printf("hipMalloc 8 GiB again: %.1f ms\n", (now() - t) * 1e3);
        CK(hipFree(p));
    }
    {   // stream-ordered allocation
        hipStream_t st; CK(hipStreamCreate(&st));
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int rep = 0; rep < 2; ++rep) {
            void *p = nullptr;
            double t = now();
            CK(hipMallocAsync(&p, (size_t)8 << 30, st));
            CK(hipStreamSynchronize(st));
            double t1 = now();
            CK(hipFreeAsync(p, st));
            CK(hipStreamSynchronize(st));
            printf("hipMallocAsync 8 GiB (rep %d): %.1f ms, free %.1f ms\n", rep, (t1 - t) * 1e3, (now() - t1) * 1e3);
        }
    }
    {   // virtual memory management: reserve + create + map
        size_t gran = 0;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        size_t sz = (size_t)8 << 30;
        void *va = nullptr;
        double t = now();
        CK(hipMemAddressReserve(&va, sz, gran, nullptr, 0));
        hipMemGenericAllocationHandle_t h;
        double t1 = now();
        CK(hipMemCreate(&h, sz, &prop, 0));
        double t2 = now();
        CK(hipMemMap(va, sz, 0, h, 0));
        hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, sz, &acc, 1));
        double t3 = now();
        printf("VMM 8 GiB: reserve %.1f ms, create %.1f ms, map+access %.1f ms (granularity %zu)\n", (t1 - t) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, gran);
        CK(hipMemUnmap(va, sz)); CK(hipMemRelease(h)); CK(hipMemAddressFree(va, sz));
    }
    {   // copies alone, then copies while another thread allocates
        size_t n = (size_t)256 << 20;
        void *h = nullptr, *d = nullptr;
        CK(hipHostMalloc(&h, n, hipHostMallocDefault)); CK(hipMalloc(&d, n));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        auto copies = [&](int reps) { double t = now(); for (int i = 0; i < reps; ++i) (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); return reps * (double)n / (now() - t) / 1e9; };
        copies(2);
        printf("H2D alone: %.1f GB/s\n", copies(16));
        std::atomic<bool> stop{false};
        double alloc_ms = 0; int allocs = 0;
        std::thread th([&] { (void)hipSetDevice(0); std::vector<void *> ps; while (!stop.load()) { void *p = nullptr; double t = now(); if (hipMalloc(&p, (size_t)4 << 30) != hipSuccess) break; alloc_ms += (now() - t) * 1e3; ++allocs; ps.push_back(p); if (ps.size() >= 20) break; } for (void *p : ps) (void)hipFree(p); });
        double r = copies(32);
        stop.store(true);
        th.join();
        printf("H2D while another thread allocates 4 GiB blocks: %.1f GB/s (%d allocations, %.1f ms each)\n", r, allocs, allocs ? alloc_ms / allocs : 0.0);
    }
    return 0;
}
