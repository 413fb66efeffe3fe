// hipHostMalloc cost by size; does it block hipMalloc / copies of other threads; D2H into pageable vs pinned memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0)); CK(hipFree(0));
    for (size_t mb : {16, 256, 1024}) {
        void *h = nullptr;
        double t = now();
        CK(hipHostMalloc(&h, mb << 20, hipHostMallocDefault));
        double t1 = now();
        CK(hipHostFree(h));
        printf("hipHostMalloc %4zu MiB: %.1f ms, free %.1f ms\n", mb, (t1 - t) * 1e3, (now() - t1) * 1e3);
    }
    {   // register malloc'ed (touched) memory instead
        size_t n = (size_t)1 << 30;
        void *m = aligned_alloc(4096, n);
        memset(m, 1, n);
        double t = now();
        CK(hipHostRegister(m, n, hipHostRegisterDefault));
        printf("hipHostRegister 1 GiB of touched memory: %.1f ms\n", (now() - t) * 1e3);
        CK(hipHostUnregister(m));
        free(m);
    }
    {   // hipMalloc + copies in this thread while another thread pins 1 GiB twice
        size_t n = (size_t)64 << 20;
        void *h = nullptr, *d = nullptr;
        CK(hipHostMalloc(&h, n, hipHostMallocDefault)); CK(hipMalloc(&d, n));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        std::atomic<bool> done{false};
        double pin_ms = 0;
        std::thread th([&] { (void)hipSetDevice(0); double t = now(); void *a = nullptr, *b = nullptr; (void)hipHostMalloc(&a, (size_t)1 << 30, hipHostMallocDefault); (void)hipHostMalloc(&b, (size_t)1 << 30, hipHostMallocDefault); pin_ms = (now() - t) * 1e3; done.store(true); (void)hipHostFree(a); (void)hipHostFree(b); });
        double worst_alloc = 0, worst_copy = 0; int k = 0;
        double t0 = now();
        while (!done.load()) {
            void *p = nullptr;
            double t = now();
            (void)hipMalloc(&p, (size_t)1 << 30);
            double t1 = now();
            (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
            (void)hipStreamSynchronize(st);
            double t2 = now();
            (void)hipFree(p);
            if (t1 - t > worst_alloc) worst_alloc = t1 - t;
            if (t2 - t1 > worst_copy) worst_copy = t2 - t1;
            ++k;
        }
        th.join();
        printf("while another thread pinned 2 x 1 GiB (%.1f ms): %d rounds in %.1f ms, slowest hipMalloc %.1f ms, slowest 64 MiB copy %.1f ms\n", pin_ms, k, (now() - t0) * 1e3, worst_alloc * 1e3, worst_copy * 1e3);
    }
    {   // D2H of 800 MB: pinned vs pageable destination
        size_t n = (size_t)800 << 20;
        void *d = nullptr, *hp = nullptr;
        CK(hipMalloc(&d, n)); CK(hipMemset(d, 1, n));
        CK(hipHostMalloc(&hp, n, hipHostMallocDefault));
        void *pg = malloc(n);
        memset(pg, 0, n);
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            double t = now();
            CK(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            double t1 = now();
            CK(hipMemcpyAsync(pg, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            double t2 = now();
            printf("D2H 800 MiB: pinned %.1f ms (%.1f GB/s), pageable %.1f ms (%.1f GB/s)\n", (t1 - t) * 1e3, n / (t1 - t) / 1e9, (t2 - t1) * 1e3, n / (t2 - t1) / 1e9);
        }
    }
    return 0;
}
