// Development probe: how fast can host threads fill pinned memory from a page-cache file, and how fast does it go to the device?
// hipcc -O2 -o gpurun_out/io_probe tools/io_probe.cpp -lpthread ; ./io_probe <file> 
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    size_t n = st.st_size;
    const size_t chunk = 8 << 20;
    size_t nchunks = n / chunk;
    printf("file %zu bytes, %zu chunks, cpus %u\n", n, nchunks, std::thread::hardware_concurrency());
    void *mm = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    for (int kind = 0; kind < 5; ++kind) {
        const char *names[] = {"malloc", "hipHostMalloc default", "hipHostMalloc noncoherent", "malloc+hipHostRegister", "hipHostMalloc numa-user"};
        for (int threads : {1, 4, 12, 32, 64}) {
            std::vector<void *> bufs(threads);
            double ta = now();
            for (auto &b : bufs) {
                if (kind == 0) { b = aligned_alloc(4096, chunk); memset(b, 1, chunk); }
                else if (kind == 1) hipHostMalloc(&b, chunk, hipHostMallocDefault);
                else if (kind == 2) hipHostMalloc(&b, chunk, hipHostMallocNonCoherent);
                else if (kind == 3) { b = aligned_alloc(4096, chunk); memset(b, 1, chunk); hipHostRegister(b, chunk, hipHostRegisterDefault); }
                else hipHostMalloc(&b, chunk, hipHostMallocNumaUser);
            }
            double talloc = now() - ta;
            for (int mode = 0; mode < 2; ++mode) {       // 0 pread, 1 memcpy from mmap
                std::atomic<size_t> next{0};
                double t0 = now();
                std::vector<std::thread> th;
                for (int t = 0; t < threads; ++t) th.emplace_back([&, t] {
                    for (;;) { size_t c = next.fetch_add(1); if (c >= nchunks) break;
                        if (mode == 0) { size_t got = 0; while (got < chunk) { ssize_t r = pread(fd, (char *)bufs[t] + got, chunk - got, c * chunk + got); if (r <= 0) break; got += r; } }
                        else memcpy(bufs[t], (char *)mm + c * chunk, chunk); }
                });
                for (auto &x : th) x.join();
                double dt = now() - t0;
                printf("%-28s threads %2d %-6s %6.1f GB/s   (alloc %.1f ms per buffer)\n", names[kind], threads, mode ? "memcpy" : "pread", nchunks * chunk / dt / 1e9, talloc * 1e3 / threads);
            }
            for (auto &b : bufs) { if (kind == 0) free(b); else if (kind == 3) { hipHostUnregister(b); free(b); } else hipHostFree(b); }
        }
    }
    // H2D rate from pinned, 8 MiB copies back to back
    void *h, *d; hipHostMalloc(&h, chunk, hipHostMallocDefault); hipMalloc(&d, chunk * 16);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) { double t0 = now(); for (int i = 0; i < 64; ++i) hipMemcpyAsync((char *)d + (i % 16) * chunk, h, chunk, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        printf("H2D pinned 8 MiB x64: %.1f GB/s\n", 64 * chunk / (now() - t0) / 1e9); }
    // pageable H2D of the mmap
    { double t0 = now(); size_t m = n < (1u << 30) ? n : (1u << 30); void *dd; hipMalloc(&dd, m); hipMemcpy(dd, mm, m, hipMemcpyHostToDevice); printf("H2D pageable from mmap: %.1f GB/s\n", m / (now() - t0) / 1e9); }
    return 0;
}
