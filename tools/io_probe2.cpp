// Development probe: first pass vs second pass over a freshly written page-cache file, pread / pread+NOREUSE / mmap+memcpy,
// readers pinned to one NUMA node or not.   io_probe2 <file> <mode: pread|noreuse|mmap> <threads> <node|-1>
#include <fcntl.h>
#include <sched.h>
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
static bool node_cpus(int node, cpu_set_t *set) {
    char path[128], line[1024] = {0};
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    bool ok = fgets(line, sizeof line, fp);
    fclose(fp);
    CPU_ZERO(set);
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) { int a, b; int k = sscanf(tok, "%d-%d", &a, &b); if (k == 1) b = a; if (k >= 1) for (int c = a; c <= b; ++c) CPU_SET(c, set); }
    return ok;
}
int main(int argc, char **argv) {
    const char *path = argv[1], *mode = argv[2];
    int threads = atoi(argv[3]), node = atoi(argv[4]);
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    size_t n = st.st_size;
    const size_t chunk = 16 << 20;
    size_t nchunks = n / chunk;
    if (!strcmp(mode, "noreuse")) printf("fadvise NOREUSE rc=%d\n", posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE));
    cpu_set_t set; bool pin = node >= 0 && node_cpus(node, &set);
    std::vector<void *> bufs(threads);
    for (auto &b : bufs) { b = aligned_alloc(4096, chunk); memset(b, 1, chunk); }
    for (int pass = 0; pass < 3; ++pass) {
        std::atomic<size_t> next{0};
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back([&, t] {
            if (pin) sched_setaffinity(0, sizeof set, &set);
            for (;;) { size_t c = next.fetch_add(1); if (c >= nchunks) break;
                if (strcmp(mode, "mmap")) { size_t got = 0; while (got < chunk) { ssize_t r = pread(fd, (char *)bufs[t] + got, chunk - got, c * chunk + got); if (r <= 0) break; got += r; } }
                else { void *m = mmap(nullptr, chunk, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, c * chunk); memcpy(bufs[t], m, chunk); munmap(m, chunk); } }
        });
        for (auto &x : th) x.join();
        double dt = now() - t0;
        printf("%-8s threads %2d node %2d pass %d: %6.1f GB/s\n", mode, threads, node, pass, nchunks * chunk / dt / 1e9);
    }
    return 0;
}
