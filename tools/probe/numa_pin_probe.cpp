// Probe: does the host-to-device rate of a pinned buffer depend on the CPU the allocating thread ran on (the NUMA node its pages
// come from)?  For every NUMA node: bind this thread to the node's CPUs, hipHostMalloc + touch 256 MiB, ask the kernel where the
// pages are, time copies to the device.   hipcc -O2 -o numa_pin_probe numa_pin_probe.cpp && ./numa_pin_probe
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <string>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool parse_cpulist(const char *path, cpu_set_t *set) {
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    char line[4096] = {0};
    bool any = false;
    CPU_ZERO(set);
    if (fgets(line, sizeof line, fp))
        for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            int k = sscanf(tok, "%d-%d", &a, &b);
            if (k == 1) b = a;
            if (k >= 1) for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET(c, set); any = true; }
        }
    fclose(fp);
    return any;
}

static int node_of(void *p) {                                   // get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR)
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0, p, 3 /* MPOL_F_NODE|MPOL_F_ADDR */) != 0) return -1;
    return node;
}

int main() {
    hipSetDevice(0);
    char bdf[64] = {0};
    hipDeviceGetPCIBusId(bdf, sizeof bdf, 0);
    for (char *c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    if (FILE *fp = fopen(path, "r")) { int n = -9; if (fscanf(fp, "%d", &n) == 1) printf("GPU %s numa_node %d\n", bdf, n); fclose(fp); }
    const size_t N = 256u << 20;
    void *d = nullptr;
    hipMalloc(&d, N);
    cpu_set_t all;
    sched_getaffinity(0, sizeof all, &all);
    for (int node = 0; node < 16; ++node) {
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        cpu_set_t set;
        if (!parse_cpulist(path, &set)) continue;
        if (sched_setaffinity(0, sizeof set, &set) != 0) { printf("node %d: cannot bind\n", node); continue; }
        for (int flags_i = 0; flags_i < 2; ++flags_i) {
            void *h = nullptr;
            const unsigned flags = flags_i ? hipHostMallocNumaUser : hipHostMallocDefault;
            if (hipHostMalloc(&h, N, flags) != hipSuccess) { printf("node %d: hipHostMalloc failed\n", node); continue; }
            memset(h, 1, N);
            const int where = node_of(h), where_end = node_of((char *)h + N - 4096);
            hipMemcpy(d, h, N, hipMemcpyHostToDevice);
            double t = now_s();
            for (int r = 0; r < 8; ++r) hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, 0);
            hipStreamSynchronize(0);
            t = now_s() - t;
            printf("thread on node %d, %s: pages on node %d..%d, H2D %.1f GB/s\n", node, flags_i ? "NumaUser" : "Default ", where, where_end, 8.0 * N / t / 1e9);
            hipHostFree(h);
        }
    }
    sched_setaffinity(0, sizeof all, &all);
    return 0;
}
