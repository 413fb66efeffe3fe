// The CPU budget of this process: how many host threads the library may start for reading files and writing text.
//
// The reference caps its per-sample process fan-out by MaxCpuCores (run.py:387-400: min(psutil.cpu_count(), MaxCpuCores)) and gives
// every process one core.  Here one process per GPU does the host work of its shard on threads, so the same cap has to be divided
// among the ranks of the node: std::thread::hardware_concurrency() says 256 on a box whose cgroup grants the time of 16, and eight
// ranks that each start "8 readers on a big host" would put 64 readers, their writer pools and 8 Python processes on those 16 cores.
//
//   usable    = CPUs in the affinity mask, capped by the cgroup's CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us),
//               capped by SNPGPU_MAX_CPU_CORES (the MaxCpuCores of this build: snpgpu_set_max_cpu_cores or the environment)
//   ranks     = processes that share the node: snpgpu_set_local_ranks, else SNPGPU_LOCAL_RANKS, else LOCAL_WORLD_SIZE (what
//               torch.distributed.run exports), else 1
//   budget    = max(1, usable / ranks)
// Host code only; no device work in this file.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>

#include "internal.h"

namespace {

std::atomic<uint32_t> g_max_cores{0};       // 0 = not set by the host program
std::atomic<uint32_t> g_local_ranks{0};

uint32_t env_u32(const char *name) {
    const char *e = getenv(name);
    if (!e || !*e) return 0;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end && *end == 0 && v > 0 && v < (1 << 20)) ? (uint32_t)v : 0;
}

bool read_words(const std::string &path, char *buf, size_t n) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    const size_t got = fread(buf, 1, n - 1, f);
    fclose(f);
    buf[got] = 0;
    return got > 0;
}

// CPUs the cgroup's quota is worth (rounded down, at least 1), or 0 when there is no quota / no cgroup file system to ask.
uint32_t quota_cpus() {
    const char *root_env = getenv("SNPGPU_CGROUP_ROOT");          // tests point this at a directory of their own
    const std::string root = root_env && *root_env ? root_env : "/sys/fs/cgroup";
    char buf[128];
    if (read_words(root + "/cpu.max", buf, sizeof buf)) {         // v2: "<quota|max> <period>"
        long long q = 0, p = 0;
        if (strncmp(buf, "max", 3) == 0) return 0;
        if (sscanf(buf, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0) return (uint32_t)(q / p > 0 ? q / p : 1);
        return 0;
    }
    char buf2[64];
    if (read_words(root + "/cpu/cpu.cfs_quota_us", buf, sizeof buf) && read_words(root + "/cpu/cpu.cfs_period_us", buf2, sizeof buf2)) {
        const long long q = atoll(buf), p = atoll(buf2);
        if (q > 0 && p > 0) return (uint32_t)(q / p > 0 ? q / p : 1);
    }
    return 0;
}

uint32_t affinity_cpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n > 0) return (uint32_t)n;
    }
    // more CPUs than cpu_set_t holds (1024), or no affinity call: the dynamically sized form, then the online count
    for (int cpus = 2048; cpus <= 65536; cpus *= 2) {
        cpu_set_t *big = CPU_ALLOC(cpus);
        if (!big) break;
        const size_t sz = CPU_ALLOC_SIZE(cpus);
        CPU_ZERO_S(sz, big);
        const int rc = sched_getaffinity(0, sz, big);
        const int n = rc == 0 ? CPU_COUNT_S(sz, big) : 0;
        CPU_FREE(big);
        if (n > 0) return (uint32_t)n;
    }
    const long on = sysconf(_SC_NPROCESSORS_ONLN);
    return on > 0 ? (uint32_t)on : 1;
}

}  // namespace

void snpgpu_host_budget(snpgpu_cpu_budget_info *out) {
    snpgpu_cpu_budget_info b;
    memset(&b, 0, sizeof b);
    b.affinity_cpus = affinity_cpus();
    b.quota_cpus = quota_cpus();
    uint32_t cap = g_max_cores.load();
    if (!cap) cap = env_u32("SNPGPU_MAX_CPU_CORES");
    b.max_cpu_cores = cap;
    uint32_t usable = b.affinity_cpus;
    if (b.quota_cpus && b.quota_cpus < usable) usable = b.quota_cpus;
    if (cap && cap < usable) usable = cap;
    b.usable_cpus = usable < 1 ? 1 : usable;
    uint32_t ranks = g_local_ranks.load();
    if (!ranks) ranks = env_u32("SNPGPU_LOCAL_RANKS");
    if (!ranks) ranks = env_u32("LOCAL_WORLD_SIZE");
    b.local_ranks = ranks ? ranks : 1;
    b.budget = b.usable_cpus / b.local_ranks;
    if (b.budget < 1) b.budget = 1;
    // readers keep preads in flight beside the issuing thread: 8 saturate the host link of a big box (tools/e2e_readers.py), a small
    // budget leaves a core to the thread that enqueues the copies
    const uint32_t c = b.budget;
    b.readers = c >= 32 ? 8 : (c >= 8 ? c / 2 : (c > 1 ? c - 1 : 1));
    // formatting threads (TSV, FASTA survey, consensus files): CPU-bound, never more than the budget
    b.writers = c >= 32 ? 16 : (c >= 4 ? c / 2 : 1);
    *out = b;
}

uint32_t snpgpu_reader_threads() {
    snpgpu_cpu_budget_info b;
    snpgpu_host_budget(&b);
    return b.readers;
}

uint32_t snpgpu_writer_threads(uint32_t at_most) {
    snpgpu_cpu_budget_info b;
    snpgpu_host_budget(&b);
    uint32_t t = b.writers;
    if (at_most && t > at_most) t = at_most;
    return t < 1 ? 1 : t;
}

uint32_t snpgpu_cpu_threads(uint32_t at_most) {
    snpgpu_cpu_budget_info b;
    snpgpu_host_budget(&b);
    uint32_t t = b.budget;
    if (at_most && t > at_most) t = at_most;
    return t < 1 ? 1 : t;
}

extern "C" {

int snpgpu_cpu_budget(snpgpu_cpu_budget_info *out) {
    if (!out) return SNPGPU_E_ARG;
    snpgpu_host_budget(out);
    return SNPGPU_OK;
}

void snpgpu_set_max_cpu_cores(uint32_t cores) { g_max_cores.store(cores); }

void snpgpu_set_local_ranks(uint32_t ranks) { g_local_ranks.store(ranks); }

}  // extern "C"
