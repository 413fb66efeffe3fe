// Streamed ingestion for call_consensus: pileup files (or host buffers) -> consensus bytes, without ever waiting for a
// whole file to be resident.
//
// Replaces the way the reference feeds pileup.Reader (snppipeline/pileup.py:408-429: a text-mode line iterator over the
// open file, driven from call_consensus.py:161) and the one-process-per-sample job array around it (run.py:704-718).
// A pileup is 0.4 GB of text per 5 Mbp x 30x sample: the 10 000-sample target is 4 TB that can only pass through HBM,
// so the rate that matters end to end is the host link's, and the job of this file is to keep that link busy:
//
//   reader threads   pread() the file (page cache) in chunks of whole 4 KiB scan tiles into a ring of PINNED staging
//                    buffers — a kernel-side memcpy of ~5 GB/s per thread, hence several threads
//   copy stream      one hipMemcpyAsync per chunk, staging -> the file's device buffer ("slot"), event per chunk
//   compute stream   waits for the chunk's event and scans the tiles that are now complete (tile + 128-byte halo inside
//                    the landed bytes) with the batch scan kernel restricted to that tile range (scan.hip:
//                    snpgpu_scan_range); lines the fast parser leaves to the exact parser are queued with their file
//                    offsets and finished when the file is complete, as is the call step (snpgpu_enqueue_call), which
//                    reads the lines the scan selected straight from the slot
//   results          base / filter bytes (+ optional per-site counts and line offsets) come back through pinned buffers
// Files alternate between n_slots device buffers, so file k+1 streams in while file k's tail (last tiles, queue, call,
// results) runs.  Everything on the compute stream is in order, so all device scratch is shared between files.
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <algorithm>

#include "internal.h"

namespace {

inline size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Source {
    const char *path = nullptr;     // a file ...
    const uint8_t *mem = nullptr;   // ... or host memory
    uint64_t size = 0;
    int fd = -1;
    int rc = SNPGPU_OK;
};

struct Job {
    uint32_t file;
    uint64_t off, len;
    bool first, last;
    uint32_t chunk;                 // index of the chunk inside its file
};

// The files of a batch are opened by a thread of their own, a few files ahead of the readers, and each is closed by the reader
// that reads its last piece: at most OPEN_WINDOW of them are open at a time.  Opening all of them up front is what costs: a
// process whose descriptor table has to grow beyond its current size waits for an RCU grace period per doubling while the
// table is shared between threads (expand_fdtable), and so does every other thread that needs a descriptor meanwhile — 0.3 s
// for the 125 files of a job on the 256-CPU box, and the pinning of the staging memory stood still with it
// (tools/pipeline_time.py --probe-open, tools/probe/open_stall_probe.py).  Sizes come from stat(); a file that cannot be opened, or
// has become smaller than stat() said, is an I/O error of that file (its pieces are read as blank lines).
struct Opener {
    static constexpr uint32_t OPEN_WINDOW = 16;
    std::vector<Source> *src = nullptr;
    std::unique_ptr<std::atomic<uint8_t>[]> opened;
    std::unique_ptr<std::atomic<uint32_t>[]> reads_left;        // pieces of the file not read yet
    std::atomic<uint32_t> n_closed{0};
    std::atomic<bool> stop{false};
    std::thread th;

    void open_one(Source &s) const {
        if (s.rc != SNPGPU_OK || s.size == 0 || !s.path) return;
        s.fd = open(s.path, O_RDONLY | O_CLOEXEC);
        struct stat stt;
        if (s.fd < 0 || fstat(s.fd, &stt) != 0 || !S_ISREG(stt.st_mode) || (uint64_t)stt.st_size < s.size) {
            if (s.fd >= 0) { close(s.fd); s.fd = -1; }
            s.rc = SNPGPU_E_IO;
            return;
        }
        (void)posix_fadvise(s.fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    }
    void run() {
        for (size_t f = 0; f < src->size(); ++f) {
            while (!stop.load(std::memory_order_relaxed) && f >= (size_t)n_closed.load(std::memory_order_acquire) + OPEN_WINDOW)
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (!stop.load(std::memory_order_relaxed)) open_one((*src)[f]);
            opened[f].store(1, std::memory_order_release);       // (after a cancel: nobody reads any more, nobody may wait either)
        }
    }
    // pieces[f]: the number of jobs of file f (each of them ends in one done_reading(f))
    void start(std::vector<Source> *sources, const std::vector<uint32_t> &pieces) {
        src = sources;
        const size_t n = sources->size() ? sources->size() : 1;
        opened.reset(new std::atomic<uint8_t>[n]);
        reads_left.reset(new std::atomic<uint32_t>[n]);
        for (size_t f = 0; f < sources->size(); ++f) {
            opened[f].store(0, std::memory_order_relaxed);
            reads_left[f].store(pieces[f], std::memory_order_relaxed);
        }
        try { th = std::thread(&Opener::run, this); } catch (const std::exception &) { stop.store(true); run(); stop.store(false); open_all_now(); }
    }
    void open_all_now() { for (auto &s : *src) open_one(s); }   // no thread to be had: everything up front after all
    void wait(uint32_t f) const {
        while (!opened[f].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    void done_reading(uint32_t f) {                             // by the reader that has just read (or skipped) a piece of file f
        if (reads_left[f].fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        Source &s = (*src)[f];
        if (s.fd >= 0 && s.path) { close(s.fd); s.fd = -1; }
        n_closed.fetch_add(1, std::memory_order_release);
    }
    void cancel() { stop.store(true); }                         // before the readers are told to give up: none of them may wait for a file
    void finish() {
        stop.store(true);
        if (th.joinable()) th.join();
    }
};

}  // namespace

// Persistent per-context resources of the streaming path (pinned memory is expensive to allocate: keep it).
struct snpgpu_stream_pool {
    hipStream_t copy_stream = nullptr;      // chunks alternate between two copy streams: the set-up of one copy hides
    hipStream_t copy_stream2 = nullptr;     // behind the transfer of the other
    cpu_set_t near_cpus;                    // CPUs of the NUMA node the device hangs off (reader threads run there)
    bool have_near_cpus = false;
    size_t chunk_bytes = 0;
    std::vector<void *> staging;            // pinned, chunk_bytes each
    std::vector<hipEvent_t> ev_copy;        // the H2D copy out of staging[i] has completed
    std::vector<void *> slot;               // device file buffers
    size_t slot_bytes = 0;
    std::vector<hipEvent_t> ev_done;        // the results of the file in slot i are in its pinned result block
    std::vector<void *> result;             // pinned result blocks, one per slot
    size_t result_bytes = 0;
    std::vector<void *> table_host;         // pinned mirrors of the per-chunk sample tables, one per slot
    size_t table_bytes = 0;
};

static void pool_free_host(std::vector<void *> &v) {
    for (void *p : v) if (p) (void)hipHostFree(p);
    v.clear();
}

void snpgpu_stream_pool_destroy(snpgpu_ctx *ctx) {
    snpgpu_stream_pool *p = ctx->pool;
    if (!p) return;
    pool_free_host(p->staging);
    pool_free_host(p->result);
    pool_free_host(p->table_host);
    for (void *d : p->slot) if (d) (void)hipFree(d);
    for (auto e : p->ev_copy) (void)hipEventDestroy(e);
    for (auto e : p->ev_done) (void)hipEventDestroy(e);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    if (p->copy_stream2) (void)hipStreamDestroy(p->copy_stream2);
    delete p;
    ctx->pool = nullptr;
}

namespace {

// n_slots device file buffers of slot_bytes; n_blocks (>= n_slots) pinned result blocks, table mirrors and done-events — files
// that go to resident memory (snpgpu_pileups) need the blocks but no slot
int pool_ensure(snpgpu_ctx *ctx, size_t chunk_bytes, uint32_t n_staging, uint32_t n_slots, size_t slot_bytes, size_t result_bytes,
                size_t table_bytes, uint32_t n_blocks = 0) {
    if (n_blocks < n_slots) n_blocks = n_slots;
    if (!ctx->pool) ctx->pool = new snpgpu_stream_pool();
    snpgpu_stream_pool *p = ctx->pool;
    if (!p->copy_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->copy_stream2, hipStreamNonBlocking));
        // the CPUs next to the device: /sys/bus/pci/devices/<bdf>/local_cpulist (best effort)
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, sizeof bdf, ctx->device) == hipSuccess) {
            for (char *c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
            char path[160];
            snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
            if (FILE *fp = fopen(path, "r")) {
                char line[1024] = {0};
                if (fgets(line, sizeof line, fp)) {
                    CPU_ZERO(&p->near_cpus);
                    int n_set = 0;
                    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
                        int a = 0, b = 0;
                        const int k = sscanf(tok, "%d-%d", &a, &b);
                        if (k == 1) b = a;
                        if (k >= 1) for (int c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET(c, &p->near_cpus); ++n_set; }
                    }
                    p->have_near_cpus = n_set > 0;
                }
                fclose(fp);
            }
        }
    }
    if (p->chunk_bytes != chunk_bytes) {
        pool_free_host(p->staging);
        p->chunk_bytes = chunk_bytes;
    }
    while (p->staging.size() < n_staging) {
        void *h = nullptr;
        hipError_t e = hipHostMalloc(&h, chunk_bytes, hipHostMallocDefault);
        if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipHostMalloc(%zu) failed: %s", chunk_bytes, hipGetErrorString(e));
        p->staging.push_back(h);
    }
    while (p->ev_copy.size() < p->staging.size()) {
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        p->ev_copy.push_back(ev);
    }
    if (p->slot_bytes < slot_bytes || p->slot.size() < n_slots) {
        for (void *d : p->slot) if (d) (void)hipFree(d);
        p->slot.clear();
        const size_t want = p->slot_bytes > slot_bytes ? p->slot_bytes : slot_bytes;
        for (uint32_t i = 0; i < n_slots; ++i) {
            void *d = nullptr;
            hipError_t e = hipMalloc(&d, want);
            if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipMalloc(%zu) for a pileup slot failed: %s", want, hipGetErrorString(e));
            p->slot.push_back(d);
        }
        p->slot_bytes = want;
    }
    if (n_blocks < p->slot.size()) n_blocks = (uint32_t)p->slot.size();
    while (p->ev_done.size() < n_blocks) {
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        p->ev_done.push_back(ev);
    }
    if (p->result_bytes < result_bytes || p->result.size() < n_blocks) {
        pool_free_host(p->result);
        if (result_bytes < p->result_bytes) result_bytes = p->result_bytes;
        for (size_t i = 0; i < n_blocks; ++i) {
            void *h = nullptr;
            hipError_t e = hipHostMalloc(&h, result_bytes, hipHostMallocDefault);
            if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipHostMalloc(%zu) failed: %s", result_bytes, hipGetErrorString(e));
            p->result.push_back(h);
        }
        p->result_bytes = result_bytes;
    }
    if (p->table_bytes < table_bytes || p->table_host.size() < n_blocks) {
        pool_free_host(p->table_host);
        if (table_bytes < p->table_bytes) table_bytes = p->table_bytes;
        for (size_t i = 0; i < n_blocks; ++i) {
            void *h = nullptr;
            hipError_t e = hipHostMalloc(&h, table_bytes, hipHostMallocDefault);
            if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipHostMalloc(%zu) failed: %s", table_bytes, hipGetErrorString(e));
            p->table_host.push_back(h);
        }
        p->table_bytes = table_bytes;
    }
    return SNPGPU_OK;
}

struct Outputs {
    uint8_t *base, *filters;
    snpgpu_site_counts *counts;     // nullable
    uint64_t *line_off;             // nullable
    uint64_t *status;
    int32_t *rc;                    // nullable
};

// What the readers and the orchestrator share.
struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint8_t> filled;    // job j has been read into staging[j % R]
    std::vector<int> job_err;       // errno of a failed read (0: fine)
    int64_t freed = 0;              // jobs whose H2D copy has completed: their staging buffers can be refilled
    uint64_t R = 1;                 // staging buffers in the ring
    std::atomic<uint64_t> ns_reading{0}, ns_waiting{0};   // summed over the readers
    bool abort = false;
    std::atomic<uint64_t> next{0};
    uint64_t base = 0;              // the ring serves jobs [base, ...): job j uses staging[(j - base) % R]
    Opener *opener = nullptr;       // (optional) the files are opened by another thread and closed by the last reader
};

void reader_main(snpgpu_ctx *ctx, Shared *sh, const std::vector<Job> *jobs, std::vector<Source> *src) {
    snpgpu_stream_pool *p = ctx->pool;
    if (p->have_near_cpus) (void)sched_setaffinity(0, sizeof p->near_cpus, &p->near_cpus);   // this thread only; best effort
    const uint64_t R = sh->R;
    for (;;) {
        const uint64_t j = sh->next.fetch_add(1);
        if (j >= jobs->size()) return;
        const Job &jb = (*jobs)[j];
        const double t_w = now_s();
        const uint64_t k = j - sh->base;                    // place in the ring
        if (k >= R) {                                       // staging[k % R] is free once the job R places before has been copied out of it
            std::unique_lock<std::mutex> lk(sh->mu);         // (the issuing thread watches the copy events: readers never call HIP)
            sh->cv.wait(lk, [&] { return sh->abort || sh->freed > (int64_t)(k - R); });
            if (sh->abort) return;
        }
        const double t_r = now_s();
        uint8_t *dst = (uint8_t *)p->staging[k % R];
        Source &s = (*src)[jb.file];
        if (sh->opener) sh->opener->wait(jb.file);
        int err = 0;
        if (s.mem) {
            memcpy(dst, s.mem + jb.off, jb.len);
        } else if (s.fd >= 0) {
            uint64_t got = 0;
            while (got < jb.len) {
                ssize_t r = pread(s.fd, dst + got, jb.len - got, (off_t)(jb.off + got));
                if (r < 0) { if (errno == EINTR) continue; err = errno ? errno : EIO; break; }
                if (r == 0) { err = EIO; break; }            // the file shrank under us
                got += (uint64_t)r;
            }
            if (err) memset(dst + got, '\n', jb.len - got);
        } else {
            memset(dst, '\n', jb.len);                       // could not be opened: its result is void (rc says so)
        }
        if (sh->opener) sh->opener->done_reading(jb.file);
        sh->ns_waiting.fetch_add((uint64_t)((t_r - t_w) * 1e9));
        sh->ns_reading.fetch_add((uint64_t)((now_s() - t_r) * 1e9));
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            sh->filled[j] = 1;
            sh->job_err[j] = err;
        }
        sh->cv.notify_all();
    }
}

struct DevScratch {                 // one carve-up of the context's scratch, shared by all files (the compute stream is in order)
    SampleDev *tables;              // [n_slots][table entries]
    size_t table_stride;            // bytes per slot
    uint64_t *totals, *site_line, *todo, *todo2, *status;
    uint32_t *todo_n;
    uint8_t *base, *filters;
    snpgpu_site_counts *counts;
    uint8_t *flag_rows;             // per slot: the site flags of its file (only with per-file exclude lists)
    size_t flag_row_stride;
};

// d_site_line: nullptr = in scratch; the single-pileup form passes the site set's own row, where
// snpgpu_siteset_line_offsets finds it afterwards.
// excl_off / excl_slots: nullptr, or per file the site-set slots of ITS exclude list (CSR, n_files + 1 offsets): the file is
// called with the set's flags | SNPGPU_SITE_EXCLUDED on those slots.
int run_stream(snpgpu_ctx *ctx, const snpgpu_siteset *ss, std::vector<Source> &src, const snpgpu_caller_params *prm,
               const Outputs &out, const snpgpu_stream_opts *opts, snpgpu_stream_stats *stats, uint64_t *d_site_line,
               const uint32_t *excl_off = nullptr, const uint32_t *excl_slots = nullptr) {
    const double t_start = now_s();
    const uint32_t n_files = (uint32_t)src.size();
    const uint32_t n_sites = ss->n_sites;
    if (stats) memset(stats, 0, sizeof *stats);
    if (!n_files) return SNPGPU_OK;
    if (excl_off) {
        if (excl_off[n_files] && !excl_slots) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "exclude slots missing");
        for (uint32_t f = 0; f < n_files; ++f) {
            if (excl_off[f + 1] < excl_off[f]) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "exclude offsets must be non-decreasing");
            for (uint32_t k = excl_off[f]; k < excl_off[f + 1]; ++k)
                if (excl_slots[k] >= n_sites) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "exclude slot %u of file %u is not in the site set", excl_slots[k], f);
        }
    }
    HIP_TRY(ctx, snpgpu_enter(ctx));
    if (out.counts) { int rc0 = snpgpu_spill_begin(ctx); if (rc0) return rc0; }
    size_t chunk = opts && opts->chunk_bytes ? opts->chunk_bytes : (size_t)16 << 20;
    chunk = up(chunk < 65536 ? 65536 : chunk, SNPGPU_SCAN_TILE);
    const int want_depth = opts && opts->want_depth_sum ? 1 : 0;

    // ---- sources and jobs -------------------------------------------------------------------------------------
    uint64_t max_size = 0, total_bytes = 0;
    for (auto &s : src) {
        if (s.path) {                                           // (sized here, opened by the opener thread as the readers get to it)
            struct stat stt;
            if (stat(s.path, &stt) != 0 || !S_ISREG(stt.st_mode)) {
                s.rc = SNPGPU_E_IO;
                s.size = 0;
            } else {
                s.size = (uint64_t)stt.st_size;
            }
        }
        if (s.size > max_size) max_size = s.size;
        total_bytes += s.size;
    }
    Opener opener;
    auto close_all = [&]() { opener.finish(); for (auto &s2 : src) if (s2.fd >= 0) { close(s2.fd); s2.fd = -1; } };
    std::vector<Job> jobs;
    std::vector<uint32_t> chunks_of(n_files);
    uint32_t max_chunks = 1;
    for (uint32_t f = 0; f < n_files; ++f) {
        const uint64_t n = src[f].size;
        const uint32_t nc = n ? (uint32_t)((n + chunk - 1) / chunk) : 1;   // an empty file still takes one (empty) job
        chunks_of[f] = nc;
        if (nc > max_chunks) max_chunks = nc;
        for (uint32_t c = 0; c < nc; ++c) {
            const uint64_t off = (uint64_t)c * chunk;
            jobs.push_back(Job{f, off, n - off < chunk ? n - off : chunk, c == 0, c + 1 == nc, c});
        }
    }
    const uint64_t J = jobs.size();
    opener.start(&src, chunks_of);                          // (the first files open while the staging memory is being pinned)

    // ---- resources ----------------------------------------------------------------------------------------------
    uint32_t n_readers = opts && opts->n_readers ? opts->n_readers : 0;
    if (!n_readers) {
        n_readers = snpgpu_reader_threads();                     // the CPU budget of this process, not the box's CPU count
    }
    if (n_readers > J) n_readers = (uint32_t)J;
    uint32_t n_staging = opts && opts->n_staging ? opts->n_staging : n_readers + 4;
    if (n_staging > J) n_staging = (uint32_t)J;
    if (n_staging < 1) n_staging = 1;
    uint32_t n_slots = opts && opts->n_slots ? opts->n_slots : 2;
    if (n_slots > n_files) n_slots = n_files;
    const size_t slot_bytes = up(max_size + SNPGPU_SCAN_TILE + 256, 4096);
    const size_t r_base = 0, r_filt = up(r_base + n_sites, 256), r_stat = up(r_filt + n_sites, 256), r_line = r_stat + 256;
    const size_t r_cnt = up(r_line + (out.line_off ? 8ull * n_sites : 0), 256);
    const size_t r_flags = up(r_cnt + (out.counts ? sizeof(snpgpu_site_counts) * (size_t)n_sites : 0), 256);
    const size_t result_bytes = r_flags + (excl_off ? up(n_sites, 256) : 0) + 256;
    const size_t table_bytes = up((size_t)(max_chunks + 1) * 2 * sizeof(SampleDev), 256);
    {
        int rc = pool_ensure(ctx, chunk, n_staging, n_slots, slot_bytes, result_bytes, table_bytes);
        if (rc) { close_all(); return rc; }
    }
    snpgpu_stream_pool *p = ctx->pool;
    const uint64_t R = p->staging.size() < n_staging ? p->staging.size() : n_staging;   // ring actually used
    DevScratch ds;
    {
        const size_t list_bytes = up(8ull * n_sites, 256);
        size_t o = 0;
        const size_t o_tab = o; o += table_bytes * p->slot.size();
        const size_t o_tot = o; o += up(snpgpu_scan_totals_bytes(ctx), 256);
        const size_t o_line = o; o += list_bytes;
        const size_t o_todon = o; o += 256;
        const size_t o_todo = o; o += list_bytes;
        const size_t o_todo2 = o; o += list_bytes;
        const size_t o_base = o; o += up(n_sites, 256);
        const size_t o_filt = o; o += up(n_sites, 256);
        const size_t o_stat = o; o += 256;
        const size_t o_cnt = o; o += out.counts ? up(sizeof(snpgpu_site_counts) * (size_t)n_sites, 256) : 0;
        const size_t o_frow = o; o += excl_off ? up(n_sites, 256) * p->slot.size() : 0;
        void *ws = nullptr;
        int rc = snpgpu_scratch(ctx, o + 256, &ws);
        if (rc) { close_all(); return rc; }
        char *b = (char *)ws;
        ds.tables = (SampleDev *)(b + o_tab); ds.table_stride = table_bytes;
        ds.totals = (uint64_t *)(b + o_tot); ds.site_line = (uint64_t *)(b + o_line); ds.todo_n = (uint32_t *)(b + o_todon);
        ds.todo = (uint64_t *)(b + o_todo); ds.todo2 = (uint64_t *)(b + o_todo2); ds.base = (uint8_t *)(b + o_base);
        ds.filters = (uint8_t *)(b + o_filt); ds.status = (uint64_t *)(b + o_stat);
        ds.counts = out.counts ? (snpgpu_site_counts *)(b + o_cnt) : nullptr;
        ds.flag_rows = excl_off ? (uint8_t *)(b + o_frow) : nullptr;
        ds.flag_row_stride = up(n_sites, 256);
        if (d_site_line) ds.site_line = d_site_line;
    }
    hipStream_t st = ctx->stream;
    // whatever the caller enqueued before on the compute stream is done before the slots are overwritten
    {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            close_all();
            return snpgpu_set_error(ctx, SNPGPU_E_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
        }
    }

    Shared sh;
    sh.R = R;
    sh.filled.assign(J, 0);
    sh.job_err.assign(J, 0);
    sh.opener = &opener;
    std::vector<std::thread> readers;
    try {
        for (uint32_t i = 0; i < n_readers; ++i) readers.emplace_back(reader_main, ctx, &sh, &jobs, &src);
    } catch (const std::exception &e) {                      // no thread to be had: nothing has been enqueued yet
        opener.cancel();
        { std::lock_guard<std::mutex> lk(sh.mu); sh.abort = true; sh.next.store(J); }
        sh.cv.notify_all();
        for (auto &t : readers) t.join();
        close_all();
        return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "cannot start the reader threads: %s", e.what());
    }

    int rc = SNPGPU_OK;
    double t_wait_read = 0, t_wait_gpu = 0, t_enqueue = 0;
    auto harvest = [&](uint32_t f) -> int {                  // results of file f: pinned block -> the caller's arrays
        const uint32_t slot = f % (uint32_t)p->slot.size();
        const double t0 = now_s();
        hipError_t e = hipEventSynchronize(p->ev_done[slot]);
        t_wait_gpu += now_s() - t0;
        if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "waiting for the results of pileup %u failed: %s", f, hipGetErrorString(e));
        const char *r = (const char *)p->result[slot];
        if (n_sites) {
            memcpy(out.base + (size_t)f * n_sites, r + r_base, n_sites);
            memcpy(out.filters + (size_t)f * n_sites, r + r_filt, n_sites);
            if (out.line_off) memcpy(out.line_off + (size_t)f * n_sites, r + r_line, 8ull * n_sites);
            if (out.counts) memcpy(out.counts + (size_t)f * n_sites, r + r_cnt, sizeof(snpgpu_site_counts) * (size_t)n_sites);
        }
        uint64_t *stw = out.status + (size_t)f * SNPGPU_SCAN_STATUS_WORDS;
        memcpy(stw, r + r_stat, 8 * SNPGPU_SCAN_STATUS_WORDS);
        if (src[f].rc == SNPGPU_OK && stw[0] != ~0ull)
            src[f].rc = (stw[0] & 0xFF) == SCAN_ERR_NON_ASCII ? SNPGPU_E_UNSUPPORTED : SNPGPU_E_PILEUP;
        return SNPGPU_OK;
    };
#define ST_TRY(expr)                                                                                                \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) { rc = snpgpu_set_error(ctx, SNPGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

    {
        uint32_t harvested = 0;                              // files [0, harvested) have been copied out
        uint32_t cur_waves = 0;                              // waves of the final (whole-file) table entry of the current file
        uint32_t tiles_done = 0;
        int64_t copies_done = 0;                             // copies [0, copies_done) have completed (events polled in order)
        for (uint64_t j = 0; j < J; ++j) {
            const Job &jb = jobs[j];
            const uint32_t f = jb.file, slot = f % (uint32_t)p->slot.size();
            Source &s = src[f];
            uint8_t *d_file = (uint8_t *)p->slot[slot];
            SampleDev *h_tab = (SampleDev *)p->table_host[slot];
            SampleDev *d_tab = (SampleDev *)((char *)ds.tables + ds.table_stride * slot);
            const uint32_t nc = chunks_of[f];
            if (jb.first) {
                // the slot, its table mirror and its result block are free once the file that used them has been harvested
                while (harvested + (uint32_t)p->slot.size() <= f) { rc = harvest(harvested); if (rc) goto done; ++harvested; }
                // one table entry (+ sentinel) per chunk: the part of the file that has landed after chunk c, and the tiles
                // that became complete with it; entry nc describes the whole file for the exact parser and the call step
                const uint64_t n_tiles = snpgpu_scan_tiles(d_file, s.size);
                uint32_t lo = 0;
                for (uint32_t c = 0; c <= nc; ++c) {
                    const bool whole = c + 1 >= nc;
                    const uint64_t landed = whole ? s.size : (uint64_t)(c + 1) * chunk;
                    SampleDev e{};
                    e.buf = d_file;
                    e.nbytes = landed;
                    e.status = ds.status;
                    e.tile_lo = c == nc ? 0 : lo;
                    e.tile_hi = whole ? (uint32_t)n_tiles : (uint32_t)((landed - SNPGPU_SCAN_HALO) / SNPGPU_SCAN_TILE);
                    if (c < nc) lo = e.tile_hi;
                    SampleDev sent{};
                    sent.wave0 = snpgpu_scan_deal(ctx, &e, 1, c == nc ? 1 : 8);
                    h_tab[2 * c] = e;
                    h_tab[2 * c + 1] = sent;
                }
                cur_waves = h_tab[2 * nc + 1].wave0;
                tiles_done = 0;
                ST_TRY(hipMemcpyAsync(d_tab, h_tab, (size_t)(nc + 1) * 2 * sizeof(SampleDev), hipMemcpyHostToDevice, st));
                rc = snpgpu_scan_begin(ctx, ss, d_tab + 2 * nc, 1, ds.site_line, ds.todo_n, 3);
                if (rc) goto done;
            }
            {
                // wait for the chunk to be read; meanwhile retire finished copies so that their buffers go back to the readers
                const double t0 = now_s();
                for (;;) {
                    bool progress = false;
                    while (copies_done < (int64_t)j && hipEventQuery(p->ev_copy[copies_done % R]) != hipErrorNotReady) { ++copies_done; progress = true; }   // (an error counts as done: the next HIP call reports it, nobody waits forever)
                    std::unique_lock<std::mutex> lk(sh.mu);
                    if (progress) { sh.freed = copies_done; lk.unlock(); sh.cv.notify_all(); lk.lock(); }
                    if (sh.filled[j]) break;
                    sh.cv.wait_for(lk, std::chrono::microseconds(copies_done < (int64_t)j ? 20 : 2000), [&] { return sh.filled[j] != 0; });
                    if (sh.filled[j]) break;
                }
                if (sh.job_err[j] && s.rc == SNPGPU_OK) s.rc = SNPGPU_E_IO;
                t_wait_read += now_s() - t0;
            }
            const double t_e = now_s();
            hipStream_t cs = (j & 1) ? p->copy_stream2 : p->copy_stream;
            if (jb.len) ST_TRY(hipMemcpyAsync(d_file + jb.off, p->staging[j % R], jb.len, hipMemcpyHostToDevice, cs));
            ST_TRY(hipEventRecord(p->ev_copy[j % R], cs));
            ST_TRY(hipStreamWaitEvent(st, p->ev_copy[j % R], 0));
            const SampleDev &e = h_tab[2 * jb.chunk];
            if (e.tile_hi > tiles_done) {
                rc = snpgpu_scan_range(ctx, ss, d_tab + 2 * jb.chunk, 1, h_tab[2 * jb.chunk + 1].wave0, ds.totals, ds.site_line, want_depth);
                if (rc) goto done;
                tiles_done = e.tile_hi;
            }
            if (jb.last) {
                const SampleDev *d_whole = d_tab + 2 * nc;
                rc = snpgpu_scan_end(ctx, ss, d_whole, 1, cur_waves, ds.totals, ds.site_line, want_depth);
                char *r = (char *)p->result[slot];
                const uint8_t *d_flags = nullptr;
                if (rc == SNPGPU_OK && excl_off && n_sites) {
                    // this file's site flags: the set's, with its own exclude list on top (the pinned row is free: the file that
                    // used this block before has been harvested)
                    uint8_t *hf = (uint8_t *)(r + r_flags);
                    memcpy(hf, ss->h_flags.data(), n_sites);
                    for (uint32_t k = excl_off[f]; k < excl_off[f + 1]; ++k) hf[excl_slots[k]] |= SNPGPU_SITE_EXCLUDED;
                    uint8_t *d_row = ds.flag_rows + ds.flag_row_stride * slot;
                    ST_TRY(hipMemcpyAsync(d_row, hf, n_sites, hipMemcpyHostToDevice, st));
                    d_flags = d_row;
                }
                if (rc == SNPGPU_OK)
                    rc = snpgpu_enqueue_call(ctx, ss, d_whole, 1, prm, ds.site_line, ds.base, ds.filters, ds.counts, ds.todo_n, ds.todo, ds.todo2,
                                             d_flags, 0);
                if (rc) goto done;
                if (n_sites) {
                    ST_TRY(hipMemcpyAsync(r + r_base, ds.base, n_sites, hipMemcpyDeviceToHost, st));
                    ST_TRY(hipMemcpyAsync(r + r_filt, ds.filters, n_sites, hipMemcpyDeviceToHost, st));
                    if (out.line_off) ST_TRY(hipMemcpyAsync(r + r_line, ds.site_line, 8ull * n_sites, hipMemcpyDeviceToHost, st));
                    if (out.counts) ST_TRY(hipMemcpyAsync(r + r_cnt, ds.counts, sizeof(snpgpu_site_counts) * (size_t)n_sites, hipMemcpyDeviceToHost, st));
                }
                ST_TRY(hipMemcpyAsync(r + r_stat, ds.status, 8 * SNPGPU_SCAN_STATUS_WORDS, hipMemcpyDeviceToHost, st));
                // (the next file that uses this slot is not started before this event has been waited for in harvest())
                ST_TRY(hipEventRecord(p->ev_done[slot], st));
            }
            t_enqueue += now_s() - t_e;
        }
        while (harvested < n_files) { rc = harvest(harvested); if (rc) goto done; ++harvested; }
    }
done:
#undef ST_TRY
    if (rc) opener.cancel();
    {
        std::lock_guard<std::mutex> lk(sh.mu);
        if (rc) { sh.abort = true; sh.next.store(J); }
    }
    sh.cv.notify_all();
    for (auto &t : readers) t.join();
    if (rc) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamSynchronize(p->copy_stream2); (void)hipStreamSynchronize(st); }
    close_all();
    if (out.rc) for (uint32_t f = 0; f < n_files; ++f) out.rc[f] = src[f].rc;
    if (stats) {
        stats->bytes = total_bytes;
        stats->seconds = now_s() - t_start;
        stats->seconds_waiting_for_readers = t_wait_read;
        stats->seconds_waiting_for_device = t_wait_gpu;
        stats->n_chunks = J;
        stats->n_readers = n_readers;
        stats->n_staging = (uint32_t)R;
        stats->chunk_bytes = (uint32_t)chunk;
        stats->reader_seconds_reading = sh.ns_reading.load() * 1e-9;
        stats->reader_seconds_waiting = sh.ns_waiting.load() * 1e-9;
        stats->seconds_enqueueing = t_enqueue;
    }
    return rc;
}

// One file into device slot 0 through the same reader threads / staging ring / two copy streams as run_stream (the
// all-lines passes: --vcfAllPos and phase-1 site calling index the whole file before they look at a line, so there is
// nothing to overlap the copy with but the reads).  Synchronous.
int load_file(snpgpu_ctx *ctx, const char *path, uint8_t **d_file, uint64_t *size) {
    std::vector<Source> src(1);
    Source &s = src[0];
    s.path = path;
    s.fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat stt;
    if (s.fd < 0 || fstat(s.fd, &stt) != 0 || !S_ISREG(stt.st_mode)) {
        if (s.fd >= 0) close(s.fd);
        return snpgpu_set_error(ctx, SNPGPU_E_IO, "cannot open the pileup file %s", path);
    }
    const uint64_t n = s.size = (uint64_t)stt.st_size;
    (void)posix_fadvise(s.fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    const size_t chunk = (size_t)16 << 20;
    std::vector<Job> jobs;
    // the first pieces are small, so that the first copy starts after ~1 MiB has been read instead of after 16 (all readers
    // start at once and each takes a few milliseconds per 16 MiB)
    for (uint64_t off = 0, c = 0; off < n; ++c) {
        const uint64_t want = c < 2 ? (uint64_t)1 << 20 : (c < 5 ? (uint64_t)1 << (18 + c) : chunk);      // 1, 1, 2, 4, 8, 16, 16, ... MiB
        const uint64_t len = n - off < want ? n - off : want;
        jobs.push_back(Job{0, off, len, c == 0, off + len >= n, (uint32_t)c});
        off += len;
    }
    const uint64_t J = jobs.size();
    uint32_t n_readers = snpgpu_reader_threads();
    if (n_readers > J) n_readers = (uint32_t)J;
    uint32_t n_staging = n_readers + 4;
    if (n_staging > J) n_staging = (uint32_t)J;
    if (n_staging < 1) n_staging = 1;
    int rc = pool_ensure(ctx, chunk, n_staging, 1, up(n + SNPGPU_SCAN_TILE + 256, 4096), 256, 256);
    if (rc) { close(s.fd); return rc; }
    snpgpu_stream_pool *p = ctx->pool;
    const uint64_t R = p->staging.size() < n_staging ? p->staging.size() : n_staging;
    uint8_t *d = (uint8_t *)p->slot[0];
    hipError_t he = hipStreamSynchronize(ctx->stream);          // earlier work on the slot is done
    Shared sh;
    sh.R = R ? R : 1;
    sh.filled.assign(J, 0);
    sh.job_err.assign(J, 0);
    std::vector<std::thread> readers;
    bool io_failed = false;
    if (he == hipSuccess && J) {
        try {
            for (uint32_t i = 0; i < n_readers; ++i) readers.emplace_back(reader_main, ctx, &sh, &jobs, &src);
        } catch (const std::exception &e) {
            { std::lock_guard<std::mutex> lk(sh.mu); sh.abort = true; sh.next.store(J); }
            sh.cv.notify_all();
            for (auto &t : readers) t.join();
            close(s.fd);
            return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "cannot start the reader threads: %s", e.what());
        }
        int64_t copies_done = 0;
        for (uint64_t j = 0; j < J && he == hipSuccess; ++j) {
            for (;;) {                                          // wait for chunk j; retire finished copies meanwhile
                bool progress = false;
                while (copies_done < (int64_t)j && hipEventQuery(p->ev_copy[copies_done % R]) != hipErrorNotReady) { ++copies_done; progress = true; }   // (an error counts as done: the next HIP call reports it, nobody waits forever)
                std::unique_lock<std::mutex> lk(sh.mu);
                if (progress) { sh.freed = copies_done; lk.unlock(); sh.cv.notify_all(); lk.lock(); }
                if (sh.filled[j]) break;
                sh.cv.wait_for(lk, std::chrono::microseconds(copies_done < (int64_t)j ? 20 : 2000), [&] { return sh.filled[j] != 0; });
                if (sh.filled[j]) break;
            }
            if (sh.job_err[j]) io_failed = true;
            hipStream_t cs = (j & 1) ? p->copy_stream2 : p->copy_stream;
            he = hipMemcpyAsync(d + jobs[j].off, p->staging[j % R], jobs[j].len, hipMemcpyHostToDevice, cs);
            if (he == hipSuccess) he = hipEventRecord(p->ev_copy[j % R], cs);
        }
        { std::lock_guard<std::mutex> lk(sh.mu); if (he != hipSuccess) { sh.abort = true; sh.next.store(J); } }
        sh.cv.notify_all();
        for (auto &t : readers) t.join();
        hipError_t e1 = hipStreamSynchronize(p->copy_stream), e2 = hipStreamSynchronize(p->copy_stream2);
        if (he == hipSuccess) he = e1 != hipSuccess ? e1 : e2;
    }
    close(s.fd);
    if (he != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "loading %s failed: %s", path, hipGetErrorString(he));
    if (io_failed) return snpgpu_set_error(ctx, SNPGPU_E_IO, "cannot read the pileup file %s", path);
    *d_file = d;
    *size = n;
    return SNPGPU_OK;
}

int scan_status_error(snpgpu_ctx *ctx, const uint64_t *status, const char *what_file) {
    unsigned code = (unsigned)(status[0] & 0xFF);
    unsigned long long off = (unsigned long long)(status[0] >> 8) - 1;
    const char *what = code == SCAN_ERR_FEW_FIELDS ? "line has fewer than 2 fields" :
                       code == SCAN_ERR_BAD_POS ? "position field is not an unsigned decimal integer" :
                       code == SCAN_ERR_NON_ASCII ? "non-ASCII byte" : "malformed line";
    return snpgpu_set_error(ctx, code == SCAN_ERR_NON_ASCII ? SNPGPU_E_UNSUPPORTED : SNPGPU_E_PILEUP,
                            "pileup%s%s: %s at byte offset %llu", what_file ? " " : "", what_file ? what_file : "", what, off);
}

}  // namespace

extern "C" {

int snpgpu_call_consensus_files(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *const *paths, uint32_t n_files,
                                const snpgpu_caller_params *params, const uint32_t *excl_off, const uint32_t *excl_slots,
                                uint8_t *out_base, uint8_t *out_filters,
                                snpgpu_site_counts *out_counts, uint64_t *out_line_off, uint64_t *out_status,
                                int32_t *out_rc, const snpgpu_stream_opts *opts, snpgpu_stream_stats *stats) {
    if (!ctx || !ss || !params || !out_status || (n_files && !paths)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (ss->n_sites && n_files && (!out_base || !out_filters)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    std::vector<Source> src(n_files);
    for (uint32_t f = 0; f < n_files; ++f) {
        if (!paths[f]) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null path %u", f);
        src[f].path = paths[f];
    }
    Outputs out{out_base, out_filters, out_counts, out_line_off, out_status, out_rc};
    return run_stream(ctx, ss, src, params, out, opts, stats, nullptr, excl_off, excl_slots);
}

// Host-buffer form for ONE pileup (an mmap, bytes read elsewhere): same pipeline, the readers memcpy instead of pread.
// Synchronous.  Returns SNPGPU_E_PILEUP / SNPGPU_E_UNSUPPORTED when the scan found a malformed line (status words still filled).
int snpgpu_call_consensus(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const uint8_t *pileup, size_t nbytes,
                          const snpgpu_caller_params *params, uint8_t *out_base, uint8_t *out_filters,
                          snpgpu_site_counts *out_counts, uint64_t *out_status, int want_depth_sum) {
    if (!ctx || !ss || !params || !out_status || (nbytes && !pileup)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (ss->n_sites && (!out_base || !out_filters)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    static const uint8_t empty = 0;
    std::vector<Source> src(1);
    src[0].mem = nbytes ? pileup : &empty;
    src[0].size = nbytes;
    snpgpu_stream_opts o{};
    o.want_depth_sum = want_depth_sum ? 1u : 0u;
    Outputs out{out_base, out_filters, out_counts, nullptr, out_status, nullptr};
    int rc = run_stream(ctx, ss, src, params, out, &o, nullptr, ss->site_line);
    if (rc) return rc;
    if (out_status[0] != ~0ull) return scan_status_error(ctx, out_status, nullptr);
    return SNPGPU_OK;
}

}  // extern "C"

namespace {

// The all-lines pass on the device (--vcfAllPos): the file into device memory, its line index, a record per line.  What it leaves
// behind lives in the context's scratch (valid until the next call that takes scratch); nothing has been copied back but the number
// of lines.  With `compact`: also the 24-byte records and the count of the wide ones (lines_out.hip).
struct AllLines {
    uint8_t *d_file = nullptr;
    uint64_t nbytes = 0;
    uint32_t n_lines = 0;
    uint64_t *d_off = nullptr;
    uint8_t *d_flags = nullptr;
    snpgpu_site_counts *d_counts = nullptr;
    uint64_t *d_status = nullptr;
    snpgpu_line_record *d_recs = nullptr;       // compact only
    uint32_t *d_compact_ws = nullptr, *d_n_wide = nullptr;
    uint32_t *d_wide_index = nullptr;           // room for n_lines of each: the gather's targets
    snpgpu_site_counts *d_wide = nullptr;
};

// Returns SNPGPU_OK with al.n_lines set; when n_lines == 0 or n_lines > capacity nothing else has been done (al.d_counts == nullptr).
int all_lines_pass(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *path, const snpgpu_caller_params *params, uint64_t capacity, bool compact,
                   AllLines &al) {
    int rc = snpgpu_spill_begin(ctx);
    if (rc) return rc;
    rc = load_file(ctx, path, &al.d_file, &al.nbytes);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    const uint8_t *d_file = al.d_file;
    const uint64_t nbytes = al.nbytes;
    const size_t ws_words = snpgpu_lines_workspace_words(nbytes);
    // first the count alone (its workspace is all the scratch it needs) ...
    void *scr = nullptr;
    rc = snpgpu_scratch(ctx, up(4 * ws_words, 256) + 512, &scr);
    if (rc) return rc;
    uint32_t *d_total = nullptr;
    rc = snpgpu_enqueue_lines_count(ctx, d_file, nbytes, (uint32_t *)scr, &d_total);
    if (rc) return rc;
    uint32_t n_lines = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_lines, d_total, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    al.n_lines = n_lines;
    if (n_lines > capacity || n_lines == 0) return SNPGPU_OK;
    // ... then everything: the scratch may move, so the count is redone in the new place (cheap next to the call step)
    size_t o = up(4 * ws_words, 256);
    const size_t o_off = o; o += up(8ull * n_lines, 256);
    const size_t o_flag = o; o += up(n_lines, 256);
    const size_t o_base = o; o += up(n_lines, 256);
    const size_t o_filt = o; o += up(n_lines, 256);
    const size_t o_stat = o; o += 256;
    const size_t o_samp = o; o += 256;
    const size_t o_cnt = o; o += up(sizeof(snpgpu_site_counts) * (size_t)n_lines, 256);
    const size_t o_todo = o; o += 2 * up(8ull * n_lines, 256) + 256;           // the lane kernels' leftover lists and their counts
    const size_t o_rec = o; if (compact) o += up(sizeof(snpgpu_line_record) * (size_t)n_lines, 256);
    const size_t o_cws = o; if (compact) o += up(4 * snpgpu_compact_lines_workspace_words(n_lines), 256);
    const size_t o_widx = o; if (compact) o += up(4ull * n_lines, 256);
    const size_t o_wide = o; if (compact) o += up(sizeof(snpgpu_site_counts) * (size_t)n_lines, 256);     // (every line may be wide; 288 GB of HBM)
    rc = snpgpu_scratch(ctx, o + 256, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    rc = snpgpu_enqueue_lines_count(ctx, d_file, nbytes, (uint32_t *)b, &d_total);
    if (rc) return rc;
    uint64_t h_status[SNPGPU_SCAN_STATUS_WORDS] = {~0ull, n_lines, 0, 0};
    SampleDev sd{};
    sd.buf = d_file;
    sd.nbytes = nbytes;
    sd.status = (uint64_t *)(b + o_stat);
    HIP_TRY(ctx, hipMemcpyAsync(b + o_stat, h_status, sizeof h_status, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(b + o_samp, &sd, sizeof sd, hipMemcpyHostToDevice, st));
    rc = snpgpu_enqueue_lines_emit(ctx, ss, d_file, nbytes, (uint32_t *)b, (uint64_t *)(b + o_off), (uint8_t *)(b + o_flag), n_lines,
                                   (uint64_t *)(b + o_stat));
    if (rc == SNPGPU_OK)
        rc = snpgpu_enqueue_call_lines(ctx, (const SampleDev *)(b + o_samp), (const uint64_t *)(b + o_off), (const uint8_t *)(b + o_flag),
                                       n_lines, params, (uint8_t *)(b + o_base), (uint8_t *)(b + o_filt), (snpgpu_site_counts *)(b + o_cnt),
                                       (uint32_t *)(b + o_todo), (uint64_t *)(b + o_todo + 256), (uint64_t *)(b + o_todo + 256 + up(8ull * n_lines, 256)));
    if (rc) return rc;
    al.d_off = (uint64_t *)(b + o_off);
    al.d_flags = (uint8_t *)(b + o_flag);
    al.d_counts = (snpgpu_site_counts *)(b + o_cnt);
    al.d_status = (uint64_t *)(b + o_stat);
    if (compact) {
        al.d_recs = (snpgpu_line_record *)(b + o_rec);
        al.d_compact_ws = (uint32_t *)(b + o_cws);
        al.d_wide_index = (uint32_t *)(b + o_widx);
        al.d_wide = (snpgpu_site_counts *)(b + o_wide);
        rc = snpgpu_enqueue_compact_lines(ctx, al.d_counts, al.d_flags, n_lines, al.d_recs, al.d_compact_ws, &al.d_n_wide);
    }
    return rc;
}

}  // namespace

extern "C" {

// call_consensus --vcfAllPos (call_consensus.py:148-151, pileup.py:418-421): a Record for EVERY line of the pileup.
// Synchronous, host outputs in file order: out_line_off[i] = 1 + byte offset of line i, out_line_flags[i] = SNPGPU_SITE_*
// of its position (0 when it is not in the site set), out_counts[i] its record.  *out_n_lines is always set; when it
// exceeds `capacity` nothing else is written and the caller comes back with larger arrays.  out_status: scan status words
// ([0] = first line whose chrom / position columns are malformed, [1] = number of lines).
int snpgpu_call_all_lines_file(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *path, const snpgpu_caller_params *params,
                               uint64_t capacity, uint64_t *out_n_lines, uint64_t *out_line_off, uint8_t *out_line_flags,
                               snpgpu_site_counts *out_counts, uint64_t *out_status) {
    if (!ctx || !ss || !path || !params || !out_n_lines || !out_status) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (capacity && (!out_line_off || !out_line_flags || !out_counts)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    AllLines al;
    int rc = all_lines_pass(ctx, ss, path, params, capacity, false, al);
    if (rc) return rc;
    const uint32_t n_lines = al.n_lines;
    *out_n_lines = n_lines;
    out_status[0] = ~0ull; out_status[1] = n_lines; out_status[2] = out_status[3] = 0;
    if (!al.d_counts) return SNPGPU_OK;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(out_line_off, al.d_off, 8ull * n_lines, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(out_line_flags, al.d_flags, n_lines, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(out_counts, al.d_counts, sizeof(snpgpu_site_counts) * (size_t)n_lines, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(out_status, al.d_status, 8 * SNPGPU_SCAN_STATUS_WORDS, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    out_status[1] = n_lines;
    if (out_status[0] != ~0ull) return scan_status_error(ctx, out_status, path);
    return SNPGPU_OK;
}

// The same with 24-byte records: see include/snpgpu.h.
int snpgpu_call_all_lines_compact_file(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *path, const snpgpu_caller_params *params,
                                       uint64_t capacity, uint64_t *out_n_lines, uint64_t *out_line_off, snpgpu_line_record *out_records,
                                       uint32_t wide_capacity, uint32_t *out_n_wide, uint32_t *out_wide_index, snpgpu_site_counts *out_wide,
                                       uint64_t *out_status) {
    if (!ctx || !ss || !path || !params || !out_n_lines || !out_n_wide || !out_status) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (capacity && (!out_line_off || !out_records)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    if (wide_capacity && (!out_wide_index || !out_wide)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    AllLines al;
    int rc = all_lines_pass(ctx, ss, path, params, capacity, true, al);
    if (rc) return rc;
    const uint32_t n_lines = al.n_lines;
    *out_n_lines = n_lines;
    *out_n_wide = 0;
    out_status[0] = ~0ull; out_status[1] = n_lines; out_status[2] = out_status[3] = 0;
    if (!al.d_counts) return SNPGPU_OK;
    hipStream_t st = ctx->stream;
    uint32_t n_wide = 0;
    // the bulk goes while the host learns how many wide lines there are
    HIP_TRY(ctx, hipMemcpyAsync(&n_wide, al.d_n_wide, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(out_status, al.d_status, 8 * SNPGPU_SCAN_STATUS_WORDS, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *out_n_wide = n_wide;
    out_status[1] = n_lines;
    if (out_status[0] != ~0ull) return scan_status_error(ctx, out_status, path);
    if (n_wide > wide_capacity) return SNPGPU_OK;                 // the caller comes back with room for them
    HIP_TRY(ctx, hipMemcpyAsync(out_line_off, al.d_off, 8ull * n_lines, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(out_records, al.d_recs, sizeof(snpgpu_line_record) * (size_t)n_lines, hipMemcpyDeviceToHost, st));
    if (n_wide) {
        rc = snpgpu_enqueue_gather_wide(ctx, al.d_counts, al.d_recs, n_lines, al.d_compact_ws, n_wide, al.d_wide_index, al.d_wide);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(out_wide_index, al.d_wide_index, 4ull * n_wide, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_wide, al.d_wide, sizeof(snpgpu_site_counts) * (size_t)n_wide, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return SNPGPU_OK;
}

// --vcfAllPos from file to file: see include/snpgpu.h.  The records come back in pieces into pinned memory while host threads format the
// rows of the pieces that have landed; the pileup's own text (CHROM and POS of every row) is read through a private mapping of the file,
// which the load has just pulled into the page cache.
int snpgpu_write_all_positions_vcf(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const char *pileup_path, const snpgpu_caller_params *params,
                                   const char *vcf_path, const char *header, const char *const *filter_names, int preserve_ref_case,
                                   char failed_snp_gt, int only_listed, int check, uint64_t *out_n_lines, uint64_t *out_n_rows,
                                   uint64_t *out_first_bad_line, uint64_t *out_first_bad_off, snpgpu_site_counts *out_first_bad, uint64_t *out_status) {
    if (!ctx || !ss || !pileup_path || !params || !vcf_path || !header || !filter_names || !out_n_lines || !out_n_rows || !out_first_bad_line ||
        !out_first_bad_off || !out_first_bad || !out_status)
        return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    *out_n_lines = *out_n_rows = 0;
    *out_first_bad_line = ~0ull;
    *out_first_bad_off = 0;
    hipStream_t st = ctx->stream;
    AllLines al;
    uint32_t n_lines = 0, n_wide = 0, n_spill = 0;
    bool scan_bad = false;
    std::vector<snpgpu_symbol_spill> spill;
    for (int attempt = 0;; ++attempt) {                          // (again when the positions asked for more spill records than the arena held)
        al = AllLines();
        int rc = all_lines_pass(ctx, ss, pileup_path, params, ~0ull, true, al);
        if (rc) return rc;
        n_lines = al.n_lines;
        *out_n_lines = n_lines;
        out_status[0] = ~0ull; out_status[1] = n_lines; out_status[2] = out_status[3] = 0;
        if (!al.d_counts) break;                                 // an empty file: the header alone
        HIP_TRY(ctx, hipMemcpyAsync(&n_wide, al.d_n_wide, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&n_spill, ctx->d_spill_n, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_status, al.d_status, 8 * SNPGPU_SCAN_STATUS_WORDS, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        out_status[1] = n_lines;
        scan_bad = out_status[0] != ~0ull;       // (reported below, after a look for an earlier line that a Record cannot be built from)
        if (scan_bad || n_spill <= ctx->spill_cap) break;
        if (attempt >= 4 || n_spill > 0xFFFFFEu) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "%u lines of %s need a spill record", n_spill, pileup_path);
        const uint64_t want = (uint64_t)n_spill + n_spill / 4 + 64;
        ctx->spill_want = want > 0xFFFFFEull ? 0xFFFFFEu : (uint32_t)want;
    }
    // the wide lines (few) and the spill records they point at: all at once
    std::vector<uint32_t> wide_index(n_wide);
    std::vector<snpgpu_site_counts> wide(n_wide);
    if (al.d_counts && n_wide) {
        int rc = snpgpu_enqueue_gather_wide(ctx, al.d_counts, al.d_recs, n_lines, al.d_compact_ws, n_wide, al.d_wide_index, al.d_wide);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(wide_index.data(), al.d_wide_index, 4ull * n_wide, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(wide.data(), al.d_wide, sizeof(snpgpu_site_counts) * (size_t)n_wide, hipMemcpyDeviceToHost, st));
        if (n_spill > ctx->spill_cap) n_spill = ctx->spill_cap;
        spill.resize(n_spill);
        if (n_spill) HIP_TRY(ctx, hipMemcpyAsync(spill.data(), ctx->d_spill, sizeof(snpgpu_symbol_spill) * (size_t)n_spill, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (check)
            for (uint32_t k = 0; k < n_wide; ++k)
                if (wide[k].status > SNPGPU_ST_OK) {             // the first line a Record cannot be built from: the caller raises, nothing is written
                    *out_first_bad_line = wide_index[k];
                    *out_first_bad = wide[k];
                    HIP_TRY(ctx, hipMemcpy(out_first_bad_off, al.d_off + wide_index[k], 8, hipMemcpyDeviceToHost));
                    break;
                }
    }
    if (scan_bad) return scan_status_error(ctx, out_status, pileup_path);
    if (*out_first_bad_line != ~0ull) return SNPGPU_OK;
    // the pileup's text for CHROM / POS
    const uint8_t *text = nullptr;
    int pfd = -1;
    if (n_lines) {
        pfd = open(pileup_path, O_RDONLY | O_CLOEXEC);
        void *m = pfd >= 0 ? mmap(nullptr, al.nbytes, PROT_READ, MAP_PRIVATE, pfd, 0) : MAP_FAILED;
        if (m == MAP_FAILED) { if (pfd >= 0) close(pfd); return snpgpu_set_error(ctx, SNPGPU_E_IO, "cannot map the pileup file %s", pileup_path); }
        text = (const uint8_t *)m;
    }
    const int vfd = open(vcf_path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    std::atomic<bool> io_ok{vfd >= 0};
    const size_t header_len = strlen(header);
    // pieces of lines: offsets + records of a piece come back into one pinned buffer each (two copies on the stream, an event behind
    // them); a pool of threads takes the pieces in order, formats, and writes its text at the offset the pieces before it end at
    const uint64_t piece = 1u << 16;                              // 65 536 lines: 2.5 MiB of offsets and records
    const uint64_t n_pieces = (n_lines + piece - 1) / piece;
    const size_t part_bytes = (size_t)4 << 20;                    // the load's pinned staging ring (16 MiB buffers), in parts of 4 MiB
    static_assert((size_t)(1u << 16) * (8 + sizeof(snpgpu_line_record)) <= ((size_t)4 << 20), "a piece fits a part");
    snpgpu_stream_pool *pool_ = ctx->pool;
    const uint32_t parts_per = pool_ ? (uint32_t)(pool_->chunk_bytes / part_bytes) : 0;
    const uint32_t parts = pool_ ? (uint32_t)pool_->staging.size() * parts_per : 0;
    if (n_pieces && parts < 2) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "no pinned staging memory for the records");
    const uint32_t T = n_pieces ? (uint32_t)std::min<uint64_t>(std::min<uint32_t>(snpgpu_cpu_threads(64), parts - 1), n_pieces) : 0;
    const uint32_t R = n_pieces ? (uint32_t)std::min<uint64_t>(std::min<uint32_t>(T + 2, parts), n_pieces) : 0;          // parts in flight
    std::vector<void *> pinned(R, nullptr);
    std::vector<hipEvent_t> ev(R, nullptr);
    hipError_t he = hipSuccess;
    for (uint32_t k = 0; k < R && he == hipSuccess; ++k) {
        pinned[k] = (char *)pool_->staging[k / parts_per] + (size_t)(k % parts_per) * part_bytes;
        he = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
    }
    struct Piece { std::vector<char> text; uint64_t rows = 0; bool done = false, bad = false; uint64_t bad_line = 0; };
    std::vector<Piece> out(n_pieces);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> landed(n_pieces, 0);                       // 1: its copies were enqueued (the event tells when they are done)
    std::vector<int> buffer_free(R, 1);
    std::atomic<uint64_t> next_piece{0};
    uint64_t written_upto = 0, file_off = header_len;           // (under mu) pieces whose text is in the file
    bool abort_all = he != hipSuccess;
    auto worker = [&]() {
        (void)hipSetDevice(ctx->device);
        for (;;) {
            const uint64_t pi = next_piece.fetch_add(1);
            if (pi >= n_pieces) return;
            const uint32_t k = (uint32_t)(pi % R);
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return landed[pi] || abort_all; }); if (abort_all) return; }
            if (hipEventSynchronize(ev[k]) != hipSuccess) { std::lock_guard<std::mutex> lk(mu); abort_all = true; cv.notify_all(); return; }
            const uint64_t lo = pi * piece, hi = std::min<uint64_t>(n_lines, lo + piece);
            const uint64_t *off = (const uint64_t *)pinned[k];
            const snpgpu_line_record *recs = (const snpgpu_line_record *)((const char *)pinned[k] + 8 * piece);
            Piece &pc = out[pi];
            pc.bad = !snpgpu_line_rows_into(text, al.nbytes, off, recs, lo, lo, hi, wide_index.data(), wide.data(), n_wide, filter_names, preserve_ref_case,
                                              failed_snp_gt, spill.data(), n_spill, only_listed, pc.text, &pc.rows, &pc.bad_line);
            std::unique_lock<std::mutex> lk(mu);
            buffer_free[k] = 1;
            pc.done = true;
            cv.notify_all();
            // the file is written in piece order by whoever finishes the piece that is next in line
            while (written_upto < n_pieces && out[written_upto].done && !abort_all) {
                Piece &w = out[written_upto];
                const uint64_t at = file_off;
                file_off += w.text.size();
                const uint64_t me = written_upto++;
                lk.unlock();
                if (io_ok && !w.bad) {
                    const char *p = w.text.data();
                    size_t left = w.text.size();
                    uint64_t pos = at;
                    while (left) {
                        const ssize_t wr = pwrite(vfd, p, left, (off_t)pos);
                        if (wr < 0) { if (errno == EINTR) continue; io_ok = false; break; }
                        p += wr; left -= (size_t)wr; pos += (uint64_t)wr;
                    }
                }
                std::vector<char>().swap(out[me].text);
                lk.lock();
            }
        }
    };
    std::vector<std::thread> pool;
    if (!abort_all) for (uint32_t t = 0; t < T; ++t) pool.emplace_back(worker);
    if (io_ok && header_len) {
        size_t left = header_len; const char *p = header; uint64_t pos = 0;
        while (left) { const ssize_t wr = pwrite(vfd, p, left, (off_t)pos); if (wr < 0) { if (errno == EINTR) continue; io_ok = false; break; } p += wr; left -= (size_t)wr; pos += (uint64_t)wr; }
    }
    for (uint64_t pi = 0; pi < n_pieces && !abort_all; ++pi) {
        const uint32_t k = (uint32_t)(pi % R);
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return buffer_free[k] || abort_all; }); if (abort_all) break; buffer_free[k] = 0; }
        const uint64_t lo = pi * piece, hi = std::min<uint64_t>(n_lines, lo + piece);
        he = hipMemcpyAsync(pinned[k], al.d_off + lo, 8 * (hi - lo), hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipMemcpyAsync((char *)pinned[k] + 8 * piece, al.d_recs + lo, sizeof(snpgpu_line_record) * (hi - lo), hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipEventRecord(ev[k], st);
        std::lock_guard<std::mutex> lk(mu);
        if (he != hipSuccess) abort_all = true; else landed[pi] = 1;
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
    for (auto &t : pool) t.join();
    (void)hipStreamSynchronize(st);
    for (uint32_t k = 0; k < R; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]);
    if (text) { munmap((void *)text, al.nbytes); close(pfd); }
    uint64_t rows = 0, bad_line = ~0ull;
    for (auto &pc : out) { rows += pc.rows; if (pc.bad && bad_line == ~0ull) bad_line = pc.bad_line; }
    if (vfd >= 0) {
        if (io_ok && ftruncate(vfd, (off_t)file_off) != 0) io_ok = false;
        if (close(vfd) != 0) io_ok = false;
    }
    if (he != hipSuccess || abort_all) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "reading the records of %s back failed: %s", pileup_path, hipGetErrorString(he));
    if (bad_line != ~0ull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "line %llu of %s has more symbols than a record keeps and no spill record", (unsigned long long)bad_line, pileup_path);
    if (!io_ok) return snpgpu_set_error(ctx, SNPGPU_E_IO, "cannot write %s", vcf_path);
    *out_n_rows = rows;
    return SNPGPU_OK;
}

}  // extern "C"

namespace {

// Phase-1 site calling over a pileup that is in device memory: one pass over the text (line starts + select) and a walk over the
// few candidate lines on the device (varscan.hip), records back in file order.  Synchronous.
int varscan_resident(snpgpu_ctx *ctx, const uint8_t *d_file, uint64_t nbytes, const char *what, const snpgpu_varscan_params *params,
                     uint32_t capacity, snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status) {
    hipStream_t st = ctx->stream;
    *out_n_sites = 0;
    out_status[0] = ~0ull; out_status[1] = 0;
    if (nbytes == 0) return SNPGPU_OK;
    size_t o = 0;
    const size_t o_ctl = o; o += 256;                          // [0] u64 status, then the kernels' eight control words
    const size_t o_rec = o; o += up(sizeof(snpgpu_varscan_site) * (size_t)capacity, 256);
    const size_t o_var = o; o += up(snpgpu_varscan_scratch_bytes(nbytes), 256);
    void *scr = nullptr;
    int rc = snpgpu_scratch(ctx, o + 256, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    uint64_t h_ctl[5] = {~0ull, 0, 0, 0, 0};                    // status; records found + candidates; long candidates + spare; lines; spare
    HIP_TRY(ctx, hipMemcpyAsync(b + o_ctl, h_ctl, sizeof h_ctl, hipMemcpyHostToDevice, st));
    rc = snpgpu_enqueue_varscan(ctx, d_file, nbytes, params, (snpgpu_varscan_site *)(b + o_rec), capacity, (uint32_t *)(b + o_ctl + 8), (uint64_t *)(b + o_ctl),
                                b + o_var);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_ctl, b + o_ctl, sizeof h_ctl, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
#ifdef SNPGPU_TUNING
    if (getenv("SNPGPU_VARSCAN_DEBUG"))
        fprintf(stderr, "varscan: %llu lines, %u candidates on the global list\n", (unsigned long long)h_ctl[3], (uint32_t)(h_ctl[1] >> 32));
#endif
    out_status[0] = h_ctl[0];
    out_status[1] = h_ctl[3];
    const uint32_t found = (uint32_t)h_ctl[1];
    *out_n_sites = found;
    if (h_ctl[0] != ~0ull)
        return snpgpu_set_error(ctx, SNPGPU_E_PILEUP, "malformed pileup line at byte %llu of %s", (unsigned long long)h_ctl[0], what);
    const uint32_t got = found < capacity ? found : capacity;
    if (got) {
        HIP_TRY(ctx, hipMemcpyAsync(out_sites, b + o_rec, sizeof(snpgpu_varscan_site) * (size_t)got, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (found <= capacity)
            std::sort(out_sites, out_sites + got, [](const snpgpu_varscan_site &x, const snpgpu_varscan_site &y) {
                return x.line_off != y.line_off ? x.line_off < y.line_off : x.alt_base < y.alt_base;
            });
    }
    return SNPGPU_OK;
}

// The same for many resident pileups at once: ONE scan launch over all of them (varscan.hip), one copy of the control words back,
// then the records of every file.  Synchronous.  out_sites [n_files][capacity], out_n_sites [n_files], out_status [n_files][2],
// out_rc [n_files].
int varscan_resident_batch(snpgpu_ctx *ctx, const void *const *d_pileups, const uint64_t *nbytes, uint32_t n_files, const snpgpu_varscan_params *params,
                           uint32_t capacity, snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status, int32_t *out_rc) {
    hipStream_t st = ctx->stream;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_files; ++i) {
        if (nbytes[i] && !d_pileups[i]) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null pileup %u", i);
        total += nbytes[i];
        out_n_sites[i] = 0; out_status[2 * i] = ~0ull; out_status[2 * i + 1] = 0; out_rc[i] = SNPGPU_OK;
    }
    size_t o = 0;
    const size_t o_ctl = o; o += up(40 * (size_t)n_files, 256);                      // per file: u64 status, eight control words
    const size_t o_rec = o; o += up(sizeof(snpgpu_varscan_site) * (size_t)capacity * n_files, 256);
    const size_t o_var = o; o += up(snpgpu_varscan_batch_scratch_bytes(total, n_files), 256);
    void *scr = nullptr;
    int rc = snpgpu_scratch(ctx, o + 256, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    // the status words first (n_files u64), then the control words (n_files x 8 u32)
    std::vector<uint64_t> h_ctl(5 * (size_t)n_files, 0);
    for (uint32_t i = 0; i < n_files; ++i) h_ctl[i] = ~0ull;
    std::vector<char> h_table(snpgpu_varscan_table_bytes(n_files));
    HIP_TRY(ctx, hipMemcpyAsync(b + o_ctl, h_ctl.data(), 40 * (size_t)n_files, hipMemcpyHostToDevice, st));
    rc = snpgpu_enqueue_varscan_batch(ctx, (const uint8_t *const *)d_pileups, nbytes, n_files, params, (snpgpu_varscan_site *)(b + o_rec), capacity,
                                      (uint32_t *)(b + o_ctl + 8 * (size_t)n_files), (uint64_t *)(b + o_ctl), b + o_var, h_table.data());
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    HIP_TRY(ctx, hipMemcpyAsync(h_ctl.data(), b + o_ctl, 40 * (size_t)n_files, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));                   // (h_table and h_ctl have been read by then)
    const uint32_t *words = (const uint32_t *)(h_ctl.data() + n_files);
    for (uint32_t i = 0; i < n_files; ++i) {
        const uint32_t *w = words + 8 * (size_t)i;
        uint64_t lines;
        memcpy(&lines, w + 4, 8);
        out_status[2 * i] = h_ctl[i];
        out_status[2 * i + 1] = lines;
        out_n_sites[i] = w[0];
        if (h_ctl[i] != ~0ull) { out_rc[i] = SNPGPU_E_PILEUP; continue; }
        const uint32_t got = w[0] < capacity ? w[0] : capacity;
        if (got) HIP_TRY(ctx, hipMemcpyAsync(out_sites + (size_t)i * capacity, b + o_rec + sizeof(snpgpu_varscan_site) * (size_t)capacity * i,
                                             sizeof(snpgpu_varscan_site) * (size_t)got, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n_files; ++i) {
        if (out_rc[i] != SNPGPU_OK || out_n_sites[i] > capacity) continue;           // (more records than the array holds: the caller repeats that file alone)
        snpgpu_varscan_site *dst = out_sites + (size_t)i * capacity;
        std::sort(dst, dst + out_n_sites[i], [](const snpgpu_varscan_site &x, const snpgpu_varscan_site &y) {
            return x.line_off != y.line_off ? x.line_off < y.line_off : x.alt_base < y.alt_base;
        });
    }
    return SNPGPU_OK;
}

// Device memory for the files of one ingest call that are to stay resident: whole files, 256-byte aligned, bump-allocated out
// of blocks of at most `block_cap` bytes (a block is allocated right before the first copy into it).  Files are placed in
// order while the budget lasts; from the first one that does not fit on, they stay non-resident.
struct Placement { uint32_t block; uint64_t off; };
const uint64_t PILEUP_BLOCK_CAP = 8ull << 30;
const uint64_t PILEUP_TAIL_PAD = SNPGPU_SCAN_TILE + 512;       // the scan reads whole tiles (+ halo) past a file's last byte

// What the readers and the issuing thread share when pieces may be copied in ANY order (resident files: every piece has its
// own place in device memory, so nobody has to wait for the slowest read — with the in-order ring one slow pread stalls the
// copy engine and, a few pieces later, every other reader).
struct AnyOrder {
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    std::vector<uint32_t> free_bufs;
    struct Ready { uint64_t job; int32_t buf; int err; };
    std::deque<Ready> ready;
    bool abort = false;
    std::atomic<uint64_t> next{0};
    uint64_t n_jobs = 0;
    std::atomic<uint64_t> ns_reading{0}, ns_waiting{0};
    Opener *opener = nullptr;
};

void reader_any_order(snpgpu_ctx *ctx, AnyOrder *sh, const std::vector<Job> *jobs, std::vector<Source> *src) {
    snpgpu_stream_pool *p = ctx->pool;
    if (p->have_near_cpus) (void)sched_setaffinity(0, sizeof p->near_cpus, &p->near_cpus);
    for (;;) {
        const uint64_t j = sh->next.fetch_add(1);
        if (j >= sh->n_jobs) return;
        const Job &jb = (*jobs)[j];
        Source &s = (*src)[jb.file];
        int32_t buf = -1;
        int err = 0;
        if (jb.len) {
            const double t_w = now_s();
            if (sh->opener) sh->opener->wait(jb.file);
            {
                std::unique_lock<std::mutex> lk(sh->mu);
                sh->cv_free.wait(lk, [&] { return sh->abort || !sh->free_bufs.empty(); });
                if (sh->abort) return;
                buf = (int32_t)sh->free_bufs.back();
                sh->free_bufs.pop_back();
            }
            const double t_r = now_s();
            uint8_t *dst = (uint8_t *)p->staging[buf];
            if (s.fd >= 0) {
                uint64_t got = 0;
                while (got < jb.len) {
                    ssize_t r = pread(s.fd, dst + got, jb.len - got, (off_t)(jb.off + got));
                    if (r < 0) { if (errno == EINTR) continue; err = errno ? errno : EIO; break; }
                    if (r == 0) { err = EIO; break; }
                    got += (uint64_t)r;
                }
                if (err) memset(dst + got, '\n', jb.len - got);
            } else {
                memset(dst, '\n', jb.len);
            }
            sh->ns_waiting.fetch_add((uint64_t)((t_r - t_w) * 1e9));
            sh->ns_reading.fetch_add((uint64_t)((now_s() - t_r) * 1e9));
        }
        if (sh->opener) sh->opener->done_reading(jb.file);
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            sh->ready.push_back(AnyOrder::Ready{j, buf, err});
        }
        sh->cv_ready.notify_one();
    }
}

// Phase-1 site calling over many pileup files: the readers run ahead across file boundaries, and a file's kernels and result
// copy run on the compute stream while the next files are being read and copied.  Without a store the files alternate between
// two device slots and their pieces are copied in file order; with one, files go to resident memory while the store's budget
// lasts (entry store->files[first + f]), their pieces copied in whatever order the reads finish, and only the files past the
// budget use the slots.  params == nullptr: no site calling, the files are only made resident.  out_done (nullable): out_done[f]
// becomes 1 when the outputs of file f are final, so that another thread can start on them while the call is still running.
int varscan_stream(snpgpu_ctx *ctx, const char *const *paths, uint32_t n_files, const snpgpu_varscan_params *params, uint32_t capacity,
                   snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status, int32_t *out_rc, snpgpu_pileups *store,
                   int32_t *out_done) {
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const double t_enter = now_s();
    // pieces of 32 MiB: 53-54 GB/s of files into resident memory where 16 MiB pieces give 50-52 (fewer, larger preads and copies;
    // 64 MiB: 46 — the pipeline of 12 staging buffers gets too coarse); tools/pipeline_time.py with SNPGPU_INGEST_CHUNK_MIB
    size_t chunk = (size_t)32 << 20;
#ifdef SNPGPU_TUNING                                            // development builds only (tools/)
    if (const char *e = getenv("SNPGPU_INGEST_CHUNK_MIB")) if (atoi(e) > 0) chunk = (size_t)atoi(e) << 20;
#endif
    std::vector<Source> src(n_files);
    uint64_t max_slot_size = 0;
    const size_t first = store ? store->files.size() : 0;
    if (store) store->files.resize(first + n_files);
    std::vector<Placement> place(n_files, Placement{~0u, 0});
    std::vector<uint64_t> block_bytes;                          // blocks of this call
    uint32_t n_prefix = 0;                                      // files [0, n_prefix) are resident (or have nothing to copy)
    bool budget_left = store != nullptr;
    for (uint32_t f = 0; f < n_files; ++f) {
        Source &s = src[f];
        s.path = paths[f];
        struct stat stt;                                        // (the file itself is opened later, by the opener thread)
        if (stat(s.path, &stt) != 0 || !S_ISREG(stt.st_mode)) {
            s.rc = SNPGPU_E_IO;
            s.size = 0;
        } else {
            s.size = (uint64_t)stt.st_size;
        }
        bool resident = false;
        if (budget_left && s.rc == SNPGPU_OK) {
            const uint64_t need = up(s.size + 1, 256);
            if (store->used + need + PILEUP_TAIL_PAD <= store->budget) {
                // (the first blocks are small — 1, 2, 4 GiB, then 8 — so that the first copy does not wait for a large allocation)
                const uint64_t cap_now = block_bytes.size() <= 3 ? (PILEUP_BLOCK_CAP >> (4 - (block_bytes.size() ? block_bytes.size() : 1))) : PILEUP_BLOCK_CAP;
                if (block_bytes.empty() || block_bytes.back() + need + PILEUP_TAIL_PAD > cap_now) { block_bytes.push_back(0); store->used += PILEUP_TAIL_PAD; }
                place[f] = Placement{(uint32_t)(block_bytes.size() - 1), block_bytes.back()};
                block_bytes.back() += need;
                store->used += need;
                resident = true;
            } else {
                budget_left = false;
            }
        }
        if (store && n_prefix == f && (resident || s.rc != SNPGPU_OK || s.size == 0)) n_prefix = f + 1;
        if (store) {
            store->files[first + f].nbytes = s.size;
            store->files[first + f].resident = resident;
            store->file_bytes += s.size;
        }
        if (!resident && s.size > max_slot_size) max_slot_size = s.size;
        out_n_sites[f] = 0;
        out_status[2 * f] = ~0ull; out_status[2 * f + 1] = 0;
        if (out_done) out_done[f] = 0;
    }
    for (auto &b : block_bytes) b += PILEUP_TAIL_PAD;
    [[maybe_unused]] const double t_opened = now_s();
    std::vector<uint8_t *> block_ptr(block_bytes.size(), nullptr);
    std::vector<uint8_t> block_seen(block_bytes.size(), 0);     // the issuing thread has waited for the block's allocation
    std::vector<Job> jobs;
    std::vector<uint32_t> chunks_left(n_files, 0);
    uint64_t JA = 0;                                            // jobs [0, JA): files of the prefix; [JA, J): the rest, in file order
    for (uint32_t f = 0; f < n_files; ++f) {
        const uint64_t n = src[f].size;
        if (f == n_prefix) JA = jobs.size();
        if (n == 0) { jobs.push_back(Job{f, 0, 0, true, true, 0}); chunks_left[f] = 1; continue; }
        for (uint64_t off = 0, c = 0; off < n; ++c) {
            // the very first pieces are small, so that the first copy starts after ~1 MiB has been read
            const uint64_t want = f == 0 && c < 5 ? (c < 2 ? (uint64_t)1 << 20 : (uint64_t)1 << (18 + c)) : chunk;
            const uint64_t len = n - off < want ? n - off : want;
            jobs.push_back(Job{f, off, len, c == 0, off + len >= n, (uint32_t)c});
            ++chunks_left[f];
            off += len;
        }
    }
    const uint64_t J = jobs.size();
    if (n_prefix == n_files) JA = J;
    uint32_t n_readers = snpgpu_reader_threads();
    uint32_t extra_staging = 4;
#ifdef SNPGPU_TUNING                                            // development builds only (tools/)
    if (const char *e = getenv("SNPGPU_INGEST_READERS")) if (atoi(e) > 0) n_readers = (uint32_t)atoi(e);
    if (const char *e = getenv("SNPGPU_INGEST_EXTRA_STAGING")) if (atoi(e) >= 0) extra_staging = (uint32_t)atoi(e);
#endif
    if (n_readers > J) n_readers = (uint32_t)J;
    if (store) store->n_readers = n_readers;
    uint32_t n_staging = n_readers + extra_staging;
    if (n_staging > J) n_staging = (uint32_t)J;
    if (n_staging < 1) n_staging = 1;
    const size_t r_rec = 256;                                  // result block: [0] u64 status, [8] u32 records found, [48] u32 lines; records at 256
    const size_t result_bytes = r_rec + sizeof(snpgpu_varscan_site) * (size_t)capacity + 256;
    Opener opener;
    auto close_all = [&]() { opener.finish(); for (auto &s2 : src) if (s2.fd >= 0) { close(s2.fd); s2.fd = -1; } };
    opener.start(&src, chunks_left);                            // (the first files open while the staging memory is being pinned)
    const uint32_t n_slots = n_files > 1 ? 2 : 1;              // result blocks / scratch halves (/ device slots) alternate between files
    const bool any_slot = n_prefix < n_files;
    int rc = pool_ensure(ctx, chunk, n_staging, any_slot ? n_slots : 0, any_slot ? up(max_slot_size + SNPGPU_SCAN_TILE + 256, 4096) : 0,
                         result_bytes, 256, n_slots);
    if (rc) { close_all(); return rc; }
    snpgpu_stream_pool *p = ctx->pool;
    const uint64_t R = p->staging.size() < n_staging ? p->staging.size() : n_staging;
    hipStream_t st = ctx->stream;
    [[maybe_unused]] const double t_pool = now_s();
    {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { close_all(); return snpgpu_set_error(ctx, SNPGPU_E_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e)); }
    }
#ifdef SNPGPU_TUNING
    if (getenv("SNPGPU_VARSCAN_DEBUG"))
        fprintf(stderr, "ingest prepare: files sized and placed %.4f s, staging pool %.4f s, stream idle %.4f s\n", t_opened - t_enter,
                t_pool - t_opened, now_s() - t_pool);
#endif
    // Files are post-processed in the order in which they become complete ("sequence numbers"); result blocks, scratch halves
    // and device slots alternate by sequence number.
    std::vector<int64_t> seq_of(n_files, -1);
    std::vector<uint32_t> file_of;
    file_of.reserve(n_files);
    auto dest = [&](uint32_t f) -> uint8_t * {
        if (place[f].block == ~0u) return (uint8_t *)p->slot[(uint32_t)seq_of[f] % n_slots];
        return block_ptr[place[f].block] + place[f].off;
    };
    std::vector<uint8_t> has_result(n_files, 0), completed(n_files, 0);   // by sequence number
    const double t_begin = now_s();
    double t_alloc = 0, t_read_wait = 0, t_dev_wait = 0;
    uint64_t ns_reading = 0, ns_waiting = 0;
    auto publish = [&](uint32_t f) { if (out_done) __atomic_store_n(&out_done[f], 1, __ATOMIC_RELEASE); };
    auto harvest = [&](uint32_t c) -> int {                     // results of the c-th completed file: pinned block -> the caller's arrays
        const uint32_t f = file_of[c];
        if (!has_result[c]) { out_rc[f] = src[f].rc; publish(f); return SNPGPU_OK; }
        const uint32_t slot = c % n_slots;
        const double tw = now_s();
        hipError_t e = hipEventSynchronize(p->ev_done[slot]);
        t_dev_wait += now_s() - tw;
        if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "waiting for the results of pileup %u failed: %s", f, hipGetErrorString(e));
        const char *r = (const char *)p->result[slot];
        uint64_t status;
        uint32_t found;
        memcpy(&status, r, 8);
        memcpy(&found, r + 8, 4);
        memcpy(&out_status[2 * f + 1], r + 24, 8);              // the file's line count
        out_status[2 * f] = status;
        out_n_sites[f] = found;
        if (status != ~0ull) { if (src[f].rc == SNPGPU_OK) src[f].rc = SNPGPU_E_PILEUP; out_rc[f] = src[f].rc; publish(f); return SNPGPU_OK; }
        const uint32_t got = found < capacity ? found : capacity;
        snpgpu_varscan_site *dst = out_sites + (size_t)f * capacity;
        if (got) memcpy(dst, r + r_rec, sizeof(snpgpu_varscan_site) * (size_t)got);
        if (found <= capacity)
            std::sort(dst, dst + got, [](const snpgpu_varscan_site &x, const snpgpu_varscan_site &y) {
                return x.line_off != y.line_off ? x.line_off < y.line_off : x.alt_base < y.alt_base;
            });
        out_rc[f] = src[f].rc;
        publish(f);
        return SNPGPU_OK;
    };
    // The work on a complete file: one pass over its text, the walk over the candidates, results into the slot's pinned block.
    // Nothing here waits for the device (round 3 needed the file's line count on the host between two halves of this).
#define VS_RET(expr)                                                                                                \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
    auto post = [&](uint32_t c) -> int {
        const uint32_t f = file_of[c], slot = c % n_slots;
        const uint64_t nbytes = src[f].size;
        const uint8_t *d_file = dest(f);
        char *res = (char *)p->result[slot];
        size_t o = 0;
        const size_t o_ctl = o; o += 256;
        const size_t o_rec = o; o += up(sizeof(snpgpu_varscan_site) * (size_t)capacity, 256) + 256;
        const size_t o_var = o; o += up(snpgpu_varscan_scratch_bytes(nbytes), 256);
        void *scr = nullptr;
        int r = snpgpu_scratch(ctx, 2 * o + 512, &scr);         // may move the scratch (it waits for the compute stream first)
        if (r) return r;
        const size_t half = ctx->scratch_bytes / 2 / 256 * 256;
        char *b = (char *)scr + half * slot;
        uint64_t h_ctl[5] = {~0ull, 0, 0, 0, 0};                // status; records found + candidates; long candidates + spare; lines; spare
        memcpy(res + 64, h_ctl, sizeof h_ctl);                  // (a pinned source that stays valid until the copy has run)
        VS_RET(hipMemcpyAsync(b + o_ctl, res + 64, sizeof h_ctl, hipMemcpyHostToDevice, st));
        r = snpgpu_enqueue_varscan(ctx, d_file, nbytes, params, (snpgpu_varscan_site *)(b + o_rec), capacity, (uint32_t *)(b + o_ctl + 8), (uint64_t *)(b + o_ctl),
                                   b + o_var);
        if (r) return r;
        VS_RET(hipMemcpyAsync(res, b + o_ctl, 32, hipMemcpyDeviceToHost, st));
        if (capacity) VS_RET(hipMemcpyAsync(res + r_rec, b + o_rec, sizeof(snpgpu_varscan_site) * (size_t)capacity, hipMemcpyDeviceToHost, st));
        VS_RET(hipEventRecord(p->ev_done[slot], st));
        has_result[c] = 1;
        return SNPGPU_OK;
    };
#undef VS_RET
    uint32_t harvested = 0;                                     // sequence numbers [0, harvested) have been copied out
    // file f is complete (every piece issued): give it its sequence number (unless it has one) and start its post-processing
    auto file_complete = [&](uint32_t f) -> int {
        if (seq_of[f] < 0) { seq_of[f] = (int64_t)file_of.size(); file_of.push_back(f); }
        const uint32_t c = (uint32_t)seq_of[f];
        completed[c] = 1;
        if (!params || src[f].rc != SNPGPU_OK || src[f].size == 0) return SNPGPU_OK;
        int r = SNPGPU_OK;
        while (!r && harvested + n_slots <= c) r = harvest(harvested++);           // its result block and scratch half are free then
        if (!r) r = post(c);
        return r;
    };
    // what can be done without waiting: hand out results that are ready
    auto opportunistic = [&]() -> int {
        int r = SNPGPU_OK;
        while (!r && harvested < file_of.size() && completed[harvested] &&
               (!has_result[harvested] || hipEventQuery(p->ev_done[harvested % n_slots]) != hipErrorNotReady))
            r = harvest(harvested++);
        return r;
    };
#define VS_TRY(expr)                                                                                                \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) { rc = snpgpu_set_error(ctx, SNPGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)
    {
        // ---- pass A: the resident prefix, pieces copied in the order in which their reads finish -----------------------------
        if (JA) {
            AnyOrder sh;
            sh.n_jobs = JA;
            sh.opener = &opener;
            for (uint32_t i = 0; i < R; ++i) sh.free_bufs.push_back(i);
            std::vector<std::thread> readers;
            const uint32_t nr = n_readers < JA ? n_readers : (uint32_t)JA;
            try {
                for (uint32_t i = 0; i < nr; ++i) readers.emplace_back(reader_any_order, ctx, &sh, &jobs, &src);
            } catch (const std::exception &e) {
                opener.cancel();
                { std::lock_guard<std::mutex> lk(sh.mu); sh.abort = true; sh.next.store(JA); }
                sh.cv_free.notify_all();
                for (auto &t : readers) t.join();
                rc = snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "cannot start the reader threads: %s", e.what());
                goto done;
            }
            // The blocks are allocated by a helper thread, in the order in which they will be needed, while this thread copies
            // into the ones that exist: an allocation from the driver costs ~15 ms per GiB (and many times that for memory that
            // was freed a moment ago), about as long as the copy into it, but it does not hold up copies issued by another thread
            // (tools/probe/alloc_dirty_probe.cpp).
            std::mutex alloc_mu;
            std::condition_variable alloc_cv;
            std::vector<hipError_t> alloc_err(block_bytes.size(), hipSuccess);
            std::vector<uint8_t> alloc_done(block_bytes.size(), 0);
            std::atomic<bool> alloc_stop{false};
            double alloc_thread_seconds = 0;
            auto allocate_all = [&] {
                (void)hipSetDevice(ctx->device);
                for (size_t b = 0; b < block_bytes.size() && !alloc_stop.load(); ++b) {
                    void *d = nullptr;
                    const double ta = now_s();
                    const hipError_t e = hipMalloc(&d, block_bytes[b]);
                    alloc_thread_seconds += now_s() - ta;
                    {
                        std::lock_guard<std::mutex> lk(alloc_mu);
                        block_ptr[b] = e == hipSuccess ? (uint8_t *)d : nullptr;
                        alloc_err[b] = e;
                        alloc_done[b] = 1;
                    }
                    alloc_cv.notify_all();
                    if (e != hipSuccess) break;
                }
            };
            std::thread allocator;
            try { allocator = std::thread(allocate_all); } catch (const std::exception &) { allocate_all(); }      // no thread to be had: up front, here
            std::deque<uint32_t> inflight;                      // staging buffers whose copies are on their way, in issue order per stream pair
            uint64_t issued = 0;
            int rcA = SNPGPU_OK;
            while (issued < JA && !rcA) {
                // staging buffers whose copy has finished go back to the readers
                bool freed = false;
                while (!inflight.empty() && hipEventQuery(p->ev_copy[inflight.front()]) != hipErrorNotReady) {
                    std::lock_guard<std::mutex> lk(sh.mu);
                    sh.free_bufs.push_back(inflight.front());
                    inflight.pop_front();
                    freed = true;
                }
                if (freed) sh.cv_free.notify_all();
                rcA = opportunistic();
                if (rcA) break;
                AnyOrder::Ready rd{0, -1, 0};
                bool have = false;
                {
                    const double tr = now_s();
                    std::unique_lock<std::mutex> lk(sh.mu);
                    if (sh.ready.empty())
                        sh.cv_ready.wait_for(lk, std::chrono::microseconds(inflight.empty() ? 2000 : 20), [&] { return !sh.ready.empty(); });
                    if (!sh.ready.empty()) { rd = sh.ready.front(); sh.ready.pop_front(); have = true; }
                    lk.unlock();
                    t_read_wait += now_s() - tr;
                }
                if (!have) continue;
                const Job &jb = jobs[rd.job];
                const uint32_t f = jb.file;
                Source &s = src[f];
                if (rd.err && s.rc == SNPGPU_OK) s.rc = SNPGPU_E_IO;
                if (place[f].block != ~0u && !block_seen[place[f].block]) {
                    const uint32_t b = place[f].block;
                    const double ta = now_s();
                    hipError_t e;
                    {
                        std::unique_lock<std::mutex> lk(alloc_mu);
                        alloc_cv.wait(lk, [&] { return alloc_done[b] != 0 || (b > 0 && alloc_done[b - 1] && alloc_err[b - 1] != hipSuccess); });
                        e = alloc_done[b] ? alloc_err[b] : hipErrorOutOfMemory;
                    }
                    t_alloc += now_s() - ta;
                    if (e != hipSuccess) {
                        rcA = snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipMalloc(%llu) for resident pileups failed: %s", (unsigned long long)block_bytes[b], hipGetErrorString(e));
                        break;
                    }
                    block_seen[b] = 1;
                }
                if (place[f].block != ~0u) store->files[first + f].d = dest(f);
                if (jb.len) {
                    hipStream_t cs = (issued & 1) ? p->copy_stream2 : p->copy_stream;
                    hipError_t e = hipMemcpyAsync(dest(f) + jb.off, p->staging[rd.buf], jb.len, hipMemcpyHostToDevice, cs);
                    if (e == hipSuccess) e = hipEventRecord(p->ev_copy[rd.buf], cs);
                    if (e == hipSuccess) e = hipStreamWaitEvent(st, p->ev_copy[rd.buf], 0);
                    if (e != hipSuccess) { rcA = snpgpu_set_error(ctx, SNPGPU_E_HIP, "copying a piece of %s failed: %s", s.path, hipGetErrorString(e)); break; }
                    store->h2d_bytes += jb.len;
                    // (the two copy streams finish their copies in their own order: a buffer is retired when the one at the head of
                    // the queue is — at most one copy later than it could be)
                    inflight.push_back((uint32_t)rd.buf);
                }
                ++issued;
                if (--chunks_left[f] == 0) rcA = file_complete(f);
            }
            if (rcA) opener.cancel();
            {
                std::lock_guard<std::mutex> lk(sh.mu);
                if (rcA) { sh.abort = true; sh.next.store(JA); }
            }
            sh.cv_free.notify_all();
            for (auto &t : readers) t.join();
            if (rcA) alloc_stop.store(true);
            if (allocator.joinable()) allocator.join();
#ifdef SNPGPU_TUNING
            if (getenv("SNPGPU_VARSCAN_DEBUG")) fprintf(stderr, "ingest: the allocator thread spent %.3f s in hipMalloc (%zu blocks)\n", alloc_thread_seconds, block_bytes.size());
#endif
            for (size_t b = 0; b < block_ptr.size(); ++b) if (block_ptr[b]) store->blocks.push_back(block_ptr[b]);      // the store owns them, used or not
            ns_reading += sh.ns_reading.load();
            ns_waiting += sh.ns_waiting.load();
            if (rcA) { rc = rcA; goto done; }
            // every staging buffer is free again before the in-order ring of pass B uses them
            if (JA < J) { VS_TRY(hipStreamSynchronize(p->copy_stream)); VS_TRY(hipStreamSynchronize(p->copy_stream2)); }
        }
        // ---- pass B: files that go through the device slots, pieces in file order -------------------------------------------
        if (JA < J) {
            Shared sh;
            sh.R = R ? R : 1;
            sh.filled.assign(J, 0);
            sh.job_err.assign(J, 0);
            sh.next.store(JA);
            sh.base = JA;
            sh.opener = &opener;
            std::vector<std::thread> readers;
            try {
                for (uint32_t i = 0; i < n_readers; ++i) readers.emplace_back(reader_main, ctx, &sh, &jobs, &src);
            } catch (const std::exception &e) {
                opener.cancel();
                { std::lock_guard<std::mutex> lk(sh.mu); sh.abort = true; sh.next.store(J); }
                sh.cv.notify_all();
                for (auto &t : readers) t.join();
                rc = snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "cannot start the reader threads: %s", e.what());
                goto done;
            }
            int rcB = SNPGPU_OK;
            int64_t copies_done = (int64_t)JA;
            for (uint64_t j = JA; j < J && !rcB; ++j) {
                const Job &jb = jobs[j];
                const uint32_t f = jb.file;
                Source &s = src[f];
                rcB = opportunistic();
                if (rcB) break;
                if (jb.first) {                                 // a sequence number now: its slot is where the pieces go
                    seq_of[f] = (int64_t)file_of.size();
                    file_of.push_back(f);
                    const uint32_t c = (uint32_t)seq_of[f];
                    while (!rcB && harvested + n_slots <= c) rcB = harvest(harvested++);
                    if (rcB) break;
                }
                uint8_t *d_file = dest(f);
                const double tr = now_s();
                for (;;) {                                      // wait for the piece to be read; retire finished copies meanwhile
                    bool progress = false;
                    while (copies_done < (int64_t)j && hipEventQuery(p->ev_copy[(copies_done - JA) % R]) != hipErrorNotReady) { ++copies_done; progress = true; }
                    std::unique_lock<std::mutex> lk(sh.mu);
                    if (progress) { sh.freed = copies_done - (int64_t)JA; lk.unlock(); sh.cv.notify_all(); lk.lock(); }
                    if (sh.filled[j]) break;
                    sh.cv.wait_for(lk, std::chrono::microseconds(copies_done < (int64_t)j ? 20 : 2000), [&] { return sh.filled[j] != 0; });
                    if (sh.filled[j]) break;
                }
                t_read_wait += now_s() - tr;
                if (sh.job_err[j] && s.rc == SNPGPU_OK) s.rc = SNPGPU_E_IO;
                hipStream_t cs = (j & 1) ? p->copy_stream2 : p->copy_stream;
                hipError_t e = hipSuccess;
                if (jb.len) e = hipMemcpyAsync(d_file + jb.off, p->staging[(j - JA) % R], jb.len, hipMemcpyHostToDevice, cs);
                if (store) store->h2d_bytes += jb.len;
                if (e == hipSuccess) e = hipEventRecord(p->ev_copy[(j - JA) % R], cs);
                if (e == hipSuccess) e = hipStreamWaitEvent(st, p->ev_copy[(j - JA) % R], 0);
                if (e != hipSuccess) { rcB = snpgpu_set_error(ctx, SNPGPU_E_HIP, "copying a piece of %s failed: %s", s.path, hipGetErrorString(e)); break; }
                if (jb.last) rcB = file_complete(f);
            }
            if (rcB) opener.cancel();
            {
                std::lock_guard<std::mutex> lk(sh.mu);
                if (rcB) { sh.abort = true; sh.next.store(J); }
            }
            sh.cv.notify_all();
            for (auto &t : readers) t.join();
            ns_reading += sh.ns_reading.load();
            ns_waiting += sh.ns_waiting.load();
            if (rcB) { rc = rcB; goto done; }
        }
        while (harvested < file_of.size()) { rc = harvest(harvested++); if (rc) goto done; }
    }
done:
#undef VS_TRY
    {   // every copy has landed before the caller looks at resident memory (or reuses a slot)
        hipError_t e1 = hipStreamSynchronize(p->copy_stream), e2 = hipStreamSynchronize(p->copy_stream2);
        if (rc) (void)hipStreamSynchronize(st);
        if (!rc && (e1 != hipSuccess || e2 != hipSuccess)) rc = snpgpu_set_error(ctx, SNPGPU_E_HIP, "pileup copies failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    close_all();
    if (store) {
        store->seconds += now_s() - t_enter;
        store->seconds_preparing += t_begin - t_enter;
        store->seconds_allocating += t_alloc;
        store->seconds_waiting_for_readers += t_read_wait;
        store->seconds_waiting_for_device += t_dev_wait;
        store->reader_seconds_reading += ns_reading * 1e-9;
        store->reader_seconds_waiting += ns_waiting * 1e-9;
    }
    for (uint32_t f = 0; f < n_files; ++f) {
        out_rc[f] = src[f].rc;
        if (store && (src[f].rc == SNPGPU_E_IO || (place[f].block != ~0u && !block_ptr[place[f].block])))
            store->files[first + f].resident = false;                                       // its bytes are void / never arrived
        if (rc && out_done) __atomic_store_n(&out_done[f], 1, __ATOMIC_RELEASE);            // nobody waits for a call that failed
    }
    return rc;
}

}  // namespace

extern "C" {

int snpgpu_varscan_file(snpgpu_ctx *ctx, const char *path, const snpgpu_varscan_params *params, uint32_t capacity,
                        snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status) {
    if (!ctx || !path || !params || !out_n_sites || !out_status) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (capacity && !out_sites) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    uint8_t *d_file = nullptr;
    uint64_t nbytes = 0;
    int rc = load_file(ctx, path, &d_file, &nbytes);
    if (rc) return rc;
    return varscan_resident(ctx, d_file, nbytes, path, params, capacity, out_sites, out_n_sites, out_status);
}

// The same over a pileup that is already in device memory (a resident file of snpgpu_pileups, a synthetic one).
int snpgpu_varscan_dev(snpgpu_ctx *ctx, const void *d_pileup, uint64_t nbytes, const snpgpu_varscan_params *params, uint32_t capacity,
                       snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status) {
    if (!ctx || !params || !out_n_sites || !out_status || (nbytes && !d_pileup)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (capacity && !out_sites) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    return varscan_resident(ctx, (const uint8_t *)d_pileup, nbytes, "the resident pileup", params, capacity, out_sites, out_n_sites, out_status);
}

// Many resident pileups, one launch.
int snpgpu_varscan_batch_dev(snpgpu_ctx *ctx, const void *const *d_pileups, const uint64_t *nbytes, uint32_t n_files, const snpgpu_varscan_params *params,
                             uint32_t capacity, snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status, int32_t *out_rc) {
    if (!ctx || !params || !out_n_sites || !out_status || !out_rc || (n_files && (!d_pileups || !nbytes)) || (capacity && !out_sites))
        return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (!n_files) return SNPGPU_OK;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    return varscan_resident_batch(ctx, d_pileups, nbytes, n_files, params, capacity, out_sites, out_n_sites, out_status, out_rc);
}

int snpgpu_varscan_files(snpgpu_ctx *ctx, const char *const *paths, uint32_t n_files, const snpgpu_varscan_params *params, uint32_t capacity,
                         snpgpu_varscan_site *out_sites, uint32_t *out_n_sites, uint64_t *out_status, int32_t *out_rc) {
    if (!ctx || !params || !out_n_sites || !out_status || !out_rc || (n_files && !paths) || (capacity && !out_sites))
        return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (!n_files) return SNPGPU_OK;
    return varscan_stream(ctx, paths, n_files, params, capacity, out_sites, out_n_sites, out_status, out_rc, nullptr, nullptr);
}

// ---- resident pileups: the input side of the one-job pipeline -----------------------------------------------------------
int snpgpu_pileups_create(snpgpu_ctx *ctx, uint64_t budget_bytes, snpgpu_pileups **out) {
    if (!ctx || !out) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    *out = nullptr;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    if (!budget_bytes) {                                        // what is free now, less room for the outputs of the later steps
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        const uint64_t keep = (uint64_t)24 << 30;
        budget_bytes = free_b > keep ? (uint64_t)free_b - keep : 0;
    }
    snpgpu_pileups *s = new snpgpu_pileups();
    s->ctx = ctx;
    s->budget = budget_bytes;
    *out = s;
    return SNPGPU_OK;
}

void snpgpu_pileups_destroy(snpgpu_pileups *s) {
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    for (void *d : s->blocks) if (d) (void)hipFree(d);
    delete s;
}

int snpgpu_pileups_ingest(snpgpu_ctx *ctx, snpgpu_pileups *store, const char *const *paths, uint32_t n_files,
                          const snpgpu_varscan_params *params, uint32_t capacity, snpgpu_varscan_site *out_sites, uint32_t *out_n_sites,
                          uint64_t *out_status, int32_t *out_rc, int32_t *out_done) {
    if (!ctx || !store || store->ctx != ctx || !out_n_sites || !out_status || !out_rc || (n_files && !paths) || (params && capacity && !out_sites))
        return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (!n_files) return SNPGPU_OK;
    return varscan_stream(ctx, paths, n_files, params, params ? capacity : 0, out_sites, out_n_sites, out_status, out_rc, store, out_done);
}

uint32_t snpgpu_pileups_count(const snpgpu_pileups *store) { return store ? (uint32_t)store->files.size() : 0; }

int snpgpu_pileups_get(const snpgpu_pileups *store, uint32_t index, void **d_ptr, uint64_t *nbytes) {
    if (!store || index >= store->files.size()) return SNPGPU_E_ARG;
    const snpgpu_pileups::Entry &e = store->files[index];
    if (d_ptr) *d_ptr = e.resident ? (void *)e.d : nullptr;
    if (nbytes) *nbytes = e.nbytes;
    return SNPGPU_OK;
}

int snpgpu_pileups_get_stats(const snpgpu_pileups *store, snpgpu_pileups_stats *out) {
    if (!store || !out) return SNPGPU_E_ARG;
    memset(out, 0, sizeof *out);
    out->h2d_bytes = store->h2d_bytes;
    out->file_bytes = store->file_bytes;
    out->resident_bytes = store->used;
    out->budget_bytes = store->budget;
    out->n_files = (uint32_t)store->files.size();
    for (const auto &e : store->files) if (e.resident) ++out->n_resident;
    out->seconds = store->seconds;
    out->seconds_allocating = store->seconds_allocating;
    out->seconds_waiting_for_readers = store->seconds_waiting_for_readers;
    out->seconds_waiting_for_device = store->seconds_waiting_for_device;
    out->reader_seconds_reading = store->reader_seconds_reading;
    out->reader_seconds_waiting = store->reader_seconds_waiting;
    out->seconds_preparing = store->seconds_preparing;
    out->n_readers = store->n_readers;
    return SNPGPU_OK;
}

}  // extern "C"
