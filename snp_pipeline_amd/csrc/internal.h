// Internal declarations shared by the HIP translation units of libsnpgpu.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/snpgpu.h"

struct snpgpu_stream_pool;          // stream.hip: pinned staging ring, device file slots, copy stream
void snpgpu_stream_pool_destroy(struct snpgpu_ctx *ctx);
void snpgpu_comm_release(struct snpgpu_ctx *ctx);       // comm.hip: the communicator goes with the context

struct snpgpu_ctx {
    int device = 0;
    snpgpu_stream_pool *pool = nullptr;
    hipStream_t stream = nullptr;       // stream work is enqueued on
    hipStream_t own_stream = nullptr;   // created by the context
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::string err;
    // scratch reused across calls (grown on demand)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int n_cu = 256;
    bool scan_lds_attr = false;         // hipFuncSetAttribute is per device: done once per context
    bool varscan_lds_attr = false;
    void *comm = nullptr;               // comm.hip: the RCCL communicator of this rank (snpgpu_comm_init), or none
    // positions with more than SNPGPU_MAX_SYMS symbols: [SNPGPU_SPILL_CAP] records + one counter word (allocated on first use)
    snpgpu_symbol_spill *d_spill = nullptr;
    uint32_t *d_spill_n = nullptr;
    uint32_t spill_cap = 0, spill_want = SNPGPU_SPILL_CAP;     // records the arena holds / should hold at the next call (it grows when a call ran out)
    // optional per-kernel timing (bench): event pairs recorded around selected launches
    bool time_kernels = false;
    struct Timed { int kernel; hipEvent_t a, b; };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> event_pool;
};

#define SNPGPU_SLOW_QUEUE_CAP (1u << 20)
#define SNPGPU_K_SCAN 0
#define SNPGPU_K_CALL 1
#define SNPGPU_K_DISTANCE 2
#define SNPGPU_K_VARSCAN 3          // everything phase-1 site calling launches for one file (scan, walk, long walk)
// RAII-less helpers: call begin before the launch and end right after it (no-ops unless timing is enabled)
// Start of a public call that produces per-site records: the spill is there and empty (enqueued on the context's stream).
int snpgpu_spill_begin(snpgpu_ctx *ctx);
hipEvent_t snpgpu_time_begin(snpgpu_ctx *ctx);
void snpgpu_time_end(snpgpu_ctx *ctx, int kernel, hipEvent_t a);

// Pileup files kept in device memory between the steps of the one-job pipeline (stream.hip): bump-allocated out of a few
// large blocks, so that a file is copied over the host link once and then read by site calling and by the consensus scan.
struct snpgpu_pileups {
    snpgpu_ctx *ctx = nullptr;
    uint64_t budget = 0;                // device bytes the resident files may take
    uint64_t used = 0;
    std::vector<void *> blocks;         // hipMalloc'd
    struct Entry { uint8_t *d = nullptr; uint64_t nbytes = 0; bool resident = false; };
    std::vector<Entry> files;           // in the order they were ingested
    uint64_t h2d_bytes = 0;             // every byte copied host -> device through this store
    uint64_t file_bytes = 0;            // sizes of the files that were ingested
    double seconds = 0, seconds_allocating = 0, seconds_waiting_for_readers = 0, seconds_waiting_for_device = 0;   // summed over the ingest calls
    double reader_seconds_reading = 0, reader_seconds_waiting = 0, seconds_preparing = 0;
    uint32_t n_readers = 0;             // reader threads of the last ingest call
};

// Device-side view of a site set.
struct SiteSetDev {
    const uint8_t *names;       // concatenated contig names (sorted bytewise)
    const uint32_t *name_off;   // n_contigs + 1
    const uint64_t *bit_off;    // n_contigs: first bit of the contig's slice of the bitmap
    const uint32_t *max_pos;    // n_contigs: highest listed position
    const uint32_t *bitmap;     // one bit per (contig, pos <= max_pos)
    const uint32_t *rank;       // per bitmap word: number of set bits in earlier words == index into keys
    const uint8_t *flags;       // n_sites
    uint64_t n_words;           // dwords in bitmap / rank
    uint32_t n_contigs;
    uint32_t n_sites;
};

struct snpgpu_siteset {
    snpgpu_ctx *ctx = nullptr;
    SiteSetDev dev{};
    void *blob = nullptr;       // one allocation backing every device array
    uint64_t total_bits = 0;
    std::vector<uint8_t> h_flags;    // host copy of the flags (per-file exclude lists are OR-ed into copies of it)
    uint64_t *site_line = nullptr;   // n_sites scratch: (offset+1) of the last matching line
    uint64_t *slow_queue = nullptr;  // SNPGPU_SLOW_QUEUE_CAP file offsets of lines the fast scan path left over
    uint32_t *slow_ctl = nullptr;    // [0] queue length, [1] overflow flag
    uint32_t n_sites = 0;
};

// One pileup of a batch, as the scan and call kernels see it.
struct SampleDev {
    const uint8_t *buf;         // first byte of the file (device memory, any alignment)
    uint64_t nbytes;
    uint64_t *status;           // SNPGPU_SCAN_STATUS_WORDS
    uint32_t wave0, n_waves;    // the scan launch's waves [wave0, wave0 + n_waves) work on this sample
    uint32_t tile_lo, tile_hi;  // ... on its 4 KiB tiles [tile_lo, tile_hi): the whole file, or — while the file is still
                                // streaming in from the host — the tiles whose bytes (and halo) have landed
};
#define SNPGPU_SCAN_TILE 4096
#define SNPGPU_SCAN_HALO 128
static inline uint64_t snpgpu_scan_tiles(const void *buf, uint64_t nbytes) {
    return (((uintptr_t)buf & 15) + nbytes + SNPGPU_SCAN_TILE - 1) / SNPGPU_SCAN_TILE;
}
// The scan of a batch in pieces (scan.hip).  A batch is described by a device table of n SampleDev entries followed by a
// sentinel whose wave0 is the number of waves of the launch; snpgpu_scan_deal fills wave0 / n_waves of a host table
// (tile ranges already set) and returns that number.
//   snpgpu_scan_begin   zero the line-offset rows, status words, queue control words (and n_zero32 caller words)
//   snpgpu_scan_range   the fast pass over the tile ranges of `d_table`; counts are ADDED to the status words, so a file
//                       may be scanned in several launches while it arrives (d_table entries then describe the part of
//                       the file that has landed: nbytes = landed bytes, tile_hi * 4096 + 128 <= landed)
//   snpgpu_scan_end     the exact parser over the queued lines, the exact pass when the queue overflowed; d_table must
//                       describe the complete files
size_t snpgpu_scan_totals_bytes(const snpgpu_ctx *ctx);
uint32_t snpgpu_scan_deal(const snpgpu_ctx *ctx, SampleDev *h_table, uint32_t n, uint32_t min_tiles_per_wave);
int snpgpu_scan_begin(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n, uint64_t *d_site_line,
                      uint32_t *d_zero32, uint32_t n_zero32);
int snpgpu_scan_range(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n, uint32_t n_waves,
                      uint64_t *d_totals, uint64_t *d_site_line, int want_depth);
int snpgpu_scan_end(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n, uint32_t n_waves,
                    uint64_t *d_totals, uint64_t *d_site_line, int want_depth);
// all of the above for files that are resident: `workspace` holds snpgpu_scan_workspace_bytes()
size_t snpgpu_scan_workspace_bytes(const snpgpu_ctx *ctx, uint32_t n_samples);
int snpgpu_enqueue_scan(snpgpu_ctx *ctx, const snpgpu_siteset *ss, std::vector<SampleDev> &h_samples, void *workspace,
                        uint64_t *d_site_line, int want_depth, uint32_t *d_zero32, uint32_t n_zero32);
// every line of a pileup (--vcfAllPos): line index + per-line site flags (scan.hip)
size_t snpgpu_lines_workspace_words(uint64_t nbytes);
int snpgpu_enqueue_lines_count(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, uint32_t *ws, uint32_t **d_total);
int snpgpu_enqueue_lines_emit(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const uint8_t *d_buf, uint64_t nbytes, uint32_t *ws,
                              uint64_t *d_line_off, uint8_t *d_flags, uint64_t n_lines, uint64_t *d_status);
// the line offsets alone (after snpgpu_enqueue_lines_count with the same ws)
int snpgpu_enqueue_lines_offsets(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, uint32_t *ws, uint64_t *d_line_off, uint64_t n_lines);
// phase-1 site calling straight over the text, no line index (varscan.hip): see snpgpu_enqueue_varscan there for the arguments
size_t snpgpu_varscan_scratch_bytes(uint64_t nbytes);
int snpgpu_enqueue_varscan(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, const snpgpu_varscan_params *prm, snpgpu_varscan_site *d_sites,
                           uint32_t capacity, uint32_t *d_ctl, uint64_t *d_status, void *d_scratch);
// ... many resident pileups in one launch (see varscan.hip); h_table: snpgpu_varscan_table_bytes(n_files) bytes of host memory that stay valid
// until the stream has passed the call
size_t snpgpu_varscan_batch_scratch_bytes(uint64_t total_bytes, uint32_t n_files);
size_t snpgpu_varscan_table_bytes(uint32_t n_files);
int snpgpu_enqueue_varscan_batch(snpgpu_ctx *ctx, const uint8_t *const *d_bufs, const uint64_t *nbytes, uint32_t n_files, const snpgpu_varscan_params *prm,
                                 snpgpu_varscan_site *d_sites, uint32_t capacity, uint32_t *d_ctl, uint64_t *d_status, void *d_scratch, void *h_table);
// ... and the wave-per-site call kernel over such a list (consensus.hip): "site" i is line i
int snpgpu_enqueue_call_lines(snpgpu_ctx *ctx, const SampleDev *d_sample, const uint64_t *d_line_off, const uint8_t *d_flags,
                              uint32_t n_lines, const snpgpu_caller_params *prm, uint8_t *d_out_base, uint8_t *d_out_filters,
                              snpgpu_site_counts *d_out_counts, uint32_t *d_todo_n = nullptr, uint64_t *d_todo = nullptr, uint64_t *d_todo2 = nullptr);
// the call kernels over a scanned batch (consensus.hip); d_todo_n: 4 words (3 zeroed), d_todo / d_todo2: n * n_sites entries each;
// lines_out.hip: the per-line records of --vcfAllPos packed into 24 bytes where they fit, the others gathered as they are
size_t snpgpu_compact_lines_workspace_words(uint64_t n_lines);
int snpgpu_enqueue_compact_lines(snpgpu_ctx *ctx, const snpgpu_site_counts *d_counts, const uint8_t *d_flags, uint64_t n_lines, snpgpu_line_record *d_recs,
                                 uint32_t *ws, uint32_t **d_n_wide);
int snpgpu_enqueue_gather_wide(snpgpu_ctx *ctx, const snpgpu_site_counts *d_counts, const snpgpu_line_record *d_recs, uint64_t n_lines, const uint32_t *ws,
                               uint32_t capacity, uint32_t *d_wide_index, snpgpu_site_counts *d_wide);
void snpgpu_expand_line_record(const snpgpu_line_record &r, snpgpu_site_counts *out);
// vcf_rows.hip: consensus.vcf rows of the lines [lo, hi) from those records (host)
bool snpgpu_line_rows_into(const uint8_t *text, uint64_t nbytes, const uint64_t *line_off, const snpgpu_line_record *recs, uint64_t first, uint64_t lo, uint64_t hi,
                             const uint32_t *wide_index, const snpgpu_site_counts *wide, uint32_t n_wide, const char *const *filter_names,
                             int preserve_ref_case, char failed_snp_gt, const snpgpu_symbol_spill *spill, uint32_t n_spill, int only_listed,
                             std::vector<char> &out, uint64_t *n_rows, uint64_t *bad_line);

// d_site_flags: nullptr = the site set's flags for every sample, else flags of sample i at d_site_flags + i * flags_stride
int snpgpu_enqueue_call(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n,
                        const snpgpu_caller_params *prm, const uint64_t *d_site_line, uint8_t *d_out_base,
                        uint8_t *d_out_filters, snpgpu_site_counts *d_out_counts, uint32_t *d_todo_n, uint64_t *d_todo,
                        uint64_t *d_todo2, const uint8_t *d_site_flags = nullptr, uint32_t flags_stride = 0);
#define SCAN_ERR_FEW_FIELDS 1
#define SCAN_ERR_BAD_POS 2
#define SCAN_ERR_NON_ASCII 3
#define SNPGPU_SCAN_MAX_BATCH 256   // samples per scan launch (each gets at least ~16 of the 4096 waves)

int snpgpu_set_error(snpgpu_ctx *ctx, int code, const char *fmt, ...);
// host_budget.hip: the host threads this process may start (affinity mask, cgroup quota, MaxCpuCores, ranks sharing the node)
void snpgpu_host_budget(snpgpu_cpu_budget_info *out);
uint32_t snpgpu_reader_threads();                       // file readers of the streamed entry points
uint32_t snpgpu_writer_threads(uint32_t at_most);       // formatting threads (0 = no other bound)
uint32_t snpgpu_cpu_threads(uint32_t at_most);          // the whole budget, for pools of one job per thread
// Start of an entry point: make the context's device current and drop whatever error an EARLIER runtime call of this thread left
// behind — other users of the HIP runtime in the process (torch probes devices and pointers) leave sticky errors that the launch
// checks (hipGetLastError after a kernel launch) would otherwise report as ours.
static inline hipError_t snpgpu_enter(snpgpu_ctx *ctx) {
    const hipError_t e = hipSetDevice(ctx->device);
    (void)hipGetLastError();
    return e;
}
int snpgpu_scratch(snpgpu_ctx *ctx, size_t bytes, void **out);

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return snpgpu_set_error((ctx), SNPGPU_E_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                    hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

// ---- byte classes -----------------------------------------------------------------------------
// What str.split() treats as a separator, restricted to ASCII (pileup.py:206/424).
__device__ __forceinline__ bool is_ws(uint32_t c) { return (c - 9u <= 4u) || (c - 28u <= 4u); }
// Universal-newline terminators of CPython's text-mode iterator.
__device__ __forceinline__ bool is_term(uint32_t c) { return c == 10u || c == 13u; }
__device__ __forceinline__ bool is_digit(uint32_t c) { return c - 48u <= 9u; }
// Python's int() on an ASCII field, one byte at a time: [+-]? digit ( _? digit )*.  Feed the bytes in order; `ok()` at the end.
struct PyInt {
    uint64_t v = 0;             // magnitude, saturating at 2^62
    uint32_t n = 0;             // bytes seen
    bool neg = false, bad = false, last_digit = false;
    __device__ __forceinline__ void feed(uint32_t c) {
        if (n == 0 && (c == '+' || c == '-')) neg = c == '-';
        else if (is_digit(c)) {                                  // (saturate BEFORE the multiply: 2^62 * 10 wraps in 64 bits)
            const uint64_t dgt = c - 48u;
            v = v > ((1ull << 62) - dgt) / 10u ? 1ull << 62 : v * 10 + dgt;
            last_digit = true;
        }
        else if (c == '_' && last_digit) last_digit = false;     // an underscore sits between two digits
        else bad = true;
        ++n;
    }
    __device__ __forceinline__ bool ok() const { return !bad && last_digit; }
};
__device__ __forceinline__ uint32_t to_upper(uint32_t c) { return (c - 97u <= 25u) ? c - 32u : c; }
__device__ __forceinline__ uint32_t to_lower(uint32_t c) { return (c - 65u <= 25u) ? c + 32u : c; }
