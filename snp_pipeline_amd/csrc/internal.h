// Internal declarations shared by the HIP translation units of libsnpgpu.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/snpgpu.h"

struct snpgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream work is enqueued on
    hipStream_t own_stream = nullptr;   // created by the context
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::string err;
    // scratch reused across calls (grown on demand)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int n_cu = 256;
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
// RAII-less helpers: call begin before the launch and end right after it (no-ops unless timing is enabled)
hipEvent_t snpgpu_time_begin(snpgpu_ctx *ctx);
void snpgpu_time_end(snpgpu_ctx *ctx, int kernel, hipEvent_t a);

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
};
size_t snpgpu_scan_workspace_bytes(const snpgpu_ctx *ctx, uint32_t n_samples);
int snpgpu_enqueue_scan(snpgpu_ctx *ctx, const snpgpu_siteset *ss, std::vector<SampleDev> &h_samples, void *workspace,
                        uint64_t *d_site_line, int want_depth, uint32_t *d_zero32, uint32_t n_zero32);
#define SNPGPU_SCAN_MAX_BATCH 256   // samples per scan launch (each gets at least ~16 of the 4096 waves)

int snpgpu_set_error(snpgpu_ctx *ctx, int code, const char *fmt, ...);
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
__device__ __forceinline__ uint32_t to_upper(uint32_t c) { return (c - 97u <= 25u) ? c - 32u : c; }
__device__ __forceinline__ uint32_t to_lower(uint32_t c) { return (c - 65u <= 25u) ? c + 32u : c; }
