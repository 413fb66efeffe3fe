// Context, error reporting, scratch memory and the site set (bitmap + rank directory).
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "internal.h"
#include "prims.h"

int snpgpu_set_error(snpgpu_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int snpgpu_scratch(snpgpu_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes + (bytes >> 2) + 4096;
        hipError_t e = hipMalloc(&ctx->scratch, want);
        if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return SNPGPU_OK;
}

static hipEvent_t take_event(snpgpu_ctx *ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

hipEvent_t snpgpu_time_begin(snpgpu_ctx *ctx) {
    if (!ctx->time_kernels) return nullptr;
    hipEvent_t a = take_event(ctx);
    (void)hipEventRecord(a, ctx->stream);
    return a;
}

void snpgpu_time_end(snpgpu_ctx *ctx, int kernel, hipEvent_t a) {
    if (!a) return;
    hipEvent_t b = take_event(ctx);
    (void)hipEventRecord(b, ctx->stream);
    ctx->timed.push_back({kernel, a, b});
}

extern "C" {

int snpgpu_ctx_kernel_timing(snpgpu_ctx *ctx, int enable) {
    if (!ctx) return SNPGPU_E_ARG;
    ctx->time_kernels = enable != 0;
    return SNPGPU_OK;
}

int snpgpu_ctx_kernel_time_ms(snpgpu_ctx *ctx, int kernel, float *total_ms, uint32_t *launches) {
    if (!ctx || !total_ms || !launches) return SNPGPU_E_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float sum = 0;
    uint32_t n = 0;
    std::vector<snpgpu_ctx::Timed> keep;
    for (auto &t : ctx->timed) {
        if (t.kernel != kernel) { keep.push_back(t); continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { sum += ms; ++n; }
        ctx->event_pool.push_back(t.a);
        ctx->event_pool.push_back(t.b);
    }
    ctx->timed.swap(keep);
    *total_ms = sum;
    *launches = n;
    return SNPGPU_OK;
}

int snpgpu_abi_version(void) { return SNPGPU_ABI_VERSION; }

int snpgpu_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int snpgpu_ctx_create(int device, snpgpu_ctx **out) {
    if (!out) return SNPGPU_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) return SNPGPU_E_HIP;   // no CPU fallback
    snpgpu_ctx *ctx = new snpgpu_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return SNPGPU_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    hipEventCreate(&ctx->ev_start);
    hipEventCreate(&ctx->ev_stop);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    *out = ctx;
    return SNPGPU_OK;
}

void snpgpu_ctx_destroy(snpgpu_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    snpgpu_comm_release(ctx);
    snpgpu_stream_pool_destroy(ctx);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->d_spill) hipFree(ctx->d_spill);
    for (auto &t : ctx->timed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->ev_start) hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// ---- positions with more than SNPGPU_MAX_SYMS symbols ----------------------------------------------------------------
}  // extern "C"  (closed for the internal helper; reopened below)
int snpgpu_spill_begin(snpgpu_ctx *ctx) {
    if (!ctx->d_spill || ctx->spill_want > ctx->spill_cap) {
        if (ctx->d_spill) {                                     // a call ran out of records: a larger arena for the repeat
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_spill);
            ctx->d_spill = nullptr;
            ctx->spill_cap = 0;
        }
        void *d = nullptr;
        const uint32_t cap = ctx->spill_want > SNPGPU_SPILL_CAP ? ctx->spill_want : SNPGPU_SPILL_CAP;
        const size_t bytes = sizeof(snpgpu_symbol_spill) * (size_t)cap + 256;
        hipError_t e = hipMalloc(&d, bytes);
        if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipMalloc(%zu) for the symbol spill failed: %s", bytes, hipGetErrorString(e));
        ctx->d_spill = (snpgpu_symbol_spill *)d;
        ctx->d_spill_n = (uint32_t *)((char *)d + sizeof(snpgpu_symbol_spill) * (size_t)cap);
        ctx->spill_cap = cap;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_spill_n, 0, 4, ctx->stream));
    return SNPGPU_OK;
}
extern "C" {

int snpgpu_symbol_spill_read(snpgpu_ctx *ctx, snpgpu_symbol_spill *out, uint32_t capacity, uint32_t *out_n) {
    if (!ctx || !out_n || (capacity && !out)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    *out_n = 0;
    if (!ctx->d_spill) return SNPGPU_OK;                       // no call has asked for per-site records yet
    HIP_TRY(ctx, snpgpu_enter(ctx));
    uint32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, ctx->d_spill_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out_n = n;                                                // what the calls asked for, which may be more than the arena held
    if (n > ctx->spill_cap) {
        // positions past the last record carry "no room" in their own record: the caller repeats the call, and the arena will
        // hold what this one asked for (and a quarter more: other files of a batch may differ)
        const uint64_t want = (uint64_t)n + n / 4 + 64;
        ctx->spill_want = want > 0xFFFFFEull ? 0xFFFFFEu : (uint32_t)want;
        if (n > 0xFFFFFEu) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "%u positions of one call need a spill record: more than a record's 24 index bits address", n);
        return SNPGPU_OK;
    }
    const uint32_t k = n < capacity ? n : capacity;
    if (k) {
        HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_spill, sizeof(snpgpu_symbol_spill) * (size_t)k, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SNPGPU_OK;
}

uint32_t snpgpu_symbol_spill_capacity(const snpgpu_ctx *ctx) { return ctx ? (ctx->spill_cap ? ctx->spill_cap : SNPGPU_SPILL_CAP) : 0; }

const char *snpgpu_last_error(const snpgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int snpgpu_ctx_set_stream(snpgpu_ctx *ctx, void *hip_stream) {
    if (!ctx) return SNPGPU_E_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;      // NULL is HIP's legacy default stream (what torch uses by default)
    return SNPGPU_OK;
}

int snpgpu_ctx_reset_stream(snpgpu_ctx *ctx) {
    if (!ctx) return SNPGPU_E_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return SNPGPU_OK;
}

int snpgpu_ctx_sync(snpgpu_ctx *ctx) {
    if (!ctx) return SNPGPU_E_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SNPGPU_OK;
}

int snpgpu_timer_start(snpgpu_ctx *ctx) {
    if (!ctx) return SNPGPU_E_ARG;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return SNPGPU_OK;
}

int snpgpu_timer_stop_ms(snpgpu_ctx *ctx, float *out_ms) {
    if (!ctx || !out_ms) return SNPGPU_E_ARG;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    HIP_TRY(ctx, hipEventElapsedTime(out_ms, ctx->ev_start, ctx->ev_stop));
    return SNPGPU_OK;
}

}  // extern "C"

// ---- site set ---------------------------------------------------------------------------------

__global__ void k_siteset_setbits(const uint64_t *keys, uint32_t n, const uint64_t *bit_off, uint32_t *bitmap) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys[i];
    uint64_t bit = bit_off[k >> 32] + (uint32_t)k;
    atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
}

__global__ void k_popcount_words(const uint32_t *bitmap, uint32_t *out, uint64_t n_words) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) out[i] = __popc(bitmap[i]);
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" {

int snpgpu_siteset_create(snpgpu_ctx *ctx, const uint8_t *contig_names, const uint32_t *contig_name_off,
                          uint32_t n_contigs, const uint64_t *site_keys, const uint8_t *site_flags,
                          uint32_t n_sites, snpgpu_siteset **out) {
    if (!ctx || !out) return SNPGPU_E_ARG;
    *out = nullptr;
    if (n_contigs && (!contig_names || !contig_name_off)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "contig table missing");
    if (n_sites && (!site_keys || !site_flags)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "site arrays missing");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    // validate ordering: names sorted + unique, keys strictly increasing, contig ids in range
    for (uint32_t c = 1; c < n_contigs; ++c) {
        uint32_t a0 = contig_name_off[c - 1], a1 = contig_name_off[c], b1 = contig_name_off[c + 1];
        uint32_t la = a1 - a0, lb = b1 - a1;
        int cmp = memcmp(contig_names + a0, contig_names + a1, la < lb ? la : lb);
        if (cmp > 0 || (cmp == 0 && la >= lb)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "contig names must be sorted bytewise and unique");
    }
    std::vector<uint32_t> max_pos(n_contigs ? n_contigs : 1, 0);
    std::vector<uint8_t> has(n_contigs ? n_contigs : 1, 0);
    for (uint32_t i = 0; i < n_sites; ++i) {
        uint64_t k = site_keys[i];
        if ((k >> 32) >= n_contigs) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "site key %u names contig %llu of %u", i, (unsigned long long)(k >> 32), n_contigs);
        if (i && site_keys[i - 1] >= k) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "site keys must be strictly increasing");
        max_pos[k >> 32] = (uint32_t)k;    // keys ascending => last one per contig is the max
        has[k >> 32] = 1;
    }
    std::vector<uint64_t> bit_off(n_contigs ? n_contigs : 1, 0);
    uint64_t total_bits = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        bit_off[c] = total_bits;
        total_bits += has[c] ? (uint64_t)max_pos[c] + 1 : 1;
    }
    if (total_bits > (1ull << 35)) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "site positions span %llu bits (> 2^35)", (unsigned long long)total_bits);
    uint64_t n_words = (total_bits + 31) / 32 + 1;
    uint32_t names_bytes = n_contigs ? contig_name_off[n_contigs] : 0;

    // one blob: names | name_off | bit_off | max_pos | bitmap | rank | flags | keys(tmp) | site_line
    size_t o_names = 0;
    size_t o_noff = align_up(o_names + names_bytes, 16);
    size_t o_boff = align_up(o_noff + 4ull * (n_contigs + 1), 16);
    size_t o_maxp = align_up(o_boff + 8ull * (n_contigs ? n_contigs : 1), 16);
    size_t o_bmap = align_up(o_maxp + 4ull * (n_contigs ? n_contigs : 1), 256);
    size_t o_rank = align_up(o_bmap + 4 * n_words, 256);
    size_t o_flag = align_up(o_rank + 4 * n_words, 256);
    size_t o_line = align_up(o_flag + (n_sites ? n_sites : 1), 256);
    size_t o_keys = align_up(o_line + 8ull * (n_sites ? n_sites : 1), 256);
    size_t o_queue = align_up(o_keys + 8ull * (n_sites ? n_sites : 1), 256);
    size_t o_ctl = o_queue + 8ull * SNPGPU_SLOW_QUEUE_CAP;
    size_t total = o_ctl + 256;

    snpgpu_siteset *ss = new snpgpu_siteset();
    ss->ctx = ctx;
    ss->n_sites = n_sites;
    ss->total_bits = total_bits;
    if (n_sites) ss->h_flags.assign(site_flags, site_flags + n_sites);
    hipError_t e = hipMalloc(&ss->blob, total);
    if (e != hipSuccess) {
        delete ss;
        return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipMalloc(%zu) for site set failed: %s", total, hipGetErrorString(e));
    }
    char *b = (char *)ss->blob;
    hipStream_t st = ctx->stream;
#define SS_TRY(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e2_ = (expr);                                                                        \
        if (e2_ != hipSuccess) {                                                                        \
            hipFree(ss->blob);                                                                          \
            delete ss;                                                                                  \
            return snpgpu_set_error(ctx, SNPGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e2_)); \
        }                                                                                               \
    } while (0)
    SS_TRY(hipMemsetAsync(b + o_bmap, 0, o_flag - o_bmap, st));
    if (names_bytes) SS_TRY(hipMemcpyAsync(b + o_names, contig_names, names_bytes, hipMemcpyHostToDevice, st));
    if (n_contigs) {
        SS_TRY(hipMemcpyAsync(b + o_noff, contig_name_off, 4ull * (n_contigs + 1), hipMemcpyHostToDevice, st));
        SS_TRY(hipMemcpyAsync(b + o_boff, bit_off.data(), 8ull * n_contigs, hipMemcpyHostToDevice, st));
        SS_TRY(hipMemcpyAsync(b + o_maxp, max_pos.data(), 4ull * n_contigs, hipMemcpyHostToDevice, st));
    }
    if (n_sites) {
        SS_TRY(hipMemcpyAsync(b + o_flag, site_flags, n_sites, hipMemcpyHostToDevice, st));
        SS_TRY(hipMemcpyAsync(b + o_keys, site_keys, 8ull * n_sites, hipMemcpyHostToDevice, st));
        k_siteset_setbits<<<(n_sites + 255) / 256, 256, 0, st>>>((const uint64_t *)(b + o_keys), n_sites,
                                                                   (const uint64_t *)(b + o_boff), (uint32_t *)(b + o_bmap));
    }
    // rank[w] = popcount of words < w
    k_popcount_words<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>((const uint32_t *)(b + o_bmap), (uint32_t *)(b + o_rank), n_words);
    {
        void *scan_ws = nullptr;
        if (snpgpu_scratch(ctx, 4 * prim_scan_workspace_words(n_words) + 256, &scan_ws) != SNPGPU_OK) {
            hipFree(ss->blob);
            delete ss;
            return SNPGPU_E_NOMEM;
        }
        prim_exclusive_scan_u32(st, (const uint32_t *)(b + o_rank), (uint32_t *)(b + o_rank), n_words, (uint32_t *)scan_ws, nullptr);
    }
    SS_TRY(hipGetLastError());
    SS_TRY(hipStreamSynchronize(st));
#undef SS_TRY
    ss->dev.names = (const uint8_t *)(b + o_names);
    ss->dev.name_off = (const uint32_t *)(b + o_noff);
    ss->dev.bit_off = (const uint64_t *)(b + o_boff);
    ss->dev.max_pos = (const uint32_t *)(b + o_maxp);
    ss->dev.bitmap = (const uint32_t *)(b + o_bmap);
    ss->dev.rank = (const uint32_t *)(b + o_rank);
    ss->dev.flags = (const uint8_t *)(b + o_flag);
    ss->dev.n_words = n_words;
    ss->dev.n_contigs = n_contigs;
    ss->dev.n_sites = n_sites;
    ss->site_line = (uint64_t *)(b + o_line);
    ss->slow_queue = (uint64_t *)(b + o_queue);
    ss->slow_ctl = (uint32_t *)(b + o_ctl);
    *out = ss;
    return SNPGPU_OK;
}

void snpgpu_siteset_destroy(snpgpu_siteset *ss) {
    if (!ss) return;
    if (ss->ctx) {
        hipSetDevice(ss->ctx->device);
        hipStreamSynchronize(ss->ctx->stream);
    }
    if (ss->blob) hipFree(ss->blob);
    delete ss;
}

uint32_t snpgpu_siteset_size(const snpgpu_siteset *ss) { return ss ? ss->n_sites : 0; }

}  // extern "C"
