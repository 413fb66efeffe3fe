// The exchange steps of the sharded hot path behind the C ABI: RCCL collectives on the context's stream, and the hand-written
// kernels that feed them.
//
// The path shards by samples (the reference's fan-out is one process per sample, run.py:704-718); ranks meet in two
// all-gathers — C1, the per-rank SNP site keys (variable length); C2, the per-rank rows of the packed consensus matrix — and one
// all-to-all of distance tiles (the row-band exchange, snp_pipeline_amd/sharding.py).  Rounds 1-4 made those calls from Python
// through torch.distributed, with ATen index / index_put kernels packing the tiles; a maintainer who binds libsnpgpu.so by ctypes
// had no multi-GPU path without PyTorch.  Here they are entry points of the library: one process per GPU, snpgpu_comm_init with a
// 128-byte id that rank 0 made with snpgpu_comm_unique_id and handed to the others by whatever means the host program has (a
// file, an environment variable, MPI, a socket); every collective is enqueued on the context's stream like a kernel.
//
// librccl.so is loaded with dlopen on first use, not linked: the library loads — and the single-GPU path runs — on hosts without
// RCCL, and in a process that already holds a librccl (PyTorch ships one) the loaded one is used, so there is one RCCL and one HIP
// runtime in the process.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("SNPGPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            (void)dlerror();
        }
        if (!r.handle) return;
#define RCCL_SYM(field, name)                                                              \
    do {                                                                                   \
        *(void **)(&r.field) = dlsym(r.handle, name);                                      \
        if (!r.field) { r.handle = nullptr; return; }                                      \
    } while (0)
        RCCL_SYM(GetVersion, "ncclGetVersion");
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        RCCL_SYM(CommInitRank, "ncclCommInitRank");
        RCCL_SYM(CommDestroy, "ncclCommDestroy");
        RCCL_SYM(CommAbort, "ncclCommAbort");
        RCCL_SYM(CommCount, "ncclCommCount");
        RCCL_SYM(AllGather, "ncclAllGather");
        RCCL_SYM(Send, "ncclSend");
        RCCL_SYM(Recv, "ncclRecv");
        RCCL_SYM(GroupStart, "ncclGroupStart");
        RCCL_SYM(GroupEnd, "ncclGroupEnd");
        RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    });
    return r.handle ? &r : nullptr;
}

const char *rccl_why() { return "RCCL is not available (librccl.so could not be loaded; SNPGPU_RCCL_LIB names another place)"; }

#define RCCL_TRY(ctx, R, expr)                                                                                              \
    do {                                                                                                                    \
        ncclResult_t r_ = (expr);                                                                                           \
        if (r_ != ncclSuccess) return snpgpu_set_error((ctx), SNPGPU_E_HIP, "%s failed: %s", #expr, (R)->GetErrorString(r_)); \
    } while (0)

// Inside a ncclGroupStart / ncclGroupEnd bracket: remember the first failure and go on to the bracket's end — returning from inside
// would leave the thread's group open, every later collective of the thread would be queued into it and never launch (a hang where
// an error belongs).  RCCL_GROUP_END closes the bracket whatever happened and returns the first failure.
#define RCCL_IN_GROUP(first, what, expr)                                                                                    \
    do {                                                                                                                    \
        if ((first) == ncclSuccess) {                                                                                       \
            (first) = (expr);                                                                                               \
            if ((first) != ncclSuccess) (what) = #expr;                                                                     \
        }                                                                                                                   \
    } while (0)
#define RCCL_GROUP_END(ctx, R, first, what)                                                                                 \
    do {                                                                                                                    \
        const ncclResult_t end_ = (R)->GroupEnd();                                                                          \
        if ((first) != ncclSuccess) return snpgpu_set_error((ctx), SNPGPU_E_HIP, "%s failed: %s", (what), (R)->GetErrorString(first)); \
        if (end_ != ncclSuccess) return snpgpu_set_error((ctx), SNPGPU_E_HIP, "ncclGroupEnd failed: %s", (R)->GetErrorString(end_));  \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};

// ---- distance tiles in and out of the matrix --------------------------------------------------------------------------------
// tile t of the list = the 128 x 128 block (rows[t], cols[t]) of an n_padded x n_padded int32 matrix (tile units).  One workgroup
// per tile and turn, 16-byte accesses: a row of a tile is 512 contiguous bytes.
constexpr uint32_t TILE = 128;

__global__ __launch_bounds__(256) void k_tiles_gather(const int32_t *__restrict__ m, uint32_t n_padded, const uint32_t *__restrict__ rows,
                                                      const uint32_t *__restrict__ cols, uint32_t n_tiles, int32_t *__restrict__ out) {
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int4 *src = (const int4 *)(m + (size_t)rows[t] * TILE * n_padded + (size_t)cols[t] * TILE);
        int4 *dst = (int4 *)(out + (size_t)t * TILE * TILE);
        for (uint32_t i = threadIdx.x; i < TILE * TILE / 4; i += blockDim.x) {
            const uint32_t r = i / (TILE / 4), c = i % (TILE / 4);
            dst[i] = src[(size_t)r * (n_padded / 4) + c];
        }
    }
}

__global__ __launch_bounds__(256) void k_tiles_scatter(const int32_t *__restrict__ in, const uint32_t *__restrict__ rows, const uint32_t *__restrict__ cols,
                                                       uint32_t n_tiles, int32_t *__restrict__ m, uint32_t n_padded) {
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int4 *src = (const int4 *)(in + (size_t)t * TILE * TILE);
        int4 *dst = (int4 *)(m + (size_t)rows[t] * TILE * n_padded + (size_t)cols[t] * TILE);
        for (uint32_t i = threadIdx.x; i < TILE * TILE / 4; i += blockDim.x) {
            const uint32_t r = i / (TILE / 4), c = i % (TILE / 4);
            dst[(size_t)r * (n_padded / 4) + c] = src[i];
        }
    }
}

// ---- what the one-job pipeline asks about a group of samples after scan + call (hot_path.py) ---------------------------------
// Per sample: is one of the positions IT is asked about malformed (snplist / preserved list: `wanted`; its own removed positions:
// its stretch of the exclude lists)?  how many positions of the set have a pileup line?  how many have a record in the spill?
// One workgroup per sample.
__global__ __launch_bounds__(256) void k_group_check(const uint8_t *__restrict__ filters, const uint8_t *__restrict__ counts, const uint64_t *__restrict__ line_off,
                                                     const uint8_t *__restrict__ wanted, const uint32_t *__restrict__ excl_off,
                                                     const uint32_t *__restrict__ excl_slots, uint32_t n_samples, uint32_t n_sites, int64_t *__restrict__ out) {
    __shared__ unsigned long long part[3][4];
    for (uint32_t s = blockIdx.x; s < n_samples; s += gridDim.x) {
        unsigned long long bad = 0, lines = 0, spilled = 0;
        auto is_bad = [&](uint64_t at) -> bool {
            // a record's status byte (offset 23) above ST_OK, else bit 7 of the filter byte
            return counts ? counts[at * sizeof(snpgpu_site_counts) + 23] > SNPGPU_ST_OK : (filters[at] & 0x80u) != 0;
        };
        for (uint32_t i = threadIdx.x; i < n_sites; i += blockDim.x) {
            const uint64_t at = (uint64_t)s * n_sites + i;
            if (wanted[i] && is_bad(at)) bad = 1;
            lines += line_off[at] != 0 ? 1u : 0u;
            if (counts) {                                       // bytes 17-19 of a record: nonzero = the position has a record in the spill
                const uint8_t *c = counts + at * sizeof(snpgpu_site_counts);
                spilled += (c[17] | c[18] | c[19]) ? 1u : 0u;
            }
        }
        if (excl_off && excl_slots)
            for (uint32_t k = excl_off[s] + threadIdx.x; k < excl_off[s + 1]; k += blockDim.x) {
                const uint32_t slot = excl_slots[k];
                if (slot < n_sites && is_bad((uint64_t)s * n_sites + slot)) bad = 1;
            }
        for (int o = 32; o; o >>= 1) { bad |= __shfl_xor(bad, o); lines += __shfl_xor(lines, o); spilled += __shfl_xor(spilled, o); }
        if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = bad; part[1][threadIdx.x >> 6] = lines; part[2][threadIdx.x >> 6] = spilled; }
        __syncthreads();
        if (threadIdx.x == 0) {
            out[3 * (size_t)s + 0] = (int64_t)((part[0][0] | part[0][1] | part[0][2] | part[0][3]) ? 1 : 0);
            out[3 * (size_t)s + 1] = (int64_t)(part[1][0] + part[1][1] + part[1][2] + part[1][3]);
            out[3 * (size_t)s + 2] = (int64_t)(part[2][0] + part[2][1] + part[2][2] + part[2][3]);
        }
        __syncthreads();
    }
}

}  // namespace

void snpgpu_comm_release(snpgpu_ctx *ctx) {                    // (ctx.hip: before the context's stream goes)
    if (!ctx || !ctx->comm) return;
    Comm *c = (Comm *)ctx->comm;
    Rccl *R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    ctx->comm = nullptr;
}

extern "C" {

int snpgpu_comm_available(void) { return rccl() ? 1 : 0; }

int snpgpu_comm_version(int *out_version) {
    Rccl *R = rccl();
    if (!R || !out_version) return SNPGPU_E_UNSUPPORTED;
    return R->GetVersion(out_version) == ncclSuccess ? SNPGPU_OK : SNPGPU_E_HIP;
}

int snpgpu_comm_unique_id(void *out_id) {
    Rccl *R = rccl();
    if (!out_id) return SNPGPU_E_ARG;
    if (!R) return SNPGPU_E_UNSUPPORTED;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return SNPGPU_E_HIP;
    static_assert(sizeof id == SNPGPU_COMM_ID_BYTES, "id size");
    memcpy(out_id, &id, sizeof id);
    return SNPGPU_OK;
}

int snpgpu_comm_init(snpgpu_ctx *ctx, int rank, int nranks, const void *unique_id) {
    if (!ctx || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "bad communicator arguments");
    Rccl *R = rccl();
    if (!R) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "%s", rccl_why());
    HIP_TRY(ctx, snpgpu_enter(ctx));
    snpgpu_comm_release(ctx);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks;
    ncclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return snpgpu_set_error(ctx, SNPGPU_E_HIP, "ncclCommInitRank failed: %s", R->GetErrorString(r)); }
    ctx->comm = c;
    return SNPGPU_OK;
}

void snpgpu_comm_destroy(snpgpu_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    snpgpu_comm_release(ctx);
}

// Give the communicator up WITHOUT waiting for the stream: what a collective that never completes leaves behind is cancelled
// (ncclCommAbort), so that the process can go on by another route or end with a message instead of hanging in a synchronize.
void snpgpu_comm_abort(snpgpu_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    Comm *c = (Comm *)ctx->comm;
    Rccl *R = rccl();
    (void)hipSetDevice(ctx->device);
    if (R && c->comm) (void)R->CommAbort(c->comm);
    delete c;
    ctx->comm = nullptr;
}

int snpgpu_comm_info(const snpgpu_ctx *ctx, int *out_rank, int *out_nranks, int *out_count_from_rccl) {
    if (!ctx) return SNPGPU_E_ARG;
    const Comm *c = (const Comm *)ctx->comm;
    if (out_rank) *out_rank = c ? c->rank : 0;
    if (out_nranks) *out_nranks = c ? c->nranks : 1;
    if (out_count_from_rccl) {
        *out_count_from_rccl = 0;
        Rccl *R = rccl();
        if (c && R) (void)R->CommCount(c->comm, out_count_from_rccl);
    }
    return SNPGPU_OK;
}

// every rank contributes the same number of bytes; block r of d_recv comes from rank r (C2 when the blocks are even)
int snpgpu_allgather(snpgpu_ctx *ctx, const void *d_send, void *d_recv, size_t bytes_per_rank) {
    if (!ctx || !ctx->comm) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "no communicator: snpgpu_comm_init first");
    if (!bytes_per_rank) return SNPGPU_OK;
    if (!d_send || !d_recv) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    Rccl *R = rccl();
    Comm *c = (Comm *)ctx->comm;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    RCCL_TRY(ctx, R, R->AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, ctx->stream));
    return SNPGPU_OK;
}

// blocks of different sizes: rank r contributes bytes[r] bytes, which land at d_recv + offsets[r] on every rank (bytes / offsets:
// host arrays of nranks entries, the same on every rank).  C1 (site keys), and C2 when the last rank holds fewer rows.  One group of
// point-to-point transfers: xGMI is point-to-point, every pair has its own link.
int snpgpu_allgatherv(snpgpu_ctx *ctx, const void *d_send, void *d_recv, const uint64_t *bytes, const uint64_t *offsets) {
    if (!ctx || !ctx->comm) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "no communicator: snpgpu_comm_init first");
    if (!bytes || !offsets) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    Rccl *R = rccl();
    Comm *c = (Comm *)ctx->comm;
    uint64_t total = 0;
    for (int p = 0; p < c->nranks; ++p) total += bytes[p];
    if (!total) return SNPGPU_OK;                                // (every rank sees the same sizes: nobody has anything)
    if (!d_recv) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const uint64_t mine = bytes[c->rank];
    if (mine && !d_send) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (mine && (const char *)d_send != (const char *)d_recv + offsets[c->rank])
        HIP_TRY(ctx, hipMemcpyAsync((char *)d_recv + offsets[c->rank], d_send, mine, hipMemcpyDeviceToDevice, ctx->stream));
    RCCL_TRY(ctx, R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (int p = 0; p < c->nranks; ++p) {
        if (p == c->rank) continue;
        if (mine) RCCL_IN_GROUP(first, what, R->Send(d_send, mine, ncclUint8, p, c->comm, ctx->stream));
        if (bytes[p]) RCCL_IN_GROUP(first, what, R->Recv((char *)d_recv + offsets[p], bytes[p], ncclUint8, p, c->comm, ctx->stream));
    }
    RCCL_GROUP_END(ctx, R, first, what);
    return SNPGPU_OK;
}

// send_bytes[p] bytes from d_send (blocks in rank order, packed) go to rank p; recv_bytes[p] bytes from rank p arrive in d_recv
// (blocks in rank order, packed).  The row-band exchange of the distance tiles.
int snpgpu_alltoallv(snpgpu_ctx *ctx, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes) {
    if (!ctx || !ctx->comm) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "no communicator: snpgpu_comm_init first");
    if (!send_bytes || !recv_bytes) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    Rccl *R = rccl();
    Comm *c = (Comm *)ctx->comm;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    uint64_t so = 0, ro = 0, my_so = 0, my_ro = 0, total_s = 0, total_r = 0;
    for (int p = 0; p < c->rank; ++p) { my_so += send_bytes[p]; my_ro += recv_bytes[p]; }
    for (int p = 0; p < c->nranks; ++p) { total_s += send_bytes[p]; total_r += recv_bytes[p]; }
    if ((total_s && !d_send) || (total_r && !d_recv)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (send_bytes[c->rank] != recv_bytes[c->rank]) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "a rank's block for itself has two sizes");
    if (send_bytes[c->rank])
        HIP_TRY(ctx, hipMemcpyAsync((char *)d_recv + my_ro, (const char *)d_send + my_so, send_bytes[c->rank], hipMemcpyDeviceToDevice, ctx->stream));
    RCCL_TRY(ctx, R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (int p = 0; p < c->nranks; ++p) {
        if (p != c->rank) {
            if (send_bytes[p]) RCCL_IN_GROUP(first, what, R->Send((const char *)d_send + so, send_bytes[p], ncclUint8, p, c->comm, ctx->stream));
            if (recv_bytes[p]) RCCL_IN_GROUP(first, what, R->Recv((char *)d_recv + ro, recv_bytes[p], ncclUint8, p, c->comm, ctx->stream));
        }
        so += send_bytes[p];
        ro += recv_bytes[p];
    }
    RCCL_GROUP_END(ctx, R, first, what);
    return SNPGPU_OK;
}

// Wait until everything enqueued on the context's stream is done, at most timeout_ms: SNPGPU_E_TIMEOUT tells a collective that
// hangs (a rank that never arrived) from one that is slow, without blocking the host thread for good.
int snpgpu_stream_wait(snpgpu_ctx *ctx, uint32_t timeout_ms) {
    if (!ctx) return SNPGPU_E_ARG;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) return SNPGPU_OK;
        if (e != hipErrorNotReady) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
        if (ms > timeout_ms) return snpgpu_set_error(ctx, SNPGPU_E_TIMEOUT, "the stream was still busy after %u ms", timeout_ms);
        timespec nap = {0, ms < 2.0 ? 20000 : 500000};
        nanosleep(&nap, nullptr);
    }
}

int snpgpu_tiles_gather_dev(snpgpu_ctx *ctx, const int32_t *d_matrix, uint32_t n_padded, const uint32_t *d_tile_rows, const uint32_t *d_tile_cols,
                            uint32_t n_tiles, int32_t *d_out) {
    if (!ctx) return SNPGPU_E_ARG;
    if (!n_tiles) return SNPGPU_OK;
    if (!d_matrix || !d_tile_rows || !d_tile_cols || !d_out || n_padded % TILE) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "bad tile arguments");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const unsigned cap = (unsigned)ctx->n_cu * 8;
    k_tiles_gather<<<n_tiles < cap ? n_tiles : cap, 256, 0, ctx->stream>>>(d_matrix, n_padded, d_tile_rows, d_tile_cols, n_tiles, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_tiles_scatter_dev(snpgpu_ctx *ctx, const int32_t *d_tiles, const uint32_t *d_tile_rows, const uint32_t *d_tile_cols, uint32_t n_tiles,
                             int32_t *d_matrix, uint32_t n_padded) {
    if (!ctx) return SNPGPU_E_ARG;
    if (!n_tiles) return SNPGPU_OK;
    if (!d_matrix || !d_tile_rows || !d_tile_cols || !d_tiles || n_padded % TILE) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "bad tile arguments");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const unsigned cap = (unsigned)ctx->n_cu * 8;
    k_tiles_scatter<<<n_tiles < cap ? n_tiles : cap, 256, 0, ctx->stream>>>(d_tiles, d_tile_rows, d_tile_cols, n_tiles, d_matrix, n_padded);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_group_check_dev(snpgpu_ctx *ctx, const uint8_t *d_filters, const snpgpu_site_counts *d_counts, const uint64_t *d_line_off,
                           const uint8_t *d_wanted, const uint32_t *d_excl_off, const uint32_t *d_excl_slots, uint32_t n_samples, uint32_t n_sites,
                           int64_t *d_out) {
    if (!ctx) return SNPGPU_E_ARG;
    if (!n_samples) return SNPGPU_OK;
    if (!d_out || (n_sites && (!d_line_off || !d_wanted || (!d_filters && !d_counts)))) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const unsigned cap = (unsigned)ctx->n_cu * 8;
    k_group_check<<<n_samples < cap ? n_samples : cap, 256, 0, ctx->stream>>>(d_filters, (const uint8_t *)d_counts, d_line_off, d_wanted, d_excl_off, d_excl_slots,
                                                                             n_samples, n_sites, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

}  // extern "C"
