// K3 dense-window filter and region algebra, K4 site merge.
//
//   find_dense_regions      snppipeline/filter_regions.py:17-71   -> k_dense_flags + compaction
//   utils.merge_regions     snppipeline/utils.py:1267-1282        -> sort by (group,start,end) + segmented running max
//   utils.in_region         snppipeline/utils.py:1314-1318        -> binary search in the merged list
//   merge_sites union       snppipeline/merge_sites.py:91-117     -> radix sort of (contig<<32|pos) + unique + CSR
//
// These are small, latency-bound steps (KBs to a few MBs); they use rocPRIM/hipCUB device scans and radix sorts
// around a few hand-written elementwise kernels.  All entry points take host pointers and are synchronous.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "internal.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T *as() { return (T *)p; }
};

#define R_TRY(ctx, expr)                                                                                \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return snpgpu_set_error((ctx), SNPGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

__global__ void k_dense_flags(const int64_t *pos, const uint32_t *seg_off, uint32_t n_segs, uint32_t n_pos,
                              const int32_t *max_snps, const int32_t *window, uint32_t n_rules, uint32_t *flag) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_pos * n_rules) return;
    uint32_t i = (uint32_t)(g / n_rules), r = (uint32_t)(g % n_rules);
    // segment of i: last s with seg_off[s] <= i
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    uint32_t end = seg_off[lo + 1];
    int64_t m = max_snps[r];
    uint32_t f = 0;
    if (m >= 0 && (uint64_t)i + (uint64_t)m < end) f = (pos[i] + (int64_t)window[r] - 1 >= pos[i + m]) ? 1u : 0u;
    flag[g] = f;
}

__global__ void k_dense_emit(const int64_t *pos, const uint32_t *seg_off, uint32_t n_segs, uint32_t n_pos,
                             const int32_t *max_snps, uint32_t n_rules, const uint32_t *flag, const uint32_t *slot,
                             int64_t *out_start, int64_t *out_end, uint32_t *out_seg) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_pos * n_rules || !flag[g]) return;
    uint32_t i = (uint32_t)(g / n_rules), r = (uint32_t)(g % n_rules);
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    uint32_t o = slot[g];
    out_start[o] = pos[i];
    out_end[o] = pos[i + max_snps[r]];
    out_seg[o] = lo;
}

// (segment << 40) | position: radix-sorting these sorts every segment's positions (filter_regions.py:425 sorted())
__global__ void k_seg_keys(const int64_t *pos, const uint32_t *seg_off, uint32_t n_segs, uint32_t n_pos, uint64_t *keys, uint32_t *bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    int64_t p = pos[i];
    if (p < 0 || p >= (1ll << 40)) { *bad = 1; p = 0; }
    keys[i] = ((uint64_t)lo << 40) | (uint64_t)p;
}

__global__ void k_seg_unkeys(const uint64_t *keys, int64_t *pos, uint32_t n_pos) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pos) pos[i] = (int64_t)(keys[i] & ((1ull << 40) - 1));
}

__global__ void k_iota(uint32_t *v, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// order-preserving map int64 -> uint64 for radix sorting
__global__ void k_gather_key_i64(const int64_t *src, const uint32_t *perm, uint64_t *dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint64_t)src[perm[i]] ^ 0x8000000000000000ull;
}

__global__ void k_gather_key_u32(const uint32_t *src, const uint32_t *perm, uint64_t *dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

struct GroupMax {
    uint32_t group;
    int64_t maxend;
};
struct GroupMaxOp {
    __host__ __device__ GroupMax operator()(const GroupMax &a, const GroupMax &b) const {
        GroupMax r;
        r.group = b.group;
        r.maxend = (a.group == b.group && a.maxend > b.maxend) ? a.maxend : b.maxend;
        return r;
    }
};

__global__ void k_merge_prepare(const uint32_t *group, const int64_t *end, const uint32_t *perm, GroupMax *gm, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { gm[i].group = group[perm[i]]; gm[i].maxend = end[perm[i]]; }
}

// head[i] = 1 when sorted interval i opens a new merged region
__global__ void k_merge_heads(const int64_t *start, const uint32_t *perm, const GroupMax *run, uint32_t *head, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = 1;
    if (i > 0 && run[i - 1].group == run[i].group) {
        int64_t le = run[i - 1].maxend;
        int64_t s = start[perm[i]];
        h = (le < INT64_MAX && s <= le + 1) || (le == INT64_MAX) ? 0u : 1u;
    }
    head[i] = h;
}

__global__ void k_merge_emit(const int64_t *start, const uint32_t *perm, const GroupMax *run, const uint32_t *head,
                             const uint32_t *slot, uint32_t n, uint32_t *out_group, int64_t *out_start, int64_t *out_end) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o = slot[i] - 1;                     // inclusive scan of heads -> 1-based region id
    if (head[i]) { out_group[o] = run[i].group; out_start[o] = start[perm[i]]; }
    bool last = (i + 1 == n) || head[i + 1];
    if (last) out_end[o] = run[i].maxend;
}

__global__ void k_in_regions(const uint32_t *pos_group, const int64_t *pos, uint32_t n, const uint32_t *reg_off,
                             const int64_t *rs, const int64_t *re, uint32_t n_groups, uint8_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t g = pos_group[i];
    uint8_t f = 0;
    if (g < n_groups) {
        int64_t p = pos[i];
        uint32_t lo = reg_off[g], hi = reg_off[g + 1];
        while (lo < hi) {                         // last region with start <= p
            uint32_t mid = (lo + hi) >> 1;
            if (rs[mid] <= p) lo = mid + 1; else hi = mid;
        }
        if (lo > reg_off[g]) f = re[lo - 1] >= p;
    }
    out[i] = f;
}

__global__ void k_sites_flags(const uint64_t *keys, const uint32_t *samp, uint32_t n, uint32_t *new_key, uint32_t *new_pair) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool nk = i == 0 || keys[i] != keys[i - 1];
    new_key[i] = nk;
    new_pair[i] = nk || samp[i] != samp[i - 1];
}

__global__ void k_sites_emit(const uint64_t *keys, const uint32_t *samp, uint32_t n, const uint32_t *new_key, const uint32_t *new_pair,
                             const uint32_t *key_slot, const uint32_t *pair_slot, uint64_t *out_unique, uint32_t *out_off, uint32_t *out_carrier) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (new_pair[i]) out_carrier[pair_slot[i]] = samp[i];
    if (new_key[i]) { out_unique[key_slot[i]] = keys[i]; out_off[key_slot[i]] = pair_slot[i]; }
}

template <typename T>
hipError_t exclusive_sum(T *d, uint32_t n, hipStream_t st) {
    size_t tb = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d, d, (int)n, st);
    if (e != hipSuccess) return e;
    DevBuf tmp;
    if ((e = tmp.alloc(tb)) != hipSuccess) return e;
    e = hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, d, d, (int)n, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

hipError_t sort_pairs_u64_u32(uint64_t *kin, uint64_t *kout, uint32_t *vin, uint32_t *vout, uint32_t n, hipStream_t st) {
    size_t tb = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb, kin, kout, vin, vout, (int)n, 0, 64, st);
    if (e != hipSuccess) return e;
    DevBuf tmp;
    if ((e = tmp.alloc(tb)) != hipSuccess) return e;
    e = hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, kin, kout, vin, vout, (int)n, 0, 64, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
}

inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int snpgpu_dense_windows(snpgpu_ctx *ctx, const int64_t *positions, const uint32_t *seg_off, uint32_t n_segs,
                         const int32_t *max_snps, const int32_t *window, uint32_t n_rules,
                         int64_t *out_start, int64_t *out_end, uint32_t *out_seg, uint32_t *out_n) {
    if (!ctx || !out_n) return SNPGPU_E_ARG;
    *out_n = 0;
    if (n_segs == 0 || n_rules == 0) return SNPGPU_OK;
    if (!seg_off || !max_snps || !window) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null dense-window argument");
    uint32_t n_pos = seg_off[n_segs];
    if (n_pos == 0) return SNPGPU_OK;
    if (!positions || !out_start || !out_end || !out_seg) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null dense-window argument");
    R_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint64_t total = (uint64_t)n_pos * n_rules;
    if (total > 0x7FFFFFFFull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many (position, rule) pairs");
    DevBuf dpos, dseg, dms, dwin, dflag, dslot, dos, doe, dog;
    R_TRY(ctx, dpos.alloc(8ull * n_pos)); R_TRY(ctx, dseg.alloc(4ull * (n_segs + 1)));
    R_TRY(ctx, dms.alloc(4ull * n_rules)); R_TRY(ctx, dwin.alloc(4ull * n_rules));
    R_TRY(ctx, dflag.alloc(4 * total)); R_TRY(ctx, dslot.alloc(4 * total));
    R_TRY(ctx, hipMemcpyAsync(dpos.p, positions, 8ull * n_pos, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(dseg.p, seg_off, 4ull * (n_segs + 1), hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(dms.p, max_snps, 4ull * n_rules, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(dwin.p, window, 4ull * n_rules, hipMemcpyHostToDevice, st));
    {   // sort the positions of every segment on the device
        if (n_segs >= (1u << 24)) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many segments");
        DevBuf ka, kb, dbad;
        R_TRY(ctx, ka.alloc(8ull * n_pos)); R_TRY(ctx, kb.alloc(8ull * n_pos)); R_TRY(ctx, dbad.alloc(4));
        R_TRY(ctx, hipMemsetAsync(dbad.p, 0, 4, st));
        k_seg_keys<<<nblk(n_pos), 256, 0, st>>>(dpos.as<int64_t>(), dseg.as<uint32_t>(), n_segs, n_pos, ka.as<uint64_t>(), dbad.as<uint32_t>());
        size_t tb = 0;
        R_TRY(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, ka.as<uint64_t>(), kb.as<uint64_t>(), (int)n_pos, 0, 64, st));
        DevBuf tmp;
        R_TRY(ctx, tmp.alloc(tb));
        R_TRY(ctx, hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, ka.as<uint64_t>(), kb.as<uint64_t>(), (int)n_pos, 0, 64, st));
        k_seg_unkeys<<<nblk(n_pos), 256, 0, st>>>(kb.as<uint64_t>(), dpos.as<int64_t>(), n_pos);
        uint32_t bad = 0;
        R_TRY(ctx, hipMemcpyAsync(&bad, dbad.p, 4, hipMemcpyDeviceToHost, st));
        R_TRY(ctx, hipStreamSynchronize(st));
        if (bad) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "SNP position outside [0, 2^40)");
    }
    k_dense_flags<<<nblk(total), 256, 0, st>>>(dpos.as<int64_t>(), dseg.as<uint32_t>(), n_segs, n_pos, dms.as<int32_t>(), dwin.as<int32_t>(), n_rules, dflag.as<uint32_t>());
    R_TRY(ctx, hipMemcpyAsync(dslot.p, dflag.p, 4 * total, hipMemcpyDeviceToDevice, st));
    R_TRY(ctx, exclusive_sum(dslot.as<uint32_t>(), (uint32_t)total, st));
    uint32_t last_slot = 0, last_flag = 0;
    R_TRY(ctx, hipMemcpy(&last_slot, dslot.as<uint32_t>() + total - 1, 4, hipMemcpyDeviceToHost));
    R_TRY(ctx, hipMemcpy(&last_flag, dflag.as<uint32_t>() + total - 1, 4, hipMemcpyDeviceToHost));
    uint32_t n_out = last_slot + last_flag;
    *out_n = n_out;
    if (n_out == 0) return SNPGPU_OK;
    R_TRY(ctx, dos.alloc(8ull * n_out)); R_TRY(ctx, doe.alloc(8ull * n_out)); R_TRY(ctx, dog.alloc(4ull * n_out));
    k_dense_emit<<<nblk(total), 256, 0, st>>>(dpos.as<int64_t>(), dseg.as<uint32_t>(), n_segs, n_pos, dms.as<int32_t>(), n_rules,
                                              dflag.as<uint32_t>(), dslot.as<uint32_t>(), dos.as<int64_t>(), doe.as<int64_t>(), dog.as<uint32_t>());
    R_TRY(ctx, hipMemcpyAsync(out_start, dos.p, 8ull * n_out, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_end, doe.p, 8ull * n_out, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_seg, dog.p, 4ull * n_out, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    return SNPGPU_OK;
}

int snpgpu_merge_regions(snpgpu_ctx *ctx, const uint32_t *group, const int64_t *start, const int64_t *end,
                         uint32_t n, uint32_t *out_group, int64_t *out_start, int64_t *out_end, uint32_t *out_n) {
    if (!ctx || !out_n) return SNPGPU_E_ARG;
    *out_n = 0;
    if (n == 0) return SNPGPU_OK;
    if (!group || !start || !end || !out_group || !out_start || !out_end) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null merge argument");
    R_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf dg, ds, de, k0, k1, p0, p1, gm, run, head, slot, og, os, oe;
    R_TRY(ctx, dg.alloc(4ull * n)); R_TRY(ctx, ds.alloc(8ull * n)); R_TRY(ctx, de.alloc(8ull * n));
    R_TRY(ctx, k0.alloc(8ull * n)); R_TRY(ctx, k1.alloc(8ull * n)); R_TRY(ctx, p0.alloc(4ull * n)); R_TRY(ctx, p1.alloc(4ull * n));
    R_TRY(ctx, gm.alloc(sizeof(GroupMax) * (size_t)n)); R_TRY(ctx, run.alloc(sizeof(GroupMax) * (size_t)n));
    R_TRY(ctx, head.alloc(4ull * n)); R_TRY(ctx, slot.alloc(4ull * n));
    R_TRY(ctx, og.alloc(4ull * n)); R_TRY(ctx, os.alloc(8ull * n)); R_TRY(ctx, oe.alloc(8ull * n));
    R_TRY(ctx, hipMemcpyAsync(dg.p, group, 4ull * n, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(ds.p, start, 8ull * n, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(de.p, end, 8ull * n, hipMemcpyHostToDevice, st));
    // three stable radix passes: by end, then start, then group  ==  sorted() on (group, start, end)
    k_iota<<<nblk(n), 256, 0, st>>>(p0.as<uint32_t>(), n);
    k_gather_key_i64<<<nblk(n), 256, 0, st>>>(de.as<int64_t>(), p0.as<uint32_t>(), k0.as<uint64_t>(), n);
    R_TRY(ctx, sort_pairs_u64_u32(k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<uint32_t>(), p1.as<uint32_t>(), n, st));
    k_gather_key_i64<<<nblk(n), 256, 0, st>>>(ds.as<int64_t>(), p1.as<uint32_t>(), k0.as<uint64_t>(), n);
    R_TRY(ctx, sort_pairs_u64_u32(k0.as<uint64_t>(), k1.as<uint64_t>(), p1.as<uint32_t>(), p0.as<uint32_t>(), n, st));
    k_gather_key_u32<<<nblk(n), 256, 0, st>>>(dg.as<uint32_t>(), p0.as<uint32_t>(), k0.as<uint64_t>(), n);
    R_TRY(ctx, sort_pairs_u64_u32(k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<uint32_t>(), p1.as<uint32_t>(), n, st));
    uint32_t *perm = p1.as<uint32_t>();
    // segmented running max of end, heads, compaction
    k_merge_prepare<<<nblk(n), 256, 0, st>>>(dg.as<uint32_t>(), de.as<int64_t>(), perm, gm.as<GroupMax>(), n);
    {
        size_t tb = 0;
        R_TRY(ctx, hipcub::DeviceScan::InclusiveScan(nullptr, tb, gm.as<GroupMax>(), run.as<GroupMax>(), GroupMaxOp(), (int)n, st));
        DevBuf tmp;
        R_TRY(ctx, tmp.alloc(tb));
        R_TRY(ctx, hipcub::DeviceScan::InclusiveScan(tmp.p, tb, gm.as<GroupMax>(), run.as<GroupMax>(), GroupMaxOp(), (int)n, st));
        R_TRY(ctx, hipStreamSynchronize(st));
    }
    k_merge_heads<<<nblk(n), 256, 0, st>>>(ds.as<int64_t>(), perm, run.as<GroupMax>(), head.as<uint32_t>(), n);
    {
        size_t tb = 0;
        R_TRY(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tb, head.as<uint32_t>(), slot.as<uint32_t>(), (int)n, st));
        DevBuf tmp;
        R_TRY(ctx, tmp.alloc(tb));
        R_TRY(ctx, hipcub::DeviceScan::InclusiveSum(tmp.p, tb, head.as<uint32_t>(), slot.as<uint32_t>(), (int)n, st));
        R_TRY(ctx, hipStreamSynchronize(st));
    }
    uint32_t n_out = 0;
    R_TRY(ctx, hipMemcpy(&n_out, slot.as<uint32_t>() + n - 1, 4, hipMemcpyDeviceToHost));
    k_merge_emit<<<nblk(n), 256, 0, st>>>(ds.as<int64_t>(), perm, run.as<GroupMax>(), head.as<uint32_t>(), slot.as<uint32_t>(), n,
                                          og.as<uint32_t>(), os.as<int64_t>(), oe.as<int64_t>());
    R_TRY(ctx, hipMemcpyAsync(out_group, og.p, 4ull * n_out, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_start, os.p, 8ull * n_out, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_end, oe.p, 8ull * n_out, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    *out_n = n_out;
    return SNPGPU_OK;
}

int snpgpu_in_regions(snpgpu_ctx *ctx, const uint32_t *pos_group, const int64_t *positions, uint32_t n_pos,
                      const uint32_t *reg_off, const int64_t *reg_start, const int64_t *reg_end,
                      uint32_t n_groups, uint8_t *out_flag) {
    if (!ctx) return SNPGPU_E_ARG;
    if (n_pos == 0) return SNPGPU_OK;
    if (!pos_group || !positions || !out_flag || (n_groups && !reg_off)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null in_regions argument");
    R_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint32_t n_reg = n_groups ? reg_off[n_groups] : 0;
    DevBuf dg, dp, doff, drs, dre, dout;
    R_TRY(ctx, dg.alloc(4ull * n_pos)); R_TRY(ctx, dp.alloc(8ull * n_pos)); R_TRY(ctx, doff.alloc(4ull * (n_groups + 1)));
    R_TRY(ctx, drs.alloc(8ull * n_reg)); R_TRY(ctx, dre.alloc(8ull * n_reg)); R_TRY(ctx, dout.alloc(n_pos));
    R_TRY(ctx, hipMemcpyAsync(dg.p, pos_group, 4ull * n_pos, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(dp.p, positions, 8ull * n_pos, hipMemcpyHostToDevice, st));
    if (n_groups) R_TRY(ctx, hipMemcpyAsync(doff.p, reg_off, 4ull * (n_groups + 1), hipMemcpyHostToDevice, st));
    if (n_reg) {
        R_TRY(ctx, hipMemcpyAsync(drs.p, reg_start, 8ull * n_reg, hipMemcpyHostToDevice, st));
        R_TRY(ctx, hipMemcpyAsync(dre.p, reg_end, 8ull * n_reg, hipMemcpyHostToDevice, st));
    }
    k_in_regions<<<nblk(n_pos), 256, 0, st>>>(dg.as<uint32_t>(), dp.as<int64_t>(), n_pos, doff.as<uint32_t>(), drs.as<int64_t>(), dre.as<int64_t>(), n_groups, dout.as<uint8_t>());
    R_TRY(ctx, hipMemcpyAsync(out_flag, dout.p, n_pos, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    return SNPGPU_OK;
}

int snpgpu_merge_sites(snpgpu_ctx *ctx, const uint64_t *keys, const uint32_t *sample_of_key, size_t n,
                       uint64_t *out_unique, uint32_t *out_off, uint32_t *out_carrier,
                       uint32_t *out_n_unique, uint32_t *out_n_carrier) {
    if (!ctx || !out_n_unique || !out_n_carrier) return SNPGPU_E_ARG;
    *out_n_unique = *out_n_carrier = 0;
    if (out_off) out_off[0] = 0;
    if (n == 0) return SNPGPU_OK;
    if (n > 0x7FFFFFFFull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many site records");
    if (!keys || !sample_of_key || !out_unique || !out_off || !out_carrier) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null merge_sites argument");
    R_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint32_t m = (uint32_t)n;
    DevBuf k0, k1, v0, v1, nk, np, ks, ps, ou, oo, oc;
    R_TRY(ctx, k0.alloc(8ull * m)); R_TRY(ctx, k1.alloc(8ull * m)); R_TRY(ctx, v0.alloc(4ull * m)); R_TRY(ctx, v1.alloc(4ull * m));
    R_TRY(ctx, nk.alloc(4ull * m)); R_TRY(ctx, np.alloc(4ull * m)); R_TRY(ctx, ks.alloc(4ull * m)); R_TRY(ctx, ps.alloc(4ull * m));
    R_TRY(ctx, ou.alloc(8ull * m)); R_TRY(ctx, oo.alloc(4ull * (m + 1))); R_TRY(ctx, oc.alloc(4ull * m));
    R_TRY(ctx, hipMemcpyAsync(k0.p, keys, 8ull * m, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(v0.p, sample_of_key, 4ull * m, hipMemcpyHostToDevice, st));
    // stable sort by key keeps each key's carriers in input (= sorted sample) order
    R_TRY(ctx, sort_pairs_u64_u32(k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), m, st));
    k_sites_flags<<<nblk(m), 256, 0, st>>>(k1.as<uint64_t>(), v1.as<uint32_t>(), m, nk.as<uint32_t>(), np.as<uint32_t>());
    R_TRY(ctx, hipMemcpyAsync(ks.p, nk.p, 4ull * m, hipMemcpyDeviceToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(ps.p, np.p, 4ull * m, hipMemcpyDeviceToDevice, st));
    R_TRY(ctx, exclusive_sum(ks.as<uint32_t>(), m, st));
    R_TRY(ctx, exclusive_sum(ps.as<uint32_t>(), m, st));
    uint32_t lk = 0, lp = 0, fk = 0, fp = 0;
    R_TRY(ctx, hipMemcpy(&lk, ks.as<uint32_t>() + m - 1, 4, hipMemcpyDeviceToHost));
    R_TRY(ctx, hipMemcpy(&lp, ps.as<uint32_t>() + m - 1, 4, hipMemcpyDeviceToHost));
    R_TRY(ctx, hipMemcpy(&fk, nk.as<uint32_t>() + m - 1, 4, hipMemcpyDeviceToHost));
    R_TRY(ctx, hipMemcpy(&fp, np.as<uint32_t>() + m - 1, 4, hipMemcpyDeviceToHost));
    uint32_t n_unique = lk + fk, n_pairs = lp + fp;
    k_sites_emit<<<nblk(m), 256, 0, st>>>(k1.as<uint64_t>(), v1.as<uint32_t>(), m, nk.as<uint32_t>(), np.as<uint32_t>(), ks.as<uint32_t>(), ps.as<uint32_t>(),
                                          ou.as<uint64_t>(), oo.as<uint32_t>(), oc.as<uint32_t>());
    R_TRY(ctx, hipMemcpyAsync(out_unique, ou.p, 8ull * n_unique, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_off, oo.p, 4ull * n_unique, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_carrier, oc.p, 4ull * n_pairs, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    out_off[n_unique] = n_pairs;
    *out_n_unique = n_unique;
    *out_n_carrier = n_pairs;
    return SNPGPU_OK;
}

}  // extern "C"
