// K3 dense-window filter and region algebra, K4 site merge — hand-written for gfx950 on the primitives of prims.h
// (DPP prefix sums, LDS bitonic tile sort + merge-path passes, three-launch scans).  No library sorts or scans.
//
//   find_dense_regions      snppipeline/filter_regions.py:17-71   -> sort (segment, position) keys, one thread per
//                                                                    (position, rule) window test, scan compaction
//   utils.merge_regions     snppipeline/utils.py:1267-1282        -> sort (group, start, end), segmented running max of the
//                                                                    ends (scan with a (group, max) operator), heads, compaction
//   utils.in_region         snppipeline/utils.py:1314-1318        -> binary search in the merged list
//   merge_sites union       snppipeline/merge_sites.py:91-117     -> sort (contig << 32 | pos, sample), unique keys + carrier CSR
//
// These steps are small next to the pileup scan (KBs to MBs) and latency-bound: what matters is that nothing between the
// first and the last launch of a step waits for the host.  Every step therefore exists as a `_dev` entry point (device
// pointers in and out, counts left in device memory, asynchronous on the context's stream: the sharded pipeline feeds
// them straight from the all-gathered tensors) and as a host-pointer wrapper that stages inputs and outputs through the
// context's scratch and synchronises once, at the end.  VCF records usually arrive sorted: the sort notices and copies.
#include <string.h>

#include <vector>

#include "internal.h"
#include "prims.h"

namespace {

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }

#define R_TRY(ctx, expr)                                                                                \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return snpgpu_set_error((ctx), SNPGPU_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// error bits a step leaves in out_n[1]
#define REG_ERR_POSITION 1u      // SNP position outside [0, 2^40)
#define REG_ERR_INTERVAL 2u      // interval with start > end

struct U64Less { __device__ bool operator()(const uint64_t &a, const uint64_t &b) const { return a < b; } };

// ------------------------------------------------------------------------------------------------ dense windows
#define MAX_RULES 64
struct Rules { int32_t max_snps[MAX_RULES], window[MAX_RULES]; uint32_t n; };

__device__ __forceinline__ uint32_t seg_of(const uint32_t *seg_off, uint32_t n_segs, uint32_t i) {   // last s with seg_off[s] <= i
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// (segment << 40) | position: sorting these sorts every segment's positions (filter_regions.py:425 sorted())
__global__ void k_seg_keys(const int64_t *pos, const uint32_t *seg_off, uint32_t n_segs, uint32_t n_pos, uint64_t *keys, uint32_t *out_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    int64_t p = pos[i];
    if (p < 0 || p >= (1ll << 40)) { atomicOr(&out_n[1], REG_ERR_POSITION); p = 0; }
    keys[i] = ((uint64_t)seg_of(seg_off, n_segs, i) << 40) | (uint64_t)p;
}

// flag of (position i, rule r): the window that starts at p[i] holds more than max_snps[r] SNPs (filter_regions.py:63-68)
__global__ void k_dense_flags(const uint64_t *keys, const uint32_t *seg_off, uint32_t n_segs, uint32_t n_pos, Rules rules, uint32_t *flag) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_pos * rules.n) return;
    const uint32_t i = (uint32_t)(g / rules.n), r = (uint32_t)(g % rules.n);
    const uint32_t end = seg_off[seg_of(seg_off, n_segs, i) + 1];
    const int64_t m = rules.max_snps[r], mask = (1ll << 40) - 1;
    uint32_t f = 0;
    if (m >= 0 && (uint64_t)i + (uint64_t)m < end) f = ((int64_t)(keys[i] & mask) + (int64_t)rules.window[r] - 1 >= (int64_t)(keys[i + m] & mask)) ? 1u : 0u;
    flag[g] = f;
}

__global__ void k_dense_emit(const uint64_t *keys, uint32_t n_pos, Rules rules, const uint32_t *flag, const uint32_t *slot,
                             int64_t *out_start, int64_t *out_end, uint32_t *out_seg) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)n_pos * rules.n || !flag[g]) return;
    const uint32_t i = (uint32_t)(g / rules.n), r = (uint32_t)(g % rules.n);
    const uint64_t mask = (1ull << 40) - 1;
    const uint32_t o = slot[g];
    out_start[o] = (int64_t)(keys[i] & mask);
    out_end[o] = (int64_t)(keys[i + rules.max_snps[r]] & mask);
    out_seg[o] = (uint32_t)(keys[i] >> 40);
}

__global__ void k_copy_word(const uint32_t *src, uint32_t *dst) { *dst = *src; }

size_t dense_ws_bytes(uint32_t n_pos, uint32_t n_rules) {
    const uint64_t total = (uint64_t)n_pos * n_rules;
    return 3 * up256(8ull * n_pos) + 256 + 2 * up256(4 * total) + up256(4 * prim_scan_workspace_words(total)) + 256;
}

int dense_enqueue(snpgpu_ctx *ctx, const int64_t *d_pos, const uint32_t *d_seg_off, uint32_t n_segs, uint32_t n_pos, const Rules &rules,
                  int64_t *d_out_start, int64_t *d_out_end, uint32_t *d_out_seg, uint32_t *d_out_n, char *ws) {
    hipStream_t st = ctx->stream;
    const uint64_t total = (uint64_t)n_pos * rules.n;
    size_t o = 0;
    uint64_t *keys = (uint64_t *)(ws + o); o += up256(8ull * n_pos);
    uint64_t *ka = (uint64_t *)(ws + o); o += up256(8ull * n_pos);
    uint64_t *kb = (uint64_t *)(ws + o); o += up256(8ull * n_pos);
    uint32_t *sflag = (uint32_t *)(ws + o); o += 256;
    uint32_t *flag = (uint32_t *)(ws + o); o += up256(4 * total);
    uint32_t *slot = (uint32_t *)(ws + o); o += up256(4 * total);
    uint32_t *scan_ws = (uint32_t *)(ws + o);
    R_TRY(ctx, hipMemsetAsync(d_out_n, 0, 8, st));            // [0] windows found, [1] error bits
    if (n_pos && rules.n) {
        k_seg_keys<<<nblk(n_pos), 256, 0, st>>>(d_pos, d_seg_off, n_segs, n_pos, keys, d_out_n);
        const uint64_t *sorted = prim_sort<uint64_t, U64Less>(st, keys, ka, kb, n_pos, sflag, U64Less());
        k_dense_flags<<<nblk(total), 256, 0, st>>>(sorted, d_seg_off, n_segs, n_pos, rules, flag);
        uint32_t *d_total = nullptr;
        prim_exclusive_scan_u32(st, flag, slot, total, scan_ws, &d_total);
        k_dense_emit<<<nblk(total), 256, 0, st>>>(sorted, n_pos, rules, flag, slot, d_out_start, d_out_end, d_out_seg);
        k_copy_word<<<1, 1, 0, st>>>(d_total, d_out_n);
    }
    R_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int make_rules(snpgpu_ctx *ctx, const int32_t *max_snps, const int32_t *window, uint32_t n_rules, Rules &rules) {
    if (n_rules > MAX_RULES) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "more than %d dense-window rules", MAX_RULES);
    if (n_rules && (!max_snps || !window)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null dense-window argument");
    memset(&rules, 0, sizeof rules);
    rules.n = n_rules;
    for (uint32_t r = 0; r < n_rules; ++r) { rules.max_snps[r] = max_snps[r]; rules.window[r] = window[r]; }
    return SNPGPU_OK;
}

// ------------------------------------------------------------------------------------------------ merge_regions
struct Ival { uint32_t group, pad; int64_t start, end; };
struct IvalLess {
    __device__ bool operator()(const Ival &a, const Ival &b) const {
        if (a.group != b.group) return a.group < b.group;
        if (a.start != b.start) return a.start < b.start;
        return a.end < b.end;
    }
};
struct GroupMax { int64_t maxend; uint32_t group, pad; };
struct GroupMaxOp {                                           // running max of the ends, restarted where the group changes
    __device__ GroupMax operator()(const GroupMax &a, const GroupMax &b) const {
        GroupMax r;
        r.group = b.group; r.pad = 0;
        r.maxend = (a.group == b.group && a.maxend > b.maxend) ? a.maxend : b.maxend;
        return r;
    }
};

__global__ void k_ival_pack(const uint32_t *group, const int64_t *start, const int64_t *end, uint32_t n, Ival *out, uint32_t *out_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ival v;
    v.group = group[i]; v.pad = 0; v.start = start[i]; v.end = end[i];
    if (v.start > v.end) atomicOr(&out_n[1], REG_ERR_INTERVAL);
    out[i] = v;
}

__global__ void k_ival_ends(const Ival *sorted, uint32_t n, GroupMax *gm) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { GroupMax g; g.group = sorted[i].group; g.pad = 0; g.maxend = sorted[i].end; gm[i] = g; }
}

// head[i] = 1 when sorted interval i opens a new merged region: other group, or start > (largest end so far) + 1
__global__ void k_ival_heads(const Ival *sorted, const GroupMax *run, uint32_t n, uint32_t *head) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = 1;
    if (i > 0 && run[i - 1].group == sorted[i].group) {
        const int64_t le = run[i - 1].maxend;
        h = (le == INT64_MAX || sorted[i].start <= le + 1) ? 0u : 1u;
    }
    head[i] = h;
}

__global__ void k_ival_emit(const Ival *sorted, const GroupMax *run, const uint32_t *head, const uint32_t *slot, uint32_t n,
                            uint32_t *out_group, int64_t *out_start, int64_t *out_end) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = slot[i] + head[i] - 1;                 // exclusive scan of heads -> index of the region i belongs to
    if (head[i]) { out_group[o] = sorted[i].group; out_start[o] = sorted[i].start; }
    if (i + 1 == n || head[i + 1]) out_end[o] = run[i].maxend;
}

size_t merge_regions_ws_bytes(uint32_t n) {
    return 3 * up256(sizeof(Ival) * (size_t)n) + 256 + 2 * up256(sizeof(GroupMax) * (size_t)n) + up256(sizeof(GroupMax) * (size_t)prim_gscan_blocks(n)) +
           2 * up256(4ull * n) + up256(4 * prim_scan_workspace_words(n)) + 256;
}

int merge_regions_enqueue(snpgpu_ctx *ctx, const uint32_t *d_group, const int64_t *d_start, const int64_t *d_end, uint32_t n,
                          uint32_t *d_out_group, int64_t *d_out_start, int64_t *d_out_end, uint32_t *d_out_n, char *ws) {
    hipStream_t st = ctx->stream;
    size_t o = 0;
    Ival *iv = (Ival *)(ws + o); o += up256(sizeof(Ival) * (size_t)n);
    Ival *ia = (Ival *)(ws + o); o += up256(sizeof(Ival) * (size_t)n);
    Ival *ib = (Ival *)(ws + o); o += up256(sizeof(Ival) * (size_t)n);
    uint32_t *sflag = (uint32_t *)(ws + o); o += 256;
    GroupMax *gm = (GroupMax *)(ws + o); o += up256(sizeof(GroupMax) * (size_t)n);
    GroupMax *run = (GroupMax *)(ws + o); o += up256(sizeof(GroupMax) * (size_t)n);
    GroupMax *aggr = (GroupMax *)(ws + o); o += up256(sizeof(GroupMax) * (size_t)prim_gscan_blocks(n));
    uint32_t *head = (uint32_t *)(ws + o); o += up256(4ull * n);
    uint32_t *slot = (uint32_t *)(ws + o); o += up256(4ull * n);
    uint32_t *scan_ws = (uint32_t *)(ws + o);
    R_TRY(ctx, hipMemsetAsync(d_out_n, 0, 8, st));            // [0] merged regions, [1] error bits
    if (n) {
        k_ival_pack<<<nblk(n), 256, 0, st>>>(d_group, d_start, d_end, n, iv, d_out_n);
        const Ival *sorted = prim_sort<Ival, IvalLess>(st, iv, ia, ib, n, sflag, IvalLess());
        k_ival_ends<<<nblk(n), 256, 0, st>>>(sorted, n, gm);
        prim_inclusive_scan<GroupMax, GroupMaxOp>(st, gm, run, n, aggr, GroupMaxOp());
        k_ival_heads<<<nblk(n), 256, 0, st>>>(sorted, run, n, head);
        uint32_t *d_total = nullptr;
        prim_exclusive_scan_u32(st, head, slot, n, scan_ws, &d_total);
        k_ival_emit<<<nblk(n), 256, 0, st>>>(sorted, run, head, slot, n, d_out_group, d_out_start, d_out_end);
        k_copy_word<<<1, 1, 0, st>>>(d_total, d_out_n);
    }
    R_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// ------------------------------------------------------------------------------------------------ in_regions
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

// ------------------------------------------------------------------------------------------------ merge_sites
struct Pair { uint64_t key; uint32_t samp, pad; };
struct PairLess {
    __device__ bool operator()(const Pair &a, const Pair &b) const { return a.key != b.key ? a.key < b.key : a.samp < b.samp; }
};

__global__ void k_pair_pack(const uint64_t *keys, const uint32_t *samp, uint32_t n, Pair *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { Pair p; p.key = keys[i]; p.samp = samp[i]; p.pad = 0; out[i] = p; }
}

__global__ void k_sites_flags(const Pair *sorted, uint32_t n, uint32_t *new_key, uint32_t *new_pair) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool nk = i == 0 || sorted[i].key != sorted[i - 1].key;
    new_key[i] = nk;
    new_pair[i] = nk || sorted[i].samp != sorted[i - 1].samp;  // the reference builds a set per sample: (key, sample) once
}

__global__ void k_sites_emit(const Pair *sorted, uint32_t n, const uint32_t *new_key, const uint32_t *new_pair, const uint32_t *key_slot,
                             const uint32_t *pair_slot, const uint32_t *n_keys, const uint32_t *n_pairs, uint64_t *out_unique,
                             uint32_t *out_off, uint32_t *out_carrier, uint32_t *out_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { out_n[0] = *n_keys; out_n[1] = *n_pairs; out_off[*n_keys] = *n_pairs; }
    if (i >= n) return;
    if (new_pair[i]) out_carrier[pair_slot[i]] = sorted[i].samp;
    if (new_key[i]) { out_unique[key_slot[i]] = sorted[i].key; out_off[key_slot[i]] = pair_slot[i]; }
}

size_t merge_sites_ws_bytes(uint32_t n) {
    return 3 * up256(sizeof(Pair) * (size_t)n) + 256 + 4 * up256(4ull * n) + 2 * up256(4 * prim_scan_workspace_words(n)) + 256;
}

int merge_sites_enqueue(snpgpu_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_samp, uint32_t n, uint64_t *d_out_unique,
                        uint32_t *d_out_off, uint32_t *d_out_carrier, uint32_t *d_out_n, char *ws) {
    hipStream_t st = ctx->stream;
    size_t o = 0;
    Pair *pr = (Pair *)(ws + o); o += up256(sizeof(Pair) * (size_t)n);
    Pair *pa = (Pair *)(ws + o); o += up256(sizeof(Pair) * (size_t)n);
    Pair *pb = (Pair *)(ws + o); o += up256(sizeof(Pair) * (size_t)n);
    uint32_t *sflag = (uint32_t *)(ws + o); o += 256;
    uint32_t *nk = (uint32_t *)(ws + o); o += up256(4ull * n);
    uint32_t *np = (uint32_t *)(ws + o); o += up256(4ull * n);
    uint32_t *ks = (uint32_t *)(ws + o); o += up256(4ull * n);
    uint32_t *ps = (uint32_t *)(ws + o); o += up256(4ull * n);
    uint32_t *scan1 = (uint32_t *)(ws + o); o += up256(4 * prim_scan_workspace_words(n));
    uint32_t *scan2 = (uint32_t *)(ws + o);
    R_TRY(ctx, hipMemsetAsync(d_out_n, 0, 8, st));            // [0] unique keys, [1] (key, sample) pairs
    R_TRY(ctx, hipMemsetAsync(d_out_off, 0, 4, st));
    if (n) {
        k_pair_pack<<<nblk(n), 256, 0, st>>>(d_keys, d_samp, n, pr);
        const Pair *sorted = prim_sort<Pair, PairLess>(st, pr, pa, pb, n, sflag, PairLess());
        k_sites_flags<<<nblk(n), 256, 0, st>>>(sorted, n, nk, np);
        uint32_t *t_keys = nullptr, *t_pairs = nullptr;
        prim_exclusive_scan_u32(st, nk, ks, n, scan1, &t_keys);
        prim_exclusive_scan_u32(st, np, ps, n, scan2, &t_pairs);
        k_sites_emit<<<nblk(n), 256, 0, st>>>(sorted, n, nk, np, ks, ps, t_keys, t_pairs, d_out_unique, d_out_off, d_out_carrier, d_out_n);
    }
    R_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int region_error(snpgpu_ctx *ctx, uint32_t bits) {
    if (bits & REG_ERR_POSITION) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "SNP position outside [0, 2^40)");
    if (bits & REG_ERR_INTERVAL) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "interval with start > end");
    return SNPGPU_OK;
}

}  // namespace

extern "C" {

// ---- device-pointer forms: asynchronous on the context's stream, counts left in device memory ------------------------
int snpgpu_dense_windows_dev(snpgpu_ctx *ctx, const int64_t *d_positions, const uint32_t *d_seg_off, uint32_t n_segs, uint32_t n_pos,
                             const int32_t *max_snps, const int32_t *window, uint32_t n_rules, int64_t *d_out_start,
                             int64_t *d_out_end, uint32_t *d_out_seg, uint32_t *d_out_n) {
    if (!ctx || !d_out_n) return SNPGPU_E_ARG;
    Rules rules;
    int rc = make_rules(ctx, max_snps, window, n_rules, rules);
    if (rc) return rc;
    if ((uint64_t)n_pos * n_rules > 0x7FFFFFFFull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many (position, rule) pairs");
    if (n_segs >= (1u << 24)) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many segments");
    if (n_pos && (!d_positions || !d_seg_off || !d_out_start || !d_out_end || !d_out_seg)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null dense-window argument");
    R_TRY(ctx, snpgpu_enter(ctx));
    void *ws = nullptr;
    rc = snpgpu_scratch(ctx, dense_ws_bytes(n_pos, n_rules), &ws);
    if (rc) return rc;
    return dense_enqueue(ctx, d_positions, d_seg_off, n_segs, n_pos, rules, d_out_start, d_out_end, d_out_seg, d_out_n, (char *)ws);
}

int snpgpu_merge_regions_dev(snpgpu_ctx *ctx, const uint32_t *d_group, const int64_t *d_start, const int64_t *d_end, uint32_t n,
                             uint32_t *d_out_group, int64_t *d_out_start, int64_t *d_out_end, uint32_t *d_out_n) {
    if (!ctx || !d_out_n) return SNPGPU_E_ARG;
    if (n && (!d_group || !d_start || !d_end || !d_out_group || !d_out_start || !d_out_end)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null merge argument");
    R_TRY(ctx, snpgpu_enter(ctx));
    void *ws = nullptr;
    int rc = snpgpu_scratch(ctx, merge_regions_ws_bytes(n), &ws);
    if (rc) return rc;
    return merge_regions_enqueue(ctx, d_group, d_start, d_end, n, d_out_group, d_out_start, d_out_end, d_out_n, (char *)ws);
}

int snpgpu_in_regions_dev(snpgpu_ctx *ctx, const uint32_t *d_pos_group, const int64_t *d_positions, uint32_t n_pos,
                          const uint32_t *d_reg_off, const int64_t *d_reg_start, const int64_t *d_reg_end, uint32_t n_groups,
                          uint8_t *d_out_flag) {
    if (!ctx) return SNPGPU_E_ARG;
    if (n_pos == 0) return SNPGPU_OK;
    if (!d_pos_group || !d_positions || !d_out_flag || (n_groups && !d_reg_off)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null in_regions argument");
    R_TRY(ctx, snpgpu_enter(ctx));
    k_in_regions<<<nblk(n_pos), 256, 0, ctx->stream>>>(d_pos_group, d_positions, n_pos, d_reg_off, d_reg_start, d_reg_end, n_groups, d_out_flag);
    R_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_merge_sites_dev(snpgpu_ctx *ctx, const uint64_t *d_keys, const uint32_t *d_sample_of_key, uint32_t n,
                           uint64_t *d_out_unique, uint32_t *d_out_off, uint32_t *d_out_carrier, uint32_t *d_out_n) {
    if (!ctx || !d_out_n || !d_out_off) return SNPGPU_E_ARG;
    if (n > 0x7FFFFFFFu) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many site records");
    if (n && (!d_keys || !d_sample_of_key || !d_out_unique || !d_out_carrier)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null merge_sites argument");
    R_TRY(ctx, snpgpu_enter(ctx));
    void *ws = nullptr;
    int rc = snpgpu_scratch(ctx, merge_sites_ws_bytes(n), &ws);
    if (rc) return rc;
    return merge_sites_enqueue(ctx, d_keys, d_sample_of_key, n, d_out_unique, d_out_off, d_out_carrier, d_out_n, (char *)ws);
}

// ---- host-pointer forms: inputs and outputs staged through the context's scratch, one synchronisation at the end ------
int snpgpu_dense_windows(snpgpu_ctx *ctx, const int64_t *positions, const uint32_t *seg_off, uint32_t n_segs,
                         const int32_t *max_snps, const int32_t *window, uint32_t n_rules,
                         int64_t *out_start, int64_t *out_end, uint32_t *out_seg, uint32_t *out_n) {
    if (!ctx || !out_n) return SNPGPU_E_ARG;
    *out_n = 0;
    if (n_segs == 0 || n_rules == 0) return SNPGPU_OK;
    if (!seg_off) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null dense-window argument");
    const uint32_t n_pos = seg_off[n_segs];
    if (n_pos == 0) return SNPGPU_OK;
    if (!positions || !out_start || !out_end || !out_seg) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null dense-window argument");
    Rules rules;
    int rc = make_rules(ctx, max_snps, window, n_rules, rules);
    if (rc) return rc;
    const uint64_t total = (uint64_t)n_pos * n_rules;
    if (total > 0x7FFFFFFFull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many (position, rule) pairs");
    if (n_segs >= (1u << 24)) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many segments");
    R_TRY(ctx, snpgpu_enter(ctx));
    hipStream_t st = ctx->stream;
    size_t o = 0;
    const size_t o_pos = o; o += up256(8ull * n_pos);
    const size_t o_seg = o; o += up256(4ull * (n_segs + 1));
    const size_t o_os = o; o += up256(8 * total);
    const size_t o_oe = o; o += up256(8 * total);
    const size_t o_og = o; o += up256(4 * total);
    const size_t o_n = o; o += 256;
    const size_t o_ws = o; o += dense_ws_bytes(n_pos, n_rules);
    void *scr = nullptr;
    rc = snpgpu_scratch(ctx, o, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    R_TRY(ctx, hipMemcpyAsync(b + o_pos, positions, 8ull * n_pos, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(b + o_seg, seg_off, 4ull * (n_segs + 1), hipMemcpyHostToDevice, st));
    rc = dense_enqueue(ctx, (const int64_t *)(b + o_pos), (const uint32_t *)(b + o_seg), n_segs, n_pos, rules, (int64_t *)(b + o_os),
                       (int64_t *)(b + o_oe), (uint32_t *)(b + o_og), (uint32_t *)(b + o_n), b + o_ws);
    if (rc) return rc;
    uint32_t res[2] = {0, 0};
    R_TRY(ctx, hipMemcpyAsync(res, b + o_n, 8, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    if (res[1]) return region_error(ctx, res[1]);
    if (res[0]) {
        R_TRY(ctx, hipMemcpyAsync(out_start, b + o_os, 8ull * res[0], hipMemcpyDeviceToHost, st));
        R_TRY(ctx, hipMemcpyAsync(out_end, b + o_oe, 8ull * res[0], hipMemcpyDeviceToHost, st));
        R_TRY(ctx, hipMemcpyAsync(out_seg, b + o_og, 4ull * res[0], hipMemcpyDeviceToHost, st));
        R_TRY(ctx, hipStreamSynchronize(st));
    }
    *out_n = res[0];
    return SNPGPU_OK;
}

int snpgpu_merge_regions(snpgpu_ctx *ctx, const uint32_t *group, const int64_t *start, const int64_t *end,
                         uint32_t n, uint32_t *out_group, int64_t *out_start, int64_t *out_end, uint32_t *out_n) {
    if (!ctx || !out_n) return SNPGPU_E_ARG;
    *out_n = 0;
    if (n == 0) return SNPGPU_OK;
    if (!group || !start || !end || !out_group || !out_start || !out_end) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null merge argument");
    R_TRY(ctx, snpgpu_enter(ctx));
    hipStream_t st = ctx->stream;
    size_t o = 0;
    const size_t o_g = o; o += up256(4ull * n);
    const size_t o_s = o; o += up256(8ull * n);
    const size_t o_e = o; o += up256(8ull * n);
    const size_t o_og = o; o += up256(4ull * n);
    const size_t o_os = o; o += up256(8ull * n);
    const size_t o_oe = o; o += up256(8ull * n);
    const size_t o_n = o; o += 256;
    const size_t o_ws = o; o += merge_regions_ws_bytes(n);
    void *scr = nullptr;
    int rc = snpgpu_scratch(ctx, o, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    R_TRY(ctx, hipMemcpyAsync(b + o_g, group, 4ull * n, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(b + o_s, start, 8ull * n, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(b + o_e, end, 8ull * n, hipMemcpyHostToDevice, st));
    rc = merge_regions_enqueue(ctx, (const uint32_t *)(b + o_g), (const int64_t *)(b + o_s), (const int64_t *)(b + o_e), n,
                               (uint32_t *)(b + o_og), (int64_t *)(b + o_os), (int64_t *)(b + o_oe), (uint32_t *)(b + o_n), b + o_ws);
    if (rc) return rc;
    uint32_t res[2] = {0, 0};
    R_TRY(ctx, hipMemcpyAsync(res, b + o_n, 8, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    if (res[1]) return region_error(ctx, res[1]);
    R_TRY(ctx, hipMemcpyAsync(out_group, b + o_og, 4ull * res[0], hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_start, b + o_os, 8ull * res[0], hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_end, b + o_oe, 8ull * res[0], hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    *out_n = res[0];
    return SNPGPU_OK;
}

int snpgpu_in_regions(snpgpu_ctx *ctx, const uint32_t *pos_group, const int64_t *positions, uint32_t n_pos,
                      const uint32_t *reg_off, const int64_t *reg_start, const int64_t *reg_end,
                      uint32_t n_groups, uint8_t *out_flag) {
    if (!ctx) return SNPGPU_E_ARG;
    if (n_pos == 0) return SNPGPU_OK;
    if (!pos_group || !positions || !out_flag || (n_groups && !reg_off)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null in_regions argument");
    R_TRY(ctx, snpgpu_enter(ctx));
    hipStream_t st = ctx->stream;
    const uint32_t n_reg = n_groups ? reg_off[n_groups] : 0;
    size_t o = 0;
    const size_t o_g = o; o += up256(4ull * n_pos);
    const size_t o_p = o; o += up256(8ull * n_pos);
    const size_t o_off = o; o += up256(4ull * (n_groups + 1));
    const size_t o_rs = o; o += up256(8ull * n_reg);
    const size_t o_re = o; o += up256(8ull * n_reg);
    const size_t o_out = o; o += up256(n_pos);
    void *scr = nullptr;
    int rc = snpgpu_scratch(ctx, o + 256, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    R_TRY(ctx, hipMemcpyAsync(b + o_g, pos_group, 4ull * n_pos, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(b + o_p, positions, 8ull * n_pos, hipMemcpyHostToDevice, st));
    if (n_groups) R_TRY(ctx, hipMemcpyAsync(b + o_off, reg_off, 4ull * (n_groups + 1), hipMemcpyHostToDevice, st));
    if (n_reg) {
        R_TRY(ctx, hipMemcpyAsync(b + o_rs, reg_start, 8ull * n_reg, hipMemcpyHostToDevice, st));
        R_TRY(ctx, hipMemcpyAsync(b + o_re, reg_end, 8ull * n_reg, hipMemcpyHostToDevice, st));
    }
    k_in_regions<<<nblk(n_pos), 256, 0, st>>>((const uint32_t *)(b + o_g), (const int64_t *)(b + o_p), n_pos, (const uint32_t *)(b + o_off),
                                              (const int64_t *)(b + o_rs), (const int64_t *)(b + o_re), n_groups, (uint8_t *)(b + o_out));
    R_TRY(ctx, hipMemcpyAsync(out_flag, b + o_out, n_pos, hipMemcpyDeviceToHost, st));
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
    R_TRY(ctx, snpgpu_enter(ctx));
    hipStream_t st = ctx->stream;
    const uint32_t m = (uint32_t)n;
    size_t o = 0;
    const size_t o_k = o; o += up256(8ull * m);
    const size_t o_s = o; o += up256(4ull * m);
    const size_t o_u = o; o += up256(8ull * m);
    const size_t o_off = o; o += up256(4ull * (m + 1));
    const size_t o_c = o; o += up256(4ull * m);
    const size_t o_n = o; o += 256;
    const size_t o_ws = o; o += merge_sites_ws_bytes(m);
    void *scr = nullptr;
    int rc = snpgpu_scratch(ctx, o, &scr);
    if (rc) return rc;
    char *b = (char *)scr;
    R_TRY(ctx, hipMemcpyAsync(b + o_k, keys, 8ull * m, hipMemcpyHostToDevice, st));
    R_TRY(ctx, hipMemcpyAsync(b + o_s, sample_of_key, 4ull * m, hipMemcpyHostToDevice, st));
    rc = merge_sites_enqueue(ctx, (const uint64_t *)(b + o_k), (const uint32_t *)(b + o_s), m, (uint64_t *)(b + o_u), (uint32_t *)(b + o_off),
                             (uint32_t *)(b + o_c), (uint32_t *)(b + o_n), b + o_ws);
    if (rc) return rc;
    uint32_t res[2] = {0, 0};
    R_TRY(ctx, hipMemcpyAsync(res, b + o_n, 8, hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    R_TRY(ctx, hipMemcpyAsync(out_unique, b + o_u, 8ull * res[0], hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_off, b + o_off, 4ull * (res[0] + 1), hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipMemcpyAsync(out_carrier, b + o_c, 4ull * res[1], hipMemcpyDeviceToHost, st));
    R_TRY(ctx, hipStreamSynchronize(st));
    *out_n_unique = res[0];
    *out_n_carrier = res[1];
    return SNPGPU_OK;
}

}  // extern "C"
