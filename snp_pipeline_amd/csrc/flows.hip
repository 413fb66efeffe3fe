// Both consensus flows of the pipeline from ONE scan + call.
//
// The reference calls every sample twice (run.py:704-710 and :712-718): once at the positions of snplist.txt, and once at
// those of snplist_preserved.txt with the sample's var.flt_removed.vcf as the exclude file (call_consensus.py:117-123).
// The counts of a position do not depend on the flow — what differs is the column subset (the preserved list is a subset of
// the full list: preserved records are records) and the "Region" filter, which call_consensus.py:165-168 appends to the
// failed filters of every parsed position that is in the sample's exclude set (and so turns its base into '-', :169-176).
// So the preserved flow is derived on the device from the result of the full-list call:
//   base_p[s][j]    = base[s][cols[j]]                         (cols: slots of the preserved list, in its order)
//   filters_p[s][i] = filters[s][i]
//   for every (sample s, slot i) of the exclude lists whose position has a well-formed pileup line:
//       filters_p[s][i] |= Region;  base_p[s][col_of[i]] = '-'  when the slot is a column of the preserved list
#include "internal.h"

namespace {

__global__ __launch_bounds__(256) void k_flow_gather(const uint8_t *__restrict__ base, uint32_t n_samples, uint32_t n_sites,
                                                     const uint32_t *__restrict__ cols, uint32_t n_cols, uint8_t *__restrict__ out) {
    const uint64_t total = (uint64_t)n_samples * n_cols;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(i / n_cols), j = (uint32_t)(i - (uint64_t)s * n_cols);
        out[i] = base[(uint64_t)s * n_sites + cols[j]];
    }
}

__global__ __launch_bounds__(256) void k_flow_exclude(const uint64_t *__restrict__ line_off, const uint32_t *__restrict__ excl_off,
                                                      const uint32_t *__restrict__ excl_slots, uint32_t n_samples, uint32_t n_sites,
                                                      const int32_t *__restrict__ col_of, uint32_t n_cols, uint8_t *base_p, uint8_t *filters_p,
                                                      uint32_t *err) {
    // one workgroup per sample: its exclude list is short (the removed records of one VCF)
    for (uint32_t s = blockIdx.x; s < n_samples; s += gridDim.x) {
        for (uint32_t k = excl_off[s] + threadIdx.x; k < excl_off[s + 1]; k += blockDim.x) {
            const uint32_t slot = excl_slots[k];
            if (slot >= n_sites) { atomicOr(err, 1u); continue; }
            const uint64_t at = (uint64_t)s * n_sites + slot;
            const uint32_t f = filters_p[at];
            if (line_off[at] == 0 || (f & 0x80u)) continue;                     // no line (or a malformed one): no record, no Region
            filters_p[at] = (uint8_t)(f | SNPGPU_F_REGION);
            const int32_t c = col_of[slot];
            if (c >= 0) base_p[(uint64_t)s * n_cols + (uint32_t)c] = '-';
        }
    }
}

// Rows of one matrix into rows of another, picked and placed by index lists: dst[dst_index[r]] = src[src_index[r]] (an absent list: r).
// One workgroup per row at a time; 16-byte lanes where both rows and the length allow, bytes otherwise.
__global__ __launch_bounds__(256) void k_rows_copy(const uint8_t *__restrict__ src, uint64_t src_stride, const uint32_t *__restrict__ src_index,
                                                   uint8_t *__restrict__ dst, uint64_t dst_stride, const uint32_t *__restrict__ dst_index,
                                                   uint32_t n_rows, uint64_t row_bytes) {
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint8_t *s = src + (uint64_t)(src_index ? src_index[r] : r) * src_stride;
        uint8_t *d = dst + (uint64_t)(dst_index ? dst_index[r] : r) * dst_stride;
        if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0) {
            const uint64_t n16 = row_bytes >> 4;
            for (uint64_t i = threadIdx.x; i < n16; i += blockDim.x) ((uint4 *)d)[i] = ((const uint4 *)s)[i];
            for (uint64_t i = (n16 << 4) + threadIdx.x; i < row_bytes; i += blockDim.x) d[i] = s[i];
        } else {
            for (uint64_t i = threadIdx.x; i < row_bytes; i += blockDim.x) d[i] = s[i];
        }
    }
}

}  // namespace

// The gathers and scatters of whole rows the one-job pipeline needs between its steps (rows of a group's resident samples to their
// places; packed rows into sorted-id order, distance.py:76-84): device pointers, strides in bytes, index lists of n_rows entries or
// null, asynchronous on the context's stream.  The destination rows must be distinct.
extern "C" int snpgpu_rows_copy_dev(snpgpu_ctx *ctx, const void *d_src, uint64_t src_stride, const uint32_t *d_src_index, void *d_dst, uint64_t dst_stride,
                                    const uint32_t *d_dst_index, uint32_t n_rows, uint64_t row_bytes) {
    if (!ctx) return SNPGPU_E_ARG;
    if (!n_rows || !row_bytes) return SNPGPU_OK;
    if (!d_src || !d_dst || row_bytes > src_stride || row_bytes > dst_stride) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "bad row copy arguments");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const unsigned cap = (unsigned)ctx->n_cu * 16;
    k_rows_copy<<<n_rows < cap ? n_rows : cap, 256, 0, ctx->stream>>>((const uint8_t *)d_src, src_stride, d_src_index, (uint8_t *)d_dst, dst_stride, d_dst_index,
                                                                       n_rows, row_bytes);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

extern "C" int snpgpu_region_flow_dev(snpgpu_ctx *ctx, const uint8_t *d_base, const uint8_t *d_filters, const uint64_t *d_line_off,
                                      uint32_t n_samples, uint32_t n_sites, const uint32_t *d_cols, const int32_t *d_col_of, uint32_t n_cols,
                                      const uint32_t *d_excl_off, const uint32_t *d_excl_slots, uint8_t *d_out_base, uint8_t *d_out_filters,
                                      uint32_t *d_err) {
    if (!ctx) return SNPGPU_E_ARG;
    if (!n_samples) return SNPGPU_OK;
    if (n_sites && (!d_base || !d_filters || !d_line_off || !d_out_filters || !d_col_of)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (n_cols && (!d_cols || !d_out_base)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (!d_excl_off || !d_err) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    hipStream_t st = ctx->stream;
    if (n_sites) HIP_TRY(ctx, hipMemcpyAsync(d_out_filters, d_filters, (size_t)n_samples * n_sites, hipMemcpyDeviceToDevice, st));
    if (n_cols) {
        const uint64_t total = (uint64_t)n_samples * n_cols, blocks = (total + 255) / 256, cap = (uint64_t)ctx->n_cu * 16;
        k_flow_gather<<<(unsigned)(blocks < cap ? blocks : cap), 256, 0, st>>>(d_base, n_samples, n_sites, d_cols, n_cols, d_out_base);
    }
    if (n_sites && d_excl_slots) {
        const unsigned grid = n_samples < (uint32_t)ctx->n_cu * 8 ? n_samples : (unsigned)ctx->n_cu * 8;
        k_flow_exclude<<<grid, 256, 0, st>>>(d_line_off, d_excl_off, d_excl_slots, n_samples, n_sites, d_col_of, n_cols, d_out_base, d_out_filters, d_err);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}
