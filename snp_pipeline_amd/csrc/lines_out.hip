// call_consensus --vcfAllPos: the per-line records on their way out of the device.
//
// The reference writes one consensus.vcf row per pileup LINE (call_consensus.py:148-151, vcf_writer.py:381-435).  The call kernels
// leave a 128-byte snpgpu_site_counts per line — 685 MB for the 5 M lines of a 5 Mbp sample, which is what the all-positions pass
// spent its time moving over the host link (69 ms for a 0.5 ms kernel).  Nearly every line has at most three distinct symbols and depths
// far below 65 536, so:
//   k_compact_lines   one thread per line: the record packed into 32 bytes (snpgpu_line_record) when it fits, else marked "wide";
//                     per-workgroup counts of the wide ones
//   (exclusive scan of those counts: prims.h)
//   k_gather_wide     the wide lines' indices (ascending) and full records, contiguous
// and the host formats rows from the 32-byte records (the full record only where a line is wide): 40 bytes per line cross the link.
// HBM-bound elementwise kernels: 128 + 1 bytes read, 32 written per line.
#include <string.h>

#include "internal.h"
#include "prims.h"

namespace {

constexpr uint32_t LINES_OUT_THREADS = 256;

__device__ __forceinline__ bool fits_u16(uint32_t a, uint32_t b, uint32_t c) { return (a | b | c) < 65536u; }

__global__ __launch_bounds__(LINES_OUT_THREADS) void k_compact_lines(const snpgpu_site_counts *__restrict__ counts, const uint8_t *__restrict__ flags,
                                                                    uint64_t n, snpgpu_line_record *__restrict__ out, uint32_t *__restrict__ block_wide) {
    __shared__ uint32_t lds[17];
    const uint64_t i = (uint64_t)blockIdx.x * LINES_OUT_THREADS + threadIdx.x;
    uint32_t wide = 0;
    if (i < n) {
        // the record as 16-byte loads (a thread owns a whole 128-byte record: one cache line)
        const uint4 *p = (const uint4 *)&counts[i];
        const uint4 h0 = p[0], h1 = p[1], t = p[2], f = p[4], r = p[6];
        const uint32_t raw = h0.x, good = h0.y, fgood = h0.z, rgood = h0.w;
        const uint32_t nsym = h1.x, bytes = h1.y;               // ref_base | cons_base << 8 | filters << 16 | status << 24
        const uint32_t syms = h1.z;                             // sym[0..4)
        const uint32_t status = bytes >> 24;
        // packed: a well-formed line (or one without a record: status 0) with at most three symbols, no spill record, 16-bit counts,
        // depths that are the sums of the symbols' counts, and nothing counted past the symbols the record says it has
        const bool ok = status <= SNPGPU_ST_OK && nsym <= (uint32_t)SNPGPU_LINE_SYMS && fits_u16(t.x, f.x, r.x) && fits_u16(t.y, f.y, r.y) && fits_u16(t.z, f.z, r.z) &&
                        good == t.x + t.y + t.z && fgood == f.x + f.y + f.z && rgood == r.x + r.y + r.z &&
                        (nsym >= 3u || t.z == 0u) && (nsym >= 2u || t.y == 0u) && (nsym >= 1u || t.x == 0u);
        wide = ok ? 0u : 1u;
        // 32 bytes: two 16-byte stores.  Layout: raw u32 | total[3] fwd[3] rev[3] u16 | sym[3] | ref cons filters status | n_symbols | site_flags | 0
        const uint32_t w1 = (t.x & 0xFFFFu) | (t.y << 16), w2 = (t.z & 0xFFFFu) | (f.x << 16), w3 = (f.y & 0xFFFFu) | (f.z << 16);
        const uint32_t w4 = (r.x & 0xFFFFu) | (r.y << 16);
        const uint32_t w5 = (r.z & 0xFFFFu) | ((syms & 0xFFFFu) << 16);                                  // bytes 20-21 rev[2], 22-23 sym[0], sym[1]
        const uint32_t w6 = ((syms >> 16) & 0xFFu) | (bytes << 8);                                       // 24 sym[2], 25 ref, 26 cons, 27 filters
        const uint32_t w7 = (bytes >> 24) | ((ok ? nsym : (uint32_t)SNPGPU_LINE_WIDE) << 8) | ((uint32_t)flags[i] << 16);   // 28 status, 29 n_symbols, 30 site_flags, 31 0
        uint4 *q = (uint4 *)&out[i];
        q[0] = make_uint4(raw, w1, w2, w3);
        q[1] = make_uint4(w4, w5, w6, w7);
    }
    uint32_t total;
    (void)block_exclusive_sum(wide, lds, total);
    if (threadIdx.x == 0) block_wide[blockIdx.x] = total;
}

__global__ __launch_bounds__(LINES_OUT_THREADS) void k_gather_wide(const snpgpu_site_counts *__restrict__ counts, const snpgpu_line_record *__restrict__ recs,
                                                                  uint64_t n, const uint32_t *__restrict__ block_base, uint32_t capacity,
                                                                  uint32_t *__restrict__ wide_index, snpgpu_site_counts *__restrict__ wide) {
    __shared__ uint32_t lds[17];
    const uint64_t i = (uint64_t)blockIdx.x * LINES_OUT_THREADS + threadIdx.x;
    const uint32_t is_wide = (i < n && recs[i].n_symbols == SNPGPU_LINE_WIDE) ? 1u : 0u;
    uint32_t total;
    const uint32_t at = block_base[blockIdx.x] + block_exclusive_sum(is_wide, lds, total);
    if (is_wide && at < capacity) {
        wide_index[at] = (uint32_t)i;
        const uint4 *p = (const uint4 *)&counts[i];
        uint4 *q = (uint4 *)&wide[at];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = p[k];
    }
}

}  // namespace

static_assert(sizeof(snpgpu_line_record) == 32, "snpgpu_line_record is 32 bytes");
static_assert(sizeof(snpgpu_site_counts) == 128, "snpgpu_site_counts is 128 bytes");

size_t snpgpu_compact_lines_workspace_words(uint64_t n_lines) {
    const uint64_t nb = (n_lines + LINES_OUT_THREADS - 1) / LINES_OUT_THREADS;
    return (size_t)nb + 1 + prim_scan_workspace_words(nb + 1);
}

// d_counts / d_flags [n_lines] -> d_recs [n_lines]; ws: snpgpu_compact_lines_workspace_words(n_lines) words; *d_n_wide (a pointer into
// ws) holds the number of wide lines once the stream has got there.
int snpgpu_enqueue_compact_lines(snpgpu_ctx *ctx, const snpgpu_site_counts *d_counts, const uint8_t *d_flags, uint64_t n_lines, snpgpu_line_record *d_recs,
                                 uint32_t *ws, uint32_t **d_n_wide) {
    const uint64_t nb = (n_lines + LINES_OUT_THREADS - 1) / LINES_OUT_THREADS;
    if (nb > 0x7FFFFFFFull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "too many lines for one pass");
    uint32_t *block_wide = ws, *scan_ws = ws + nb + 1;
    if (nb) k_compact_lines<<<(unsigned)nb, LINES_OUT_THREADS, 0, ctx->stream>>>(d_counts, d_flags, n_lines, d_recs, block_wide);
    prim_exclusive_scan_u32(ctx->stream, block_wide, block_wide, nb, scan_ws, d_n_wide);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// after snpgpu_enqueue_compact_lines with the same ws: the wide lines' indices and records, ascending, at most `capacity` of them
int snpgpu_enqueue_gather_wide(snpgpu_ctx *ctx, const snpgpu_site_counts *d_counts, const snpgpu_line_record *d_recs, uint64_t n_lines, const uint32_t *ws,
                               uint32_t capacity, uint32_t *d_wide_index, snpgpu_site_counts *d_wide) {
    const uint64_t nb = (n_lines + LINES_OUT_THREADS - 1) / LINES_OUT_THREADS;
    if (nb && capacity) k_gather_wide<<<(unsigned)nb, LINES_OUT_THREADS, 0, ctx->stream>>>(d_counts, d_recs, n_lines, ws, capacity, d_wide_index, d_wide);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// The full record a packed one stands for (host side: the row formatter and the tests' expander work on snpgpu_site_counts).
void snpgpu_expand_line_record(const snpgpu_line_record &r, snpgpu_site_counts *out) {
    memset(out, 0, sizeof *out);
    out->raw_depth = r.raw_depth;
    out->good_depth = (uint32_t)r.total[0] + r.total[1] + r.total[2];
    out->fwd_good_depth = (uint32_t)r.fwd[0] + r.fwd[1] + r.fwd[2];
    out->rev_good_depth = (uint32_t)r.rev[0] + r.rev[1] + r.rev[2];
    out->n_symbols = r.n_symbols;
    out->ref_base = r.ref_base; out->cons_base = r.cons_base; out->filters = r.filters; out->status = r.status;
    for (int k = 0; k < SNPGPU_LINE_SYMS; ++k) { out->sym[k] = r.sym[k]; out->total[k] = r.total[k]; out->fwd[k] = r.fwd[k]; out->rev[k] = r.rev[k]; }
}
