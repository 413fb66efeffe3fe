// Phase-1 site calling over every line of a pileup: the counting and selection half of `VarScan mpileup2snp`.
//
// Replaces what snppipeline/call_sites.py:89-108 obtains from the VarScan v2.3.9 jar (net.sf.varscan:
// VarScan.qualityDepth, VarScan.getReadCounts, the selection tests of VarScan.callPosition): per pileup line, the raw
// depth, the number of qualities >= min-avg-qual, reads per allele and strand at that quality with their quality sums,
// the indel-carrying reads (they count in the frequency's denominator), and the tests min-coverage / min-reads2 /
// min-avg-qual / min-var-freq.  Lines that pass leave one 48-byte record per passing allele; Fisher's exact test, the
// strand filter and the VCF text are host work on those few records (snp_pipeline_amd/varscan.py).
//
// One lane per line over the line index of scan.hip (k_lines_index): neighbouring lanes read neighbouring lines, so the
// wave's byte loads fall into a few KiB that stay in the vector L1; the pass is bounded by the host-to-device copy of
// the file (~50 GB/s) long before it is bounded by this kernel.  The read-base automaton follows the restatement in
// oracle/varscan_oracle.py (which tests compare it with); see its header for what the reference's fixtures pin.
#include "internal.h"

namespace {

struct Acc { uint32_t f, r, q; };

__device__ __forceinline__ bool is_digit(uint32_t c) { return c - 0x30u < 10u; }

__global__ __launch_bounds__(256) void k_varscan_lines(const uint8_t *__restrict__ buf, uint64_t nbytes, const uint64_t *__restrict__ line_off,
                                                       uint64_t n_lines, snpgpu_varscan_params prm, snpgpu_varscan_site *out, uint32_t capacity,
                                                       uint32_t *out_n, unsigned long long *status) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t line = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; line < n_lines; line += stride) {
        const uint64_t p0 = line_off[line] - 1;
        uint64_t end = line + 1 < n_lines ? line_off[line + 1] - 1 : nbytes;
        while (end > p0 && (buf[end - 1] == 10u || buf[end - 1] == 13u)) --end;       // readLine() strips the terminator
        if (end == p0) continue;                                                      // an empty line
        // String.split("\t"): the first five TABs delimit chrom, position, ref, depth, bases; qualities run to the next TAB
        uint64_t tab[6];
        int nt = 0;
        for (uint64_t p = p0; p < end && nt < 6; ++p)
            if (buf[p] == 9u) tab[nt++] = p;
        if (nt == 5) tab[nt++] = end;
        bool ok = nt == 6 && tab[0] > p0 && tab[1] > tab[0] + 1 && tab[2] == tab[1] + 2 && tab[3] > tab[2] + 1 && tab[4] > tab[3] + 1 &&
                  tab[5] > tab[4] + 1;                                                // six non-empty columns, a one-byte reference
        uint32_t depth = 0;
        if (ok) {
            if (tab[3] - tab[2] - 1 > 9) ok = false;
            for (uint64_t p = tab[2] + 1; ok && p < tab[3]; ++p) {
                const uint32_t c = buf[p];
                if (!is_digit(c)) ok = false;
                depth = depth * 10u + (c - 0x30u);
            }
        }
        if (!ok) {
            atomicMin(status, (unsigned long long)p0);
            continue;
        }
        if (depth < prm.min_coverage) continue;
        const uint64_t b0 = tab[3] + 1, b1 = tab[4], q0 = tab[4] + 1, q1 = tab[5];
        const uint32_t qmin = prm.min_avg_qual + 33u;
        uint32_t dp = 0;
        for (uint64_t p = q0; p < q1; ++p) dp += buf[p] >= qmin ? 1u : 0u;
        if (dp < prm.min_coverage) continue;
        uint32_t ref = buf[tab[1] + 1];
        if (ref >= 0x61u && ref <= 0x7Au) ref -= 32u;
        Acc rf{0, 0, 0}, al[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        uint32_t indel = 0;
        uint64_t j = q0;
        for (uint64_t i = b0; i < b1; ++i) {
            const uint32_t ch = buf[i];
            const uint32_t q = j < q1 ? (uint32_t)buf[j] : 33u;                       // past the end: quality 0
            const bool good = q >= qmin;
            const uint32_t up = ch & 0xDFu;
            if (ch == '.' || ch == ',') {
                if (good) { if (ch == '.') ++rf.f; else ++rf.r; rf.q += q - 33u; }
                ++j;
            } else if (up == 'A' || up == 'C' || up == 'G' || up == 'T') {
                if (good) {
                    Acc &a = al[up == 'A' ? 0 : up == 'C' ? 1 : up == 'G' ? 2 : 3];
                    if (ch < 0x61u) ++a.f; else ++a.r;
                    a.q += q - 33u;
                }
                ++j;
            } else if (ch == '+' || ch == '-') {                                       // digits, then that many bases; no quality
                uint64_t k = i + 1;
                uint64_t size = 0;
                while (k < b1 && is_digit(buf[k])) { if (size < (1ull << 40)) size = size * 10 + (buf[k] - 0x30u); ++k; }
                if (k > i + 1) {
                    ++indel;
                    i = k + size - 1;                                                  // the loop's ++i steps past the last indel base
                }
            } else if (up == 'N' || ch == '*') {
                ++j;                                                                   // not counted, but owns a quality
            } else if (ch == '^') {
                ++i;                                                                   // the next byte is a mapping quality
            }                                                                          // '$' and the rest: skipped
        }
        const uint32_t reads1 = rf.f + rf.r;
        uint32_t total = reads1 + indel;
        for (int a = 0; a < 4; ++a) total += al[a].f + al[a].r;
        for (int a = 0; a < 4; ++a) {
            const uint32_t allele = a == 0 ? 'A' : a == 1 ? 'C' : a == 2 ? 'G' : 'T';
            const uint32_t reads2 = al[a].f + al[a].r;
            if (allele == ref || reads2 == 0) continue;
            if (reads2 < prm.min_reads2 || al[a].q / reads2 < prm.min_avg_qual) continue;
            if ((double)reads2 / (double)total < prm.min_var_freq) continue;
            const uint32_t slot = atomicAdd(out_n, 1u);
            if (slot >= capacity) continue;
            snpgpu_varscan_site s;
            s.line_off = p0;
            s.sdp = depth; s.dp = dp; s.total = total;
            s.rdf = rf.f; s.rdr = rf.r; s.ref_qual_sum = rf.q;
            s.adf = al[a].f; s.adr = al[a].r; s.alt_qual_sum = al[a].q;
            s.ref_base = (uint8_t)ref; s.alt_base = (uint8_t)allele; s.reserved[0] = s.reserved[1] = 0;
            out[slot] = s;
        }
    }
}

}  // namespace

// d_n: one zeroed word; d_status: one u64 preset to UINT64_MAX (becomes the offset of the first malformed line)
int snpgpu_enqueue_varscan(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, const uint64_t *d_line_off, uint64_t n_lines,
                           const snpgpu_varscan_params *prm, snpgpu_varscan_site *d_sites, uint32_t capacity, uint32_t *d_n, uint64_t *d_status) {
    if (n_lines == 0) return SNPGPU_OK;
    const uint64_t blocks = (n_lines + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->n_cu * 32;
    k_varscan_lines<<<(unsigned)(blocks < cap ? blocks : cap), 256, 0, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n,
                                                                                       (unsigned long long *)d_status);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}
