// Phase-1 site calling over every line of a pileup: the counting and selection half of `VarScan mpileup2snp`.
//
// Replaces what snppipeline/call_sites.py:89-108 obtains from the VarScan v2.3.9 jar (net.sf.varscan:
// VarScan.qualityDepth, VarScan.getReadCounts, the selection tests of VarScan.callPosition): per pileup line, the raw
// depth, the number of qualities >= min-avg-qual, reads per allele and strand at that quality with their quality sums,
// the indel-carrying reads (they count in the frequency's denominator), and the tests min-coverage / min-reads2 /
// min-avg-qual / min-var-freq.  Lines that pass leave one 48-byte record per passing allele; Fisher's exact test, the
// strand filter and the VCF text are host work on those few records (snp_pipeline_amd/varscan.py).
//
// One lane per line over the line index of scan.hip (k_lines_index); a block of 256 / 128 / 64 lanes (by mean line length)
// first copies the contiguous span of its lines to LDS with 16-byte loads, so every byte of the file crosses HBM once and
// the byte-wise walk of the read-base automaton runs out of LDS (a span over 32 KiB — a block of very deep lines — is
// read from global memory instead).  The file arrives over PCIe at ~50 GB/s, so the pass as a whole is bounded by that
// copy, not by this kernel.  The read-base automaton follows the restatement in oracle/varscan_oracle.py (which tests
// compare it with); see its header for what the reference's fixtures pin.
#include "internal.h"

namespace {

struct Acc { uint32_t f, r, q; };

__device__ __forceinline__ bool is_digit(uint32_t c) { return c - 0x30u < 10u; }

constexpr uint32_t VS_LDS_BYTES = 32 * 1024;

// Where a block reads its lines: the LDS copy of its span (32-bit offsets into it) or the file in global memory.
struct LdsBytes {
    const uint8_t *l;
    __device__ __forceinline__ uint32_t operator()(uint32_t p) const { return l[p]; }
};
struct GlobalBytes {
    const uint8_t *g;
    __device__ __forceinline__ uint32_t operator()(uint64_t p) const { return g[p]; }
};

// One line: VarScan.qualityDepth + VarScan.getReadCounts + the count tests of VarScan.callPosition.
// Off: the offset type of the reader B; offset 0 of B is byte `zero` of the file.
template <typename Off, typename Rd>
__device__ void varscan_line(const Rd B, Off p0, Off end, uint64_t zero, const snpgpu_varscan_params &prm, snpgpu_varscan_site *out, uint32_t capacity,
                             uint32_t *out_n, unsigned long long *status) {
    while (end > p0 && (B(end - 1) == 10u || B(end - 1) == 13u)) --end;               // readLine() strips the terminator
    if (end == p0) return;                                                             // an empty line
    // String.split("\t"): the first five TABs delimit chrom, position, ref, depth, bases; qualities run to the next TAB
    Off tab[6];
    int nt = 0;
    for (Off p = p0; p < end && nt < 6; ++p)
        if (B(p) == 9u) tab[nt++] = p;
    if (nt == 5) tab[nt++] = end;
    bool ok = nt == 6 && tab[0] > p0 && tab[1] > tab[0] + 1 && tab[2] == tab[1] + 2 && tab[3] > tab[2] + 1 && tab[4] > tab[3] + 1 &&
              tab[5] > tab[4] + 1;                                                     // six non-empty columns, a one-byte reference
    uint32_t depth = 0;
    if (ok) {
        if (tab[3] - tab[2] - 1 > 9) ok = false;
        for (Off p = tab[2] + 1; ok && p < tab[3]; ++p) {
            const uint32_t c = B(p);
            if (!is_digit(c)) ok = false;
            depth = depth * 10u + (c - 0x30u);
        }
    }
    if (!ok) {
        atomicMin(status, (unsigned long long)(zero + p0));
        return;
    }
    if (depth < prm.min_coverage) return;
    const Off b0 = tab[3] + 1, b1 = tab[4], q0 = tab[4] + 1, q1 = tab[5];
    const uint32_t qmin = prm.min_avg_qual + 33u;
    uint32_t dp = 0;
    for (Off p = q0; p < q1; ++p) dp += B(p) >= qmin ? 1u : 0u;
    if (dp < prm.min_coverage) return;
    uint32_t ref = B(tab[1] + 1);
    if (ref >= 0x61u && ref <= 0x7Au) ref -= 32u;
    Acc rf{0, 0, 0}, al[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    uint32_t indel = 0;
    Off j = q0;
    for (Off i = b0; i < b1; ++i) {
        const uint32_t ch = B(i);
        const uint32_t q = j < q1 ? B(j) : 33u;                                        // past the end: quality 0
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
        } else if (ch == '+' || ch == '-') {                                           // digits, then that many bases; no quality
            Off k = i + 1;
            uint64_t size = 0;
            while (k < b1 && is_digit(B(k))) { if (size < (1ull << 40)) size = size * 10 + (B(k) - 0x30u); ++k; }
            if (k > i + 1) {
                ++indel;
                i = size >= (uint64_t)(b1 - k) ? b1 - 1 : (Off)(k + (Off)size - 1);     // the loop's ++i steps past the last indel base
            }
        } else if (up == 'N' || ch == '*') {
            ++j;                                                                       // not counted, but owns a quality
        } else if (ch == '^') {
            ++i;                                                                       // the next byte is a mapping quality
        }                                                                              // '$' and the rest: skipped
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
        s.line_off = zero + p0;
        s.sdp = depth; s.dp = dp; s.total = total;
        s.rdf = rf.f; s.rdr = rf.r; s.ref_qual_sum = rf.q;
        s.adf = al[a].f; s.adr = al[a].r; s.alt_qual_sum = al[a].q;
        s.ref_base = (uint8_t)ref; s.alt_base = (uint8_t)allele; s.reserved[0] = s.reserved[1] = 0;
        out[slot] = s;
    }
}

// A block takes kThreads consecutive lines: their bytes are one contiguous span of the file, copied to LDS with 16-byte
// loads (coalesced; every byte of the file crosses HBM once) when it fits, and each lane then walks its own line there.
template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_varscan_lines(const uint8_t *__restrict__ buf, uint64_t nbytes, const uint64_t *__restrict__ line_off,
                                                            uint64_t n_lines, snpgpu_varscan_params prm, snpgpu_varscan_site *out, uint32_t capacity,
                                                            uint32_t *out_n, unsigned long long *status) {
    extern __shared__ uint4 vs_lds[];
    const uint64_t n_groups = (n_lines + kThreads - 1) / kThreads;
    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint64_t first = grp * kThreads, last = first + kThreads < n_lines ? first + kThreads : n_lines;
        const uint64_t s0 = line_off[first] - 1, s1 = last < n_lines ? line_off[last] - 1 : nbytes;
        // 16-byte chunks of the aligned span [a0, a1) that covers [s0, s1)
        const uint64_t a0 = ((uintptr_t)buf + s0) & ~(uint64_t)15, a1 = (((uintptr_t)buf + s1) + 15) & ~(uint64_t)15;
        const bool staged = a1 - a0 <= VS_LDS_BYTES;
        if (staged) {
            const uint4 *src = (const uint4 *)a0;
            for (uint32_t c = threadIdx.x; c < (uint32_t)((a1 - a0) / 16); c += kThreads) vs_lds[c] = src[c];
        }
        __syncthreads();
        const uint64_t line = first + threadIdx.x;
        if (line < last) {
            const uint64_t p0 = line_off[line] - 1, end = line + 1 < n_lines ? line_off[line + 1] - 1 : nbytes;
            const uint64_t zero = a0 - (uintptr_t)buf;                                  // file offset of LDS byte 0 (mod 2^64)
            if (staged) varscan_line<uint32_t>(LdsBytes{(const uint8_t *)vs_lds}, (uint32_t)(p0 - zero), (uint32_t)(end - zero), zero, prm, out, capacity, out_n, status);
            else varscan_line<uint64_t>(GlobalBytes{buf}, p0, end, 0, prm, out, capacity, out_n, status);
        }
        __syncthreads();
    }
}

}  // namespace

// d_n: one zeroed word; d_status: one u64 preset to UINT64_MAX (becomes the offset of the first malformed line)
int snpgpu_enqueue_varscan(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, const uint64_t *d_line_off, uint64_t n_lines,
                           const snpgpu_varscan_params *prm, snpgpu_varscan_site *d_sites, uint32_t capacity, uint32_t *d_n, uint64_t *d_status) {
    if (n_lines == 0) return SNPGPU_OK;
    static bool attr_set = false;
    if (!attr_set) {
        for (auto f : {(const void *)k_varscan_lines<256>, (const void *)k_varscan_lines<128>, (const void *)k_varscan_lines<64>})
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VS_LDS_BYTES);
        attr_set = true;
    }
    // lines per block by the mean line length, so that a block's span fits the 32 KiB it may stage (5 blocks per CU)
    const uint64_t mean = nbytes / n_lines + 1;
    const int threads = mean * 256 <= VS_LDS_BYTES * 3 / 4 ? 256 : (mean * 128 <= VS_LDS_BYTES * 3 / 4 ? 128 : 64);
    const uint64_t groups = (n_lines + threads - 1) / threads;
    const uint64_t cap = (uint64_t)ctx->n_cu * 5 * 8;
    const unsigned grid = (unsigned)(groups < cap ? groups : cap);
    auto *st = (unsigned long long *)d_status;
    if (threads == 256) k_varscan_lines<256><<<grid, 256, VS_LDS_BYTES, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n, st);
    else if (threads == 128) k_varscan_lines<128><<<grid, 128, VS_LDS_BYTES, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n, st);
    else k_varscan_lines<64><<<grid, 64, VS_LDS_BYTES, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n, st);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}
