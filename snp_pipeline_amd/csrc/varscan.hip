// Phase-1 site calling over every line of a pileup: the counting and selection half of `VarScan mpileup2snp`.
//
// Replaces what snppipeline/call_sites.py:89-108 obtains from the VarScan v2.3.9 jar (net.sf.varscan:
// VarScan.qualityDepth, VarScan.getReadCounts, the selection tests of VarScan.callPosition): per pileup line, the raw
// depth, the number of qualities >= min-avg-qual, reads per allele and strand at that quality with their quality sums,
// the indel-carrying reads (they count in the frequency's denominator), and the tests min-coverage / min-reads2 /
// min-avg-qual / min-var-freq.  Lines that pass leave one 48-byte record per passing allele; Fisher's exact test, the
// strand filter and the VCF text are host work on those few records (snp_pipeline_amd/varscan.py).
//
// One lane per line over the line index of scan.hip (k_lines_index); a wave first copies the contiguous span of its 64
// lines to LDS with 16-byte loads, so every byte of the file crosses HBM once and the byte-wise walk of the read-base
// automaton runs out of LDS (a span over the wave's LDS — 64 very deep lines — is read from global memory instead).  The file arrives over PCIe at ~50 GB/s, so the pass as a whole is bounded by that
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

// ---- the same line walk, written for LDS: 32-bit words instead of bytes ---------------------------------------------
// 0x80 in every byte of w equal to the byte replicated in c4 / at or above the byte replicated in c4 (c4 bytes < 0x80)
__device__ __forceinline__ uint32_t eq4(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
__device__ __forceinline__ uint32_t ge4(uint32_t w, uint32_t c4) {
    return ((((w & 0x7F7F7F7Fu) | 0x80808080u) - c4) | w) & 0x80808080u;
}
// byte sets of the read-base column as bit masks over (ch & 63), for ch in [0, 64) and [64, 128)
constexpr uint64_t bit_of(char c) { return 1ull << ((unsigned)c & 63u); }
constexpr uint64_t VS_REF_LO = bit_of('.') | bit_of(',');
constexpr uint64_t VS_OWN_LO = VS_REF_LO | bit_of('*');                                   // own a quality, ch < 64
constexpr uint64_t VS_ALLELE_HI = bit_of('A') | bit_of('C') | bit_of('G') | bit_of('T') | bit_of('a') | bit_of('c') | bit_of('g') | bit_of('t');
constexpr uint64_t VS_OWN_HI = VS_ALLELE_HI | bit_of('N') | bit_of('n');

// One line out of the LDS copy (offsets into it; lines there are shorter than 32 KiB, so 16-bit fields hold any count).
// Same results as varscan_line: TABs and the quality test on four bytes at a time, the read-base walk byte by byte out
// of a cached word, counters packed so that no register array is indexed at run time.
__device__ void varscan_line_lds(const uint32_t *lds32, uint32_t p0, uint32_t end, uint64_t zero, const snpgpu_varscan_params &prm,
                                 snpgpu_varscan_site *out, uint32_t capacity, uint32_t *out_n, unsigned long long *status) {
    auto byte_at = [&](uint32_t p) -> uint32_t { return (lds32[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu; };
    while (end > p0) { const uint32_t c = byte_at(end - 1); if (c != 10u && c != 13u) break; --end; }
    if (end == p0) return;
    // the first six TABs (16 bits each): t0..t3 in P0, t4 t5 in P1
    uint64_t P0 = 0, P1 = 0;
    uint32_t nt = 0;
    {
        uint32_t wi = p0 >> 2;
        uint32_t t = eq4(lds32[wi], 0x09090909u) & (0xFFFFFFFFu << ((p0 & 3u) * 8u));
        for (;;) {
            while (t && nt < 6) {
                const uint32_t pos = wi * 4u + (((uint32_t)__ffs((int)t) - 1u) >> 3);
                t &= t - 1u;
                if (pos >= end) { t = 0; break; }
                if (nt < 4) P0 |= (uint64_t)pos << (16u * nt); else P1 |= (uint64_t)pos << (16u * (nt - 4u));
                ++nt;
            }
            if (nt == 6 || (wi + 1u) * 4u >= end) break;
            ++wi;
            t = eq4(lds32[wi], 0x09090909u);
        }
    }
    if (nt == 5) { P1 |= (uint64_t)end << 16; ++nt; }
    const uint32_t t0 = (uint32_t)P0 & 0xFFFFu, t1 = (uint32_t)(P0 >> 16) & 0xFFFFu, t2 = (uint32_t)(P0 >> 32) & 0xFFFFu, t3 = (uint32_t)(P0 >> 48),
                   t4 = (uint32_t)P1 & 0xFFFFu, t5 = (uint32_t)(P1 >> 16) & 0xFFFFu;
    bool ok = nt == 6 && t0 > p0 && t1 > t0 + 1 && t2 == t1 + 2 && t3 > t2 + 1 && t4 > t3 + 1 && t5 > t4 + 1;
    uint32_t depth = 0;
    if (ok) {
        if (t3 - t2 - 1 > 9) ok = false;
        for (uint32_t p = t2 + 1; ok && p < t3; ++p) {
            const uint32_t c = byte_at(p);
            if (!is_digit(c)) ok = false;
            depth = depth * 10u + (c - 0x30u);
        }
    }
    if (!ok) {
        atomicMin(status, (unsigned long long)(zero + p0));
        return;
    }
    if (depth < prm.min_coverage) return;
    const uint32_t b0 = t3 + 1, b1 = t4, q0 = t4 + 1, q1 = t5;
    const uint32_t qmin = prm.min_avg_qual + 33u;
    // qualities at or above the threshold, four at a time (a threshold above 127 can only be met by bytes >= 0x80)
    uint32_t dp = 0;
    {
        const uint32_t c4 = (qmin < 128u ? qmin : 128u) * 0x01010101u;
        for (uint32_t wi = q0 >> 2; wi * 4u < q1; ++wi) {
            const uint32_t w = lds32[wi];
            uint32_t f = qmin < 128u ? ge4(w, c4) : 0u;
            if (qmin >= 128u)
                for (int b = 0; b < 4; ++b) f |= ((w >> (8 * b)) & 0xFFu) >= qmin ? 0x80u << (8 * b) : 0u;
            if (wi * 4u < q0) f &= 0xFFFFFFFFu << ((q0 & 3u) * 8u);
            if (wi * 4u + 4u > q1) f &= 0xFFFFFFFFu >> ((4u - (q1 & 3u)) * 8u);
            dp += (uint32_t)__popc(f);
        }
    }
    if (dp < prm.min_coverage) return;
    uint32_t ref = byte_at(t1 + 1);
    if (ref >= 0x61u && ref <= 0x7Au) ref -= 32u;
    // counters: reference per strand + quality sum; alleles indexed (ch >> 1) & 3 = A 0, C 1, T 2, G 3:
    // F / R = forward / reverse counts, 16 bits each; QS01 / QS23 = quality sums, 32 bits each
    uint32_t rf_f = 0, rf_r = 0, rf_q = 0, indel = 0;
    uint64_t F = 0, R = 0, QS01 = 0, QS23 = 0;
    uint32_t bw = lds32[b0 >> 2], qw = 0, qwi = 0xFFFFFFFFu;
    uint32_t j = q0;
    const uint32_t c4 = (qmin < 128u ? qmin : 128u) * 0x01010101u;
    for (uint32_t i = b0; i < b1; ++i) {
        if ((i & 3u) == 0u) {
            bw = lds32[i >> 2];
            // four reference matches in a row ('.' 0x2E / ',' 0x2C), the common case: their four qualities in one go
            if ((bw & 0xFDFDFDFDu) == 0x2C2C2C2Cu && i + 4u <= b1 && j + 4u <= q1 && qmin < 128u) {
                const uint64_t two = (uint64_t)lds32[j >> 2] | ((uint64_t)lds32[(j >> 2) + 1u] << 32);
                const uint32_t qq = (uint32_t)(two >> ((j & 3u) * 8u));
                const uint32_t good = ge4(qq, c4) >> 7;                                // 0x01 per quality at or above the threshold
                const uint32_t fwd = (bw >> 1) & good;                                 // ... that belongs to a '.'
                rf_f += (uint32_t)__popc(fwd);
                rf_r += (uint32_t)__popc(good ^ fwd);
                rf_q += __builtin_amdgcn_udot4(qq & (good * 0xFFu), 0x01010101u, 0u, false) - 33u * (uint32_t)__popc(good);
                j += 4u;
                i += 3u;                                                               // (the loop adds the fourth)
                continue;
            }
        }
        const uint32_t ch = (bw >> ((i & 3u) * 8u)) & 0xFFu;
        const uint64_t bit = 1ull << (ch & 63u);
        const bool lo = ch < 64u, hi = (ch ^ 64u) < 64u;
        const bool is_ref = lo && (bit & VS_REF_LO), is_all = hi && (bit & VS_ALLELE_HI);
        const bool owns = (lo && (bit & VS_OWN_LO)) || (hi && (bit & VS_OWN_HI));
        if (owns) {
            uint32_t q = 33u;                                                          // past the end: quality 0
            if (j < q1) {
                if ((j >> 2) != qwi) { qwi = j >> 2; qw = lds32[qwi]; }
                q = (qw >> ((j & 3u) * 8u)) & 0xFFu;
            }
            ++j;
            if (q >= qmin) {
                const uint32_t qv = q - 33u;
                if (is_ref) {
                    if (ch == '.') ++rf_f; else ++rf_r;
                    rf_q += qv;
                } else if (is_all) {
                    const uint32_t idx = (ch >> 1) & 3u;
                    const uint64_t one = 1ull << (16u * idx);
                    if (ch & 0x20u) R += one; else F += one;
                    const uint64_t qa = (uint64_t)qv << (32u * (idx & 1u));
                    if (idx & 2u) QS23 += qa; else QS01 += qa;
                }
            }
        } else if (ch == '+' || ch == '-') {                                           // digits, then that many bases; no quality
            uint32_t k = i + 1;
            uint64_t size = 0;
            while (k < b1) {
                const uint32_t c = byte_at(k);
                if (!is_digit(c)) break;
                if (size < (1ull << 40)) size = size * 10 + (c - 0x30u);
                ++k;
            }
            if (k > i + 1) {
                ++indel;
                i = size >= (uint64_t)(b1 - k) ? b1 - 1 : (uint32_t)(k + (uint32_t)size - 1);
                if ((i & 3u) != 3u) bw = lds32[i >> 2];                                // the next byte's word
            }
        } else if (ch == '^') {
            ++i;                                                                       // the next byte is a mapping quality
            if ((i & 3u) != 3u && i < b1) bw = lds32[i >> 2];
        }                                                                              // '$' and the rest: skipped
    }
    const uint32_t reads1 = rf_f + rf_r;
    uint32_t total = reads1 + indel;
    for (int a = 0; a < 4; ++a) total += (uint32_t)(F >> (16 * a)) & 0xFFFFu, total += (uint32_t)(R >> (16 * a)) & 0xFFFFu;
#pragma unroll
    for (int a = 0; a < 4; ++a) {                                                      // in the order A, C, G, T
        const int idx = a == 0 ? 0 : a == 1 ? 1 : a == 2 ? 3 : 2;
        const uint32_t allele = a == 0 ? 'A' : a == 1 ? 'C' : a == 2 ? 'G' : 'T';
        const uint32_t af = (uint32_t)(F >> (16 * idx)) & 0xFFFFu, ar = (uint32_t)(R >> (16 * idx)) & 0xFFFFu;
        const uint32_t aq = (uint32_t)((idx & 2 ? QS23 : QS01) >> (32 * (idx & 1)));
        const uint32_t reads2 = af + ar;
        if (allele == ref || reads2 == 0) continue;
        if (reads2 < prm.min_reads2 || aq / reads2 < prm.min_avg_qual) continue;
        if ((double)reads2 / (double)total < prm.min_var_freq) continue;
        const uint32_t slot = atomicAdd(out_n, 1u);
        if (slot >= capacity) continue;
        snpgpu_varscan_site s;
        s.line_off = zero + p0;
        s.sdp = depth; s.dp = dp; s.total = total;
        s.rdf = rf_f; s.rdr = rf_r; s.ref_qual_sum = rf_q;
        s.adf = af; s.adr = ar; s.alt_qual_sum = aq;
        s.ref_base = (uint8_t)ref; s.alt_base = (uint8_t)allele; s.reserved[0] = s.reserved[1] = 0;
        out[slot] = s;
    }
}

// A workgroup is ONE wave and takes 64 consecutive lines: their bytes are one contiguous span of the file, copied to LDS
// with 16-byte loads (coalesced; every byte of the file crosses HBM once) when it fits the launch's LDS size, and each lane
// then walks its own line there.  One wave per workgroup: no wave ever waits at a barrier for a slower one, and the LDS
// size (by the file's mean line length) sets how many waves a CU holds.
__global__ __launch_bounds__(64) void k_varscan_lines(const uint8_t *__restrict__ buf, uint64_t nbytes, const uint64_t *__restrict__ line_off,
                                                      uint64_t n_lines, snpgpu_varscan_params prm, snpgpu_varscan_site *out, uint32_t capacity,
                                                      uint32_t *out_n, unsigned long long *status, uint32_t lds_bytes) {
    extern __shared__ uint4 vs_lds[];
    const uint64_t n_groups = (n_lines + 63) / 64;
    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint64_t first = grp * 64, last = first + 64 < n_lines ? first + 64 : n_lines;
        const uint64_t s0 = line_off[first] - 1, s1 = last < n_lines ? line_off[last] - 1 : nbytes;
        // 16-byte chunks of the aligned span [a0, a1) that covers [s0, s1)
        const uint64_t a0 = ((uintptr_t)buf + s0) & ~(uint64_t)15, a1 = (((uintptr_t)buf + s1) + 15) & ~(uint64_t)15;
        const bool staged = a1 - a0 <= lds_bytes;
        if (staged) {
            const uint4 *src = (const uint4 *)a0;
            for (uint32_t c = threadIdx.x; c < (uint32_t)((a1 - a0) / 16); c += 64) vs_lds[c] = src[c];
        }
        __syncthreads();
        const uint64_t line = first + threadIdx.x;
        if (line < last) {
            const uint64_t p0 = line_off[line] - 1, end = line + 1 < n_lines ? line_off[line + 1] - 1 : nbytes;
            const uint64_t zero = a0 - (uintptr_t)buf;                                  // file offset of LDS byte 0 (mod 2^64)
            if (staged) varscan_line_lds((const uint32_t *)vs_lds, (uint32_t)(p0 - zero), (uint32_t)(end - zero), zero, prm, out, capacity, out_n, status);
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
    // LDS per wave: 64 lines of the file's mean length with 40 % to spare, 4 .. 32 KiB (a span that does not fit is walked in
    // global memory); a CU's 160 KiB then hold 160 / that many waves
    const uint64_t mean = nbytes / n_lines + 1;
    uint32_t lds = 4096;
    while (lds < VS_LDS_BYTES && (uint64_t)lds * 5 < mean * 64 * 7) lds *= 2;
    const uint32_t waves_per_cu = 160 * 1024 / lds < 32 ? 160 * 1024 / lds : 32;
    const uint64_t groups = (n_lines + 63) / 64;
    const uint64_t cap = (uint64_t)ctx->n_cu * waves_per_cu * 4;
    const unsigned grid = (unsigned)(groups < cap ? groups : cap);
    k_varscan_lines<<<grid, 64, lds, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n, (unsigned long long *)d_status, lds);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}
