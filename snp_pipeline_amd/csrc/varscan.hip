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

constexpr uint32_t VS_CAND_LOCAL = 256;
// A list entry: the line's index (bits 0-31), and for a line whose shape k_varscan_select has checked (VS_ENTRY_PLAIN) its depth
// (bits 32-51) and where, counted from the line's first byte, its second and fourth TAB are (bits 52-56, 57-61)
constexpr uint64_t VS_ENTRY_PLAIN = 1ull << 63;
          // candidate lines a select wave collects in LDS before it takes a place on the list

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
    // String.split drops TRAILING empty strings only: more than five columns are left when some byte after the fifth TAB is not a
    // TAB; chrom, position, reference (one byte here) and depth must not be empty, the read bases and the qualities may be
    bool ok = nt == 6 && tab[0] > p0 && tab[1] > tab[0] + 1 && tab[2] == tab[1] + 2 && tab[3] > tab[2] + 1;
    if (ok && tab[5] == tab[4] + 1) {                                                  // an empty quality column: is there anything behind it?
        ok = false;
        for (Off p = tab[5] + 1; p < end && !ok; ++p) ok = B(p) != 9u;
    }
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

// The counting part of a line whose columns are known: ref at `ref_at`, read bases [b0, b1), qualities [q0, q1); p0 is the
// line's first byte (for the record).  Offsets into the LDS copy, as varscan_line_lds.
__device__ __forceinline__ void varscan_core_lds(const uint32_t *lds32, uint32_t p0, uint32_t ref_at, uint32_t depth, uint32_t b0, uint32_t b1,
                                                 uint32_t q0, uint32_t q1, uint64_t zero, const snpgpu_varscan_params &prm, snpgpu_varscan_site *out,
                                                 uint32_t capacity, uint32_t *out_n) {
    auto byte_at = [&](uint32_t p) -> uint32_t { return (lds32[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu; };
    if (depth < prm.min_coverage) return;
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
    uint32_t ref = byte_at(ref_at);
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
        if (reads2 < prm.min_reads2 || (uint64_t)aq < (uint64_t)prm.min_avg_qual * reads2) continue;      // (aq / reads2 < min, without the division)
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

// One line out of the LDS copy (offsets into it; lines there are shorter than 32 KiB, so 16-bit fields hold any count).
// Same results as varscan_line: TABs and the quality test on four bytes at a time, the read-base walk byte by byte out
// of a cached word, counters packed so that no register array is indexed at run time.
struct LineCols { uint32_t ref_at, depth, b0, b1, q0, q1; };     // where the columns of a well-formed line are
// Returns false for an empty line and for a malformed one (reported in *status); else the line's columns.
__device__ __forceinline__ bool varscan_parse_lds(const uint32_t *lds32, uint32_t p0, uint32_t end, uint64_t zero, unsigned long long *status, LineCols &cols) {
    auto byte_at = [&](uint32_t p) -> uint32_t { return (lds32[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu; };
    while (end > p0) { const uint32_t c = byte_at(end - 1); if (c != 10u && c != 13u) break; --end; }
    if (end == p0) return false;
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
    bool ok = nt == 6 && t0 > p0 && t1 > t0 + 1 && t2 == t1 + 2 && t3 > t2 + 1;     // (read bases and qualities may be empty: varscan_line)
    if (ok && t5 == t4 + 1) {
        ok = false;
        for (uint32_t p = t5 + 1; p < end && !ok; ++p) ok = byte_at(p) != 9u;
    }
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
        return false;
    }
    cols = LineCols{t1 + 1, depth, t3 + 1, t4, t4 + 1, t5};
    return true;
}

// ---- two passes: select, then walk ------------------------------------------------------------------------------------------
// Four lines in five of a real pileup show nothing but reference matches: no allele can reach min-reads2 there, and the
// byte-wise walk over their read bases — 90 % of this step's instructions — computes nothing that is used.  So the step runs
// as two kernels.  k_varscan_select looks at every line the cheap way: the first four TABs and the depth, then — the quality
// column of a well-formed line being exactly `depth` bytes long — the fifth TAB where it must be, and ONE pass of word-wide
// "is any byte ..." tests over the two long columns: no further TAB (so the line has exactly six columns, as the walk would
// find), and any of ACGTacgt among the read bases.  A line without such a letter is done (it passed the format checks and can
// call nothing); a line with one goes on the candidate list; a line whose shape the shortcut cannot vouch for (fewer TABs, a
// quality column of another length, more columns) takes the complete walk right there, as before.  k_varscan_walk then gives
// every lane one candidate: the lane copies its line into its own strip of LDS and runs the same walk as before over it.
// Letters that are not read bases (inside an indel, after '^') only cost a walk, never a record.

// nonzero iff some byte of w is zero (exact as a yes/no test)
__device__ __forceinline__ uint32_t any_zero_byte(uint32_t w) { return (w - 0x01010101u) & ~w & 0x80808080u; }
__device__ __forceinline__ uint32_t any_byte_eq(uint32_t w, uint32_t c4) { return any_zero_byte(w ^ c4); }

// One line for k_varscan_select: [p0, end) in the LDS copy.  Returns whether the line goes on the list, and its entry's extras.
__device__ __forceinline__ bool select_line(const uint32_t *lds32, const uint32_t p0, uint32_t end, const snpgpu_varscan_params &prm, uint64_t &entry) {
    bool is_cand = false;
                auto byte_at = [&](uint32_t p) -> uint32_t { return (lds32[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu; };
        while (end > p0) { const uint32_t c = byte_at(end - 1); if (c != 10u && c != 13u) break; --end; }
        bool plain = false;                                                     // the shortcut vouches for the line's shape
        if (end > p0) {
            // the first four TABs out of a bit mask over the 32 bytes from the line's first word on (chrom, position, reference
            // and depth columns of a usual line end well inside; a longer prefix leaves the line to the walk)
            const uint32_t w0 = p0 >> 2, sh0 = p0 & 3u;
            uint32_t M = 0;
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)
                M |= (__builtin_amdgcn_udot4(eq4(lds32[w0 + q], 0x09090909u), 0x08040201u, 0u, false) >> 7) << (4u * q);
            M &= 0xFFFFFFFFu << sh0;
            if (end - w0 * 4u < 32u) M &= (1u << (end - w0 * 4u)) - 1u;
            if (__popc(M) >= 4) {
                const uint32_t t0 = w0 * 4u + (uint32_t)__ffs((int)M) - 1u; M &= M - 1u;
                const uint32_t t1 = w0 * 4u + (uint32_t)__ffs((int)M) - 1u; M &= M - 1u;
                const uint32_t t2 = w0 * 4u + (uint32_t)__ffs((int)M) - 1u; M &= M - 1u;
                const uint32_t t3 = w0 * 4u + (uint32_t)__ffs((int)M) - 1u;
                bool ok = t0 > p0 && t1 > t0 + 1 && t2 == t1 + 2 && t3 > t2 + 1 && t3 - t2 - 1 <= 9;
                uint32_t depth = 0;
                if (ok)
                    for (uint32_t p = t2 + 1; p < t3; ++p) {
                        const uint32_t c = byte_at(p);
                        if (!is_digit(c)) ok = false;
                        depth = depth * 10u + (c - 0x30u);
                    }
                const uint32_t b0 = t3 + 1;
                // six columns with a quality column of `depth` bytes: the fifth TAB sits depth + 1 bytes before the end
                if (ok && depth >= 1 && (uint64_t)b0 + 1 + depth < end) {
                    const uint32_t t4 = end - depth - 1;
                    if (byte_at(t4) == 9u) {
                        // a further TAB in [b0, t4) or (t4, end)?  a read-base letter in [b0, t4)?  Whole words in the
                        // loops, the first and last word of each column with the bytes outside it replaced by '.'
                        uint32_t tabs = 0, letters = 0;
                        auto keep = [](uint32_t v, uint32_t lo, uint32_t hi) -> uint32_t {       // bytes [lo, hi) of the word, hi <= 4
                            const uint32_t m = (0xFFFFFFFFu << (lo * 8u)) & (hi >= 4u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (hi * 8u)));
                            return (v & m) | (0x2E2E2E2Eu & ~m);
                        };
                        auto bases_word = [&](uint32_t v) {
                            tabs |= any_byte_eq(v, 0x09090909u);
                            const uint32_t x = v & 0xDFDFDFDFu;
                            letters |= any_byte_eq(x, 0x41414141u) | any_byte_eq(x, 0x43434343u) | any_byte_eq(x, 0x47474747u) | any_byte_eq(x, 0x54545454u);
                        };
                        {
                            const uint32_t wf = b0 >> 2, wl = (t4 - 1u) >> 2;
                            if (wf == wl) bases_word(keep(lds32[wf], b0 & 3u, ((t4 - 1u) & 3u) + 1u));
                            else {
                                bases_word(keep(lds32[wf], b0 & 3u, 4u));
                                for (uint32_t w = wf + 1u; w < wl; ++w) bases_word(lds32[w]);
                                bases_word(keep(lds32[wl], 0u, ((t4 - 1u) & 3u) + 1u));
                            }
                        }
                        {
                            const uint32_t q0 = t4 + 1u, wf = q0 >> 2, wl = (end - 1u) >> 2;
                            if (wf == wl) tabs |= any_byte_eq(keep(lds32[wf], q0 & 3u, ((end - 1u) & 3u) + 1u), 0x09090909u);
                            else {
                                tabs |= any_byte_eq(keep(lds32[wf], q0 & 3u, 4u), 0x09090909u);
                                for (uint32_t w = wf + 1u; w < wl; ++w) tabs |= any_byte_eq(lds32[w], 0x09090909u);
                                tabs |= any_byte_eq(keep(lds32[wl], 0u, ((end - 1u) & 3u) + 1u), 0x09090909u);
                            }
                        }
                        if (!tabs) {
                            plain = true;
                            is_cand = depth >= prm.min_coverage && letters != 0;
                            if (depth < (1u << 20))
                                entry |= VS_ENTRY_PLAIN | ((uint64_t)depth << 32) | ((uint64_t)(t1 - p0) << 52) | ((uint64_t)(t3 - p0) << 57);
                        }
                    }
                }
            }
        }
        if (!plain && end > p0) is_cand = true;                                 // the walk looks at it in full (format errors included)
    return is_cand;
}

// One wave per workgroup; a wave takes groups of 64 consecutive lines.  A group's bytes are one contiguous span of the file:
// it is fetched with 16-byte loads (coalesced; every byte of the file crosses HBM once) into REGISTERS one round ahead — the
// line offsets that say where it is, two rounds ahead — and written to LDS when the round before is done, so that the loads
// of the next group are in flight while this one's lines are looked at (without that the kernel sits out two dependent trips
// to memory per group and runs at a third of its arithmetic).  kChunks: 16-byte pieces per lane a span may have (LDS bytes /
// 1024); a longer span (a group of very long lines) goes to the walk as it is.
template <int kChunks>
__global__ __launch_bounds__(64) void k_varscan_select(const uint8_t *__restrict__ buf, uint64_t nbytes, const uint64_t *__restrict__ line_off,
                                                       uint64_t n_lines, snpgpu_varscan_params prm, uint64_t *cand, uint32_t *cand_n) {
    // (a workgroup is ONE wave: its LDS operations execute in order, so no s_barrier is needed between writing the span and
    // reading it — and __syncthreads() would also wait for the loads that are meant to stay in flight)
    extern __shared__ uint4 vs_lds[];
    const uint32_t *lds32 = (const uint32_t *)vs_lds;
    constexpr uint32_t lds_bytes = kChunks * 1024;
    // candidates collect in LDS (behind the span) and leave with ONE atomic per VS_CAND_LOCAL of them: an atomic per wave of
    // 64 lines on one address (~12 ns each, 78 000 waves per 5 Mbp sample) would cost more than everything else in this kernel
    uint64_t *cand_local = (uint64_t *)((char *)vs_lds + lds_bytes);
    uint32_t n_local = 0;
    auto flush = [&]() {
        if (n_local == 0) return;
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(cand_n, n_local);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t k = threadIdx.x; k < n_local; k += 64) cand[base + k] = cand_local[k];
        n_local = 0;
    };
    const uint64_t n_groups = (n_lines + 63) / 64;
    const uint32_t lane = threadIdx.x;
    // where the lines of a group start: lane l holds the offset (+1) of line first + l, every lane that of the line after the group
    struct Meta { uint64_t lo, next; };
    auto load_meta = [&](uint64_t grp) -> Meta {
        Meta m{0, 0};
        if (grp < n_groups) {
            const uint64_t first = grp * 64;
            m.lo = first + lane < n_lines ? line_off[first + lane] : nbytes + 1;
            m.next = first + 64 < n_lines ? line_off[first + 64] : nbytes + 1;
        }
        return m;
    };
    struct Span { uint64_t a0; uint32_t n16; bool staged; };
    auto span_of = [&](uint64_t grp, const Meta &m) -> Span {
        Span sp{0, 0, false};
        if (grp < n_groups) {
            const uint64_t s0 = __shfl(m.lo, 0) - 1, s1 = m.next - 1;
            const uint64_t a0 = ((uintptr_t)buf + s0) & ~(uint64_t)15, a1 = (((uintptr_t)buf + s1) + 15) & ~(uint64_t)15;
            sp.a0 = a0;
            sp.staged = a1 - a0 <= lds_bytes;
            sp.n16 = sp.staged ? (uint32_t)((a1 - a0) / 16) : 0;
        }
        return sp;
    };
    // the span in flight: one named register quadruple per chunk (an array indexed in a loop ends up in scratch memory); every
    // lane loads every round, from a clamped index (a0 of a group past the end is the file's first chunk)
#define VS_EACH(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
#define VS_DECL(k) uint4 R##k = make_uint4(0, 0, 0, 0);
    VS_EACH(VS_DECL)
#undef VS_DECL
#define VS_LOAD(k) if constexpr (kChunks > k) { const uint32_t c_ = k * 64u + lane; R##k = src_[c_ < top_ ? c_ : top_]; }
#define VS_FETCH(sp)                                                                          \
    {                                                                                         \
        const uint4 *src_ = (const uint4 *)((sp).n16 ? (sp).a0 : ((uintptr_t)buf & ~(uint64_t)15)); \
        const uint32_t top_ = (sp).n16 ? (sp).n16 - 1u : 0u;                                  \
        VS_EACH(VS_LOAD)                                                                      \
    }
#define VS_STORE(k) if constexpr (kChunks > k) { const uint32_t c_ = k * 64u + lane; if (c_ < sp0.n16) vs_lds[c_] = R##k; }
    uint64_t grp = blockIdx.x;
    Meta m0 = load_meta(grp), m1 = load_meta(grp + gridDim.x);
    Span sp0 = span_of(grp, m0);
    VS_FETCH(sp0)
    for (; grp < n_groups; grp += gridDim.x) {
        // this group's bytes: registers -> LDS
        VS_EACH(VS_STORE)
        __builtin_amdgcn_wave_barrier();
        // the next group's bytes and the offsets of the one after it are on their way while this group is looked at
        const Span sp1 = span_of(grp + gridDim.x, m1);
        VS_FETCH(sp1)
        const Meta m2 = load_meta(grp + 2 * (uint64_t)gridDim.x);
        const uint64_t first = grp * 64, last = first + 64 < n_lines ? first + 64 : n_lines;
        const uint64_t line = first + lane;
        bool is_cand = false;
        uint64_t entry = line;                                                          // + what the walk need not find out again (VS_ENTRY_*)
        {
            uint64_t up1 = __shfl_down(m0.lo, 1);                                       // the next line's start
            if (lane == 63 || line + 1 >= last) up1 = m0.next;
            if (line < last) {
                if (!sp0.staged) is_cand = true;                                        // a span of very long lines: all of it to the walk
                else {
                    const uint64_t zero = sp0.a0 - (uintptr_t)buf;                      // file offset of LDS byte 0 (mod 2^64)
                    is_cand = select_line(lds32, (uint32_t)(m0.lo - 1 - zero), (uint32_t)(up1 - 1 - zero), prm, entry);
                }
            }
        }
        const unsigned long long m = __ballot(is_cand);
        if (m) {
            if (is_cand) cand_local[n_local + __popcll(m & ((1ull << lane) - 1ull))] = entry;
            n_local += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        if (n_local + 64 > VS_CAND_LOCAL) { flush(); __builtin_amdgcn_wave_barrier(); }
        m0 = m1; m1 = m2; sp0 = sp1;
    }
    __builtin_amdgcn_wave_barrier();
    flush();
#undef VS_FETCH
#undef VS_LOAD
#undef VS_STORE
#undef VS_EACH
}

// One lane per candidate line: the 64 lines of a wave are copied into LDS back to back (16-byte chunks; a wave prefix sum of
// the chunk counts gives every lane its place) and every lane walks its own there.  What does not fit the wave's LDS (a few
// lines several times the mean length) goes on a second list.  The kernel is a chain of dependent loads per line — list entry,
// line offsets, bytes — so the entries are fetched two rounds ahead and the offsets one round ahead of their use.
__global__ __launch_bounds__(64) void k_varscan_walk(const uint8_t *__restrict__ buf, uint64_t nbytes, const uint64_t *__restrict__ line_off,
                                                     uint64_t n_lines, snpgpu_varscan_params prm, snpgpu_varscan_site *out, uint32_t capacity,
                                                     uint32_t *out_n, unsigned long long *status, uint32_t lds_bytes, const uint64_t *__restrict__ cand,
                                                     const uint32_t *__restrict__ cand_n, uint32_t *long_lines, uint32_t *long_n) {
    extern __shared__ uint4 vs_lds[];
    const uint32_t n = *cand_n;
    const uint64_t stride = (uint64_t)gridDim.x * 64;
    uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t NONE = ~0ull;
    auto entry = [&](uint64_t k) -> uint64_t { return k < n ? cand[k] : NONE; };
    auto bounds = [&](uint64_t e, uint64_t &p0, uint64_t &end) {
        p0 = end = 0;
        const uint32_t ln = (uint32_t)e;
        if (e != NONE) { p0 = line_off[ln] - 1; end = (uint64_t)ln + 1 < n_lines ? line_off[ln + 1] - 1 : nbytes; }
    };
    uint64_t line = entry(i), line1 = entry(i + stride);
    uint64_t p0, end;
    bounds(line, p0, end);
#ifdef SNPGPU_TUNING
    unsigned long long t_copy = 0, t_walk = 0, t_all = __builtin_readcyclecounter();
#endif
    for (uint64_t i0 = (uint64_t)blockIdx.x * 64; i0 < n; i0 += stride, i += stride) {
#ifdef SNPGPU_TUNING
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        const uint64_t line2 = entry(i + 2 * stride);               // two rounds ahead
        uint64_t p0n, endn;
        bounds(line1, p0n, endn);                                    // one round ahead
        const bool have = line != NONE;
        const uint64_t a0 = ((uintptr_t)buf + p0) & ~(uint64_t)15, a1 = have ? (((uintptr_t)buf + end) + 15) & ~(uint64_t)15 : a0;
        const uint32_t chunks = (uint32_t)((a1 - a0) / 16);
        // exclusive prefix sum of the chunk counts over the wave
        uint32_t incl = chunks;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up_ = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up_;
        }
        const uint32_t first_chunk = incl - chunks;
        const bool fits = (uint64_t)incl * 16 <= lds_bytes;
        if (have && fits) {
            uint4 *mine = vs_lds + first_chunk;
            const uint4 *src = (const uint4 *)a0;
            for (uint32_t c = 0; c < chunks; ++c) mine[c] = src[c];
#ifdef SNPGPU_TUNING
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (threadIdx.x == 0) t_copy += __builtin_readcyclecounter() - c0;
#endif
            const uint32_t lane0 = first_chunk * 16u;                                   // LDS offset of this lane's bytes
            const uint64_t a0_off = a0 - (uintptr_t)buf;                                // their file offset
            const uint32_t *lds32 = (const uint32_t *)vs_lds;
            const uint32_t l0 = lane0 + (uint32_t)(p0 - a0_off);
            uint32_t l1 = lane0 + (uint32_t)(end - a0_off);
            LineCols cols;
            bool ok;
            if (line & VS_ENTRY_PLAIN) {
                // the columns are where k_varscan_select found them
                while (l1 > l0) { const uint32_t c = (lds32[(l1 - 1u) >> 2] >> (((l1 - 1u) & 3u) * 8u)) & 0xFFu; if (c != 10u && c != 13u) break; --l1; }
                const uint32_t depth = (uint32_t)(line >> 32) & 0xFFFFFu, t1 = l0 + ((uint32_t)(line >> 52) & 31u), t3 = l0 + ((uint32_t)(line >> 57) & 31u);
                const uint32_t t4 = l1 - depth - 1u;
                cols = LineCols{t1 + 1u, depth, t3 + 1u, t4, t4 + 1u, l1};
                ok = true;
            } else {
                ok = varscan_parse_lds(lds32, l0, l1, a0_off - lane0, status, cols);
            }
            if (ok) varscan_core_lds(lds32, l0, cols.ref_at, cols.depth, cols.b0, cols.b1, cols.q0, cols.q1, a0_off - lane0, prm, out, capacity, out_n);
        } else if (have) {
            long_lines[atomicAdd(long_n, 1u)] = (uint32_t)line;                         // (rare)
        }
        line = line1; line1 = line2;
        p0 = p0n; end = endn;
#ifdef SNPGPU_TUNING
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) t_walk += __builtin_readcyclecounter() - c0;
#endif
    }
#ifdef SNPGPU_TUNING
    if (threadIdx.x == 0) {                                                             // [0] cycles until the lines were in LDS, [1] whole rounds, [2] whole kernel (summed over the waves)
        unsigned long long *dbg = (unsigned long long *)(long_n + 2);
        atomicAdd(dbg, t_copy); atomicAdd(dbg + 1, t_walk); atomicAdd(dbg + 2, __builtin_readcyclecounter() - t_all);
    }
#endif
}

// The candidates that did not fit a strip: one lane per line, bytes straight from global memory.
__global__ __launch_bounds__(64) void k_varscan_walk_long(const uint8_t *__restrict__ buf, uint64_t nbytes, const uint64_t *__restrict__ line_off,
                                                          uint64_t n_lines, snpgpu_varscan_params prm, snpgpu_varscan_site *out, uint32_t capacity,
                                                          uint32_t *out_n, unsigned long long *status, const uint32_t *__restrict__ long_lines,
                                                          const uint32_t *__restrict__ long_n) {
    const uint32_t n = *long_n;
    for (uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 64) {
        const uint64_t line = long_lines[i];
        const uint64_t p0 = line_off[line] - 1, end = line + 1 < n_lines ? line_off[line + 1] - 1 : nbytes;
        varscan_line<uint64_t>(GlobalBytes{buf}, p0, end, 0, prm, out, capacity, out_n, status);
    }
}

}  // namespace

// d_n: FOUR zeroed words — [0] records found, [1] candidate lines, [2] candidates too long for a strip (scratch of the passes);
// d_status: one u64 preset to UINT64_MAX (becomes the offset of the first malformed line); d_cand: 2 * n_lines words of scratch
int snpgpu_enqueue_varscan(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, const uint64_t *d_line_off, uint64_t n_lines,
                           const snpgpu_varscan_params *prm, snpgpu_varscan_site *d_sites, uint32_t capacity, uint32_t *d_n, uint64_t *d_status,
                           uint32_t *d_cand) {
    if (n_lines == 0) return SNPGPU_OK;
    // select: LDS per wave = the span of 64 lines of the file's mean length with a quarter to spare, in KiB steps up to 32 KiB (a
    // span that does not fit goes to the walk as it is) + the local candidate list; a CU's 160 KiB then hold 160 / that many waves
    const uint64_t mean = nbytes / n_lines + 1;
    uint64_t want_lds = mean * 64 * 5 / 4 + 64;
    const uint32_t lds = want_lds <= 4096 ? 4096 : want_lds <= 6144 ? 6144 : want_lds <= 8192 ? 8192 : want_lds <= 12288 ? 12288 : want_lds <= 16384 ? 16384 : 24576;
    const uint32_t sel_lds = lds + 8 * VS_CAND_LOCAL;
    const uint32_t waves_per_cu = 160 * 1024 / sel_lds < 32 ? 160 * 1024 / sel_lds : 32;
    const uint64_t groups = (n_lines + 63) / 64;
    const uint64_t cap = (uint64_t)ctx->n_cu * waves_per_cu * 2;
    const unsigned grid = (unsigned)(groups < cap ? groups : cap);
    uint64_t *d_list = (uint64_t *)d_cand;
#define VS_SELECT(K) k_varscan_select<K><<<grid, 64, sel_lds, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_list, d_n + 1)
    if (lds == 4096) VS_SELECT(4); else if (lds == 6144) VS_SELECT(6); else if (lds == 8192) VS_SELECT(8); else if (lds == 12288) VS_SELECT(12);
    else if (lds == 16384) VS_SELECT(16); else VS_SELECT(24);
#undef VS_SELECT
    // walk: LDS for 64 lines of 1.3 x the mean length + the alignment slack of each (candidates are the deeper lines), 2 .. 60 KiB
    uint64_t want = (uint64_t)64 * (mean * 13 / 10 + 40);
    want = (want + 1023) / 1024 * 1024;
    const uint32_t walk_bytes = (uint32_t)(want < 2048 ? 2048 : (want > 60 * 1024 ? 60 * 1024 : want));
    const uint32_t walk_lds = walk_bytes + 16;                   // (+ one chunk: the walk may read the word after a line)
    const uint32_t walk_waves_per_cu = 160 * 1024 / walk_lds < 16 ? 160 * 1024 / walk_lds : 16;       // (its registers allow 4 per SIMD)
    const unsigned walk_grid = (unsigned)((uint64_t)ctx->n_cu * (walk_waves_per_cu ? walk_waves_per_cu : 1));
    uint32_t *d_long = d_cand + 2 * n_lines;
    k_varscan_walk<<<walk_grid, 64, walk_lds, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n, (unsigned long long *)d_status,
                                                              walk_bytes, d_list, d_n + 1, d_long, d_n + 2);
    k_varscan_walk_long<<<ctx->n_cu, 64, 0, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, *prm, d_sites, capacity, d_n, (unsigned long long *)d_status, d_long,
                                                           d_n + 2);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}
