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
#include "prims.h"

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

// ---- two passes over what matters: scan + select every line, then walk the few that can call something ------------------------
// Nearly every line of a real pileup cannot reach min-reads2 for any allele, and the byte-wise walk over its read bases — 90 % of
// this step's instructions in round 2 — computes nothing that is used.  So the step runs as two kernels, and since round 4 the
// first of them needs NO line index and has no loop over the bytes of a line: k_varscan_scan reads the text once, in tiles, and
// per tile
//   A1  classifies every byte, 16 at a time, into three bit strings over the tile: line terminators, TABs, and "letters" — bytes
//       with bit 6 set and bit 3 clear.  Every read-base letter (ACGTacgt: 0x41 43 47 54 and 0x20 more) is such a byte; N, n and
//       '^' are not (bit 3), '$', digits, '.', ',' are not (bit 6); a mapping-quality character after '^' or a letter inside an
//       indel may be, which only makes a count of them an upper bound of any allele's reads;
//   A2  turns the terminator string into the list of line starts (a wave prefix sum) and the other two into running counts per
//       32 bytes;
//   B   gives every line one lane, which answers in constant time, from the bit strings alone: where are the first four TABs
//       (find-first-set on 64 bits of the TAB string), the depth (one unaligned load, SWAR digits), is the fifth TAB where a
//       quality column of exactly `depth` bytes puts it, are there five TABs in all (difference of two running counts), how many
//       letters has the read-base column (another difference).
// A line whose shape checks out ("plain") with fewer letters than min-reads2 cannot call anything and is done; with at least that
// many it goes on the candidate list; a line the shortcut cannot vouch for goes on the list as it is and the walk parses it in
// full (format errors included).  k_varscan_walk then gives every lane one candidate: the lane copies its line into its own strip
// of LDS and runs the exact automaton over it.  Round 3 looked at every byte of every line in a lane-per-line loop (a third of the
// lanes idle at 100x) and put every line with ANY letter on the list — 14 % of the lines at 30x, 39 % at 100x, which made the
// walk grow 6.4 x for 2.6 x the bytes; now it is the variant sites and little else, at any depth.

// A list entry (16 bytes): x, y = file offset of the line's first byte; z = its length in bytes (terminator included; 0 for a
// line that did not end inside the tile's LDS window: VS_W_LONG, the walk finds its end itself); w = what the walk need not find
// out again for a line whose shape the scan has checked (VS_W_PLAIN): depth (bits 0-13), and where, counted from the line's
// first byte, its second and fourth TAB are (bits 14-19, 20-25)
constexpr uint32_t VS_W_PLAIN = 1u << 31, VS_W_LONG = 1u << 30;

// A candidate straight from global memory, byte by byte (the generic form of the walk; also what a full list falls back on).
__device__ __noinline__ void walk_entry_global(const uint8_t *buf, uint64_t nbytes, uint4 e, const snpgpu_varscan_params &prm, snpgpu_varscan_site *out,
                                               uint32_t capacity, uint32_t *out_n, unsigned long long *status) {
    const uint64_t p0 = (uint64_t)e.x | ((uint64_t)e.y << 32);
    uint64_t end = p0 + e.z;
    if (e.w & VS_W_LONG) {                                                              // ends at the first line terminator
        end = p0;
        while (end < nbytes && buf[end] != 10u && buf[end] != 13u) ++end;
    }
    varscan_line<uint64_t>(GlobalBytes{buf}, p0, end < nbytes ? end : nbytes, 0, prm, out, capacity, out_n, status);
}

#define VS_LIST_CAP 256u          // line starts held in LDS per pass (a tile with more makes extra passes)
#define VS_CAND_LOCAL 96u         // candidate entries a wave collects in LDS before it takes a place on the list

// A1 for one chunk: the 16 bytes of v -> 16 bits of each string.  The SWAR tests run on 8 bytes at a time (one v_lshl_add_u64 per
// add), and two dwords' flags are gathered into one register by a second v_dot4 that accumulates with weights 16 times the first's.
__device__ __forceinline__ uint32_t gather16(uint64_t f_lo, uint64_t f_hi, uint32_t shift) {
    // f: flag bit `shift` of every byte (7, or 6 for the letters); byte k of the chunk -> bit k
    uint32_t a = __builtin_amdgcn_udot4((uint32_t)f_lo, 0x08040201u, 0u, false);
    a = __builtin_amdgcn_udot4((uint32_t)(f_lo >> 32), 0x80402010u, a, false);
    uint32_t b = __builtin_amdgcn_udot4((uint32_t)f_hi, 0x08040201u, 0u, false);
    b = __builtin_amdgcn_udot4((uint32_t)(f_hi >> 32), 0x80402010u, b, false);
    return ((a >> shift) & 0xFFu) | (((b >> shift) & 0xFFu) << 8);
}
template <bool kExact>
__device__ __forceinline__ void classify_chunk(const uint4 v, uint32_t c, uint16_t *nl, uint16_t *cr, uint16_t *tab, uint16_t *let) {
    constexpr uint64_t k7F = 0x7F7F7F7F7F7F7F7Full, k80 = 0x8080808080808080ull;
    const uint64_t q[2] = {(uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32)};
    uint64_t ft[2], fl[2], fn[2], fc[2];
    auto eq8 = [&](uint64_t w, uint64_t c8) -> uint64_t {      // 0x80 in every byte of w equal to the byte replicated in c8 (exact)
        const uint64_t x = w ^ c8;
        return ~(((x & k7F) + k7F) | x) & k80;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t w = q[h];
        ft[h] = eq8(w, 0x0909090909090909ull);
        fl[h] = w & ~(w << 3) & 0x4040404040404040ull;           // bit 6 set, bit 3 clear (what crosses a byte in the shift is masked)
        if (!kExact) {
            const uint64_t m = w & k7F;                            // the adds run on seven bits: nothing carries between bytes
            fn[h] = (m + 0x7676767676767676ull) & ~(m + 0x7272727272727272ull) & ~w & k80;
            fc[h] = 0;
        } else {
            fn[h] = eq8(w, 0x0A0A0A0A0A0A0A0Aull);
            fc[h] = eq8(w, 0x0D0D0D0D0D0D0D0Dull);
        }
    }
    nl[c] = (uint16_t)gather16(fn[0], fn[1], 7);
    tab[c] = (uint16_t)gather16(ft[0], ft[1], 7);
    let[c] = (uint16_t)gather16(fl[0], fl[1], 6);
    if (kExact) cr[c] = (uint16_t)gather16(fc[0], fc[1], 7);
}

// One wave per workgroup, every wave one contiguous run of tiles of the file.  A tile's slot in LDS holds the bytes
// [t0 - 16, t0 + tile + halo) = 64 * kCPL chunks of 16 bytes: tile k+1 streams into the other slot with LDS-DMA
// (global_load_lds_dwordx4, kCPL wave instructions, no register round trip) while tile k is looked at; the wave waits for its own
// requests with a counted s_waitcnt, nothing else waits for anything.  A line that starts in the tile lies completely in the slot
// when it is shorter than the halo; one that does not is left to the walk (VS_W_LONG).  The halo is fetched again with the next
// tile (mostly from L2).  Geometries (chunks per lane, tile, halo): 4, 3840, 240 for lines of ~100 bytes (30x); 6, 5120, 1008 up
// to ~480 bytes; 8, 4096, 4080 beyond.  Coordinates are "aligned": byte a of abase = file byte a - lo, abase 16-byte aligned, the
// file at [lo, hi).
// Line terminators are Java's readLine(): LF, CR, CR LF.  The fast form of A1 flags bytes 0x0A..0x0D with two adds and takes every
// one for LF; phase B looks at the byte that ends each line, and the first that is not LF switches the wave to the exact form
// (separate LF and CR strings, starts after LF, or after a CR that no LF follows) for this and all its later tiles.
template <int kCPL>
__global__ __launch_bounds__(1024) void k_varscan_scan(const uint8_t *__restrict__ abase, uint64_t lo, uint64_t hi, uint32_t tile_bytes, uint64_t n_tiles,
                                                       snpgpu_varscan_params prm, uint4 *cand, uint32_t cand_cap, uint32_t *ctl, unsigned long long *status,
                                                       snpgpu_varscan_site *out, uint32_t capacity, uint32_t *wave_lines, uint32_t lds_chunks_per_wave, uint4 share) {
    constexpr uint32_t kChunks = 64u * kCPL;                                            // 16-byte chunks of a slot
    constexpr uint32_t kSlotBytes = kChunks * 16u;
    constexpr uint32_t kW = kSlotBytes / 32u;                                           // words of a bit string over the slot
    constexpr uint32_t kMW = kW / 64u;                                                  // ... per lane in A2
    extern __shared__ uint4 vs_lds_all[];
    // the waves of a workgroup are independent (no barrier anywhere): each has its own stretch of the workgroup's LDS
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 *vs_lds = vs_lds_all + (size_t)wave_in_wg * lds_chunks_per_wave;
    const uint32_t gwave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg, n_gwaves = gridDim.x * (blockDim.x >> 6);
    uint4 *slot0 = vs_lds, *slot1 = vs_lds + kChunks;
    uint32_t *nlbits = (uint32_t *)(vs_lds + 2 * kChunks);                              // terminators (exact form: LF)
    uint32_t *crbits = nlbits + kW + 4, *tabbits = crbits + kW + 4, *letbits = tabbits + kW + 4, *pre = letbits + kW + 4;   // (+4: the window reads run two words over)
    uint4 *cand_local = (uint4 *)(pre + kW + 4);
    uint16_t *lstart = (uint16_t *)(cand_local + VS_CAND_LOCAL);                        // VS_LIST_CAP + 2 entries
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t *fbuf = abase + lo;                                                   // file byte 0
    const uint64_t nbytes = hi - lo;
    uint32_t n_local = 0, lines_seen = 0;
    bool exact = false;                                                                 // wave-uniform: the exact form of A1 / A2
    if (lane < 8) { nlbits[kW + (lane & 3)] = 0; crbits[kW + (lane & 3)] = 0; tabbits[kW + (lane & 3)] = 0; letbits[kW + (lane & 3)] = 0; pre[kW + (lane & 3)] = 0; }
    auto flush = [&]() {
        if (n_local == 0) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ctl + 1, n_local);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t k = lane; k < n_local; k += 64) {
            const uint4 e = cand_local[k];
            if (base + k < cand_cap) cand[base + k] = e;
            else walk_entry_global(fbuf, nbytes, e, prm, out, capacity, ctl, status);  // the list is full: looked at on the spot
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // (the requests in flight are counted from zero again)
        n_local = 0;
    };
    // my run of tiles
    // ... in proportion to its share: wave w of a workgroup sits on SIMD w % 4 and is the (w / 4)-th oldest there; the SIMD issues
    // oldest first, so with equal shares the oldest wave finishes early and the youngest runs on alone (share.x: oldest)
    uint64_t t_first, t_end;
    {
        const uint32_t wpb = blockDim.x >> 6, sh[4] = {share.x, share.y, share.z, share.w};
        uint64_t blk = 0, before = 0;
        for (uint32_t w = 0; w < wpb; ++w) { const uint32_t x = sh[(w >> 2) & 3u]; blk += x; before += w < wave_in_wg ? x : 0u; }
        const uint64_t total = blk * gridDim.x, c0 = (uint64_t)blockIdx.x * blk + before, c1 = c0 + sh[(wave_in_wg >> 2) & 3u];
        t_first = (uint64_t)((unsigned __int128)n_tiles * c0 / total);
        t_end = (uint64_t)((unsigned __int128)n_tiles * c1 / total);
    }
    auto interior = [&](uint64_t tt) { const uint64_t x0 = tt * tile_bytes; return x0 >= lo + 16 && x0 - 16 + kSlotBytes <= hi; };
    // request tile tt into `slot`; returns whether it travels by DMA (else it has been staged synchronously)
    auto request = [&](uint64_t tt, uint4 *slot) -> bool {
        if (interior(tt)) {
            const uint8_t *gs = abase + tt * tile_bytes - 16;
            const uint32_t voff = lane * 16u;
            const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)slot);
#pragma unroll
            for (uint32_t r = 0; r < (uint32_t)kCPL; ++r) {
                const uint64_t ga = (uint64_t)(uintptr_t)gs + (r >> 2) * 4096u;
                const uint64_t gr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ga >> 32)) << 32) |
                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ga);
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (r >> 2) * 4096u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(gr), "s"(m0v), "n"((r & 3u) * 1024u) : "memory");
            }
            return true;
        }
        // first / last tiles of the file: byte loads; the byte before the file reads as '\n' (byte 0 starts a line), the other
        // bytes outside it as NUL (no line starts past the end)
        const int64_t x0 = (int64_t)(tt * tile_bytes) - 16;
#pragma nounroll
        for (uint32_t e = lane; e < kChunks; e += 64) {
            uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t idx = x0 + (int64_t)e * 16 + 4 * k + j;
                    const uint32_t bb = (idx >= (int64_t)lo && idx < (int64_t)hi) ? (uint32_t)abase[idx] : (idx + 1 == (int64_t)lo ? 10u : 0u);
                    d[k] |= bb << (8 * j);
                }
            slot[e] = make_uint4(d[0], d[1], d[2], d[3]);
        }
        return false;
    };
    // the bits of a 32-bit word of a string (over the slot bytes from `b` on) that lie below slot offset x
    auto below = [](uint32_t x, uint32_t b) -> uint32_t { return x <= b ? 0u : (x - b >= 32u ? 0xFFFFFFFFu : (1u << (x - b)) - 1u); };
    // ... for this lane's words in A2: starts in [16, 16 + tile) are the tile's lines, in [16 + tile, slot end) end its last one
    uint32_t own_mask[kMW], halo_mask[kMW];
#pragma unroll
    for (uint32_t k = 0; k < kMW; ++k) {
        const uint32_t b = (lane * kMW + k) * 32u;
        own_mask[k] = below(16u + tile_bytes, b) & ~below(16u, b);
        halo_mask[k] = below(kSlotBytes, b) & ~below(16u + tile_bytes, b);
    }
    bool dma_next = false;
    if (t_first < t_end) (void)request(t_first, slot0);
    if (t_first + 1 < t_end) dma_next = request(t_first + 1, slot1);
    uint4 *cur = slot0, *other = slot1;
    for (uint64_t tt = t_first; tt < t_end; ++tt) {
        // the current tile's request has landed when only the next tile's is outstanding
        if (dma_next && tt + 1 < t_end) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kCPL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint32_t *lds32 = (const uint32_t *)cur;
        const uint64_t t0 = tt * tile_bytes;                                            // aligned coordinate of slot byte 16
        // slot offset of the file's end (what a line without a terminator runs to), when that is inside the slot
        const uint64_t hi_rel = hi - (t0 - 16);
        const uint32_t hi_slot = hi_rel < kSlotBytes ? (uint32_t)hi_rel : 0xFFFFFFFFu;
        const uint32_t own_end = 16u + tile_bytes < hi_slot ? 16u + tile_bytes : hi_slot;   // starts in [16, own_end) are this tile's lines
        const uint32_t halo_end = hi_slot < kSlotBytes ? hi_slot : kSlotBytes;              // ... in [own_end, halo_end) end its last one
        bool redo;
        do {
            redo = false;
            __builtin_amdgcn_wave_barrier();
            // ---- A1: chunk i * 64 + lane (conflict-free ds_read_b128) -> 16 bits of each string ------------------------------
            if (!exact) {
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)kCPL; ++i)
                    classify_chunk<false>(cur[i * 64u + lane], i * 64u + lane, (uint16_t *)nlbits, (uint16_t *)crbits, (uint16_t *)tabbits, (uint16_t *)letbits);
            } else {
#pragma unroll
                for (uint32_t i = 0; i < (uint32_t)kCPL; ++i)
                    classify_chunk<true>(cur[i * 64u + lane], i * 64u + lane, (uint16_t *)nlbits, (uint16_t *)crbits, (uint16_t *)tabbits, (uint16_t *)letbits);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- A2: words lane * kMW .. of the strings: line starts, running counts ----------------------------------------
            uint32_t S[kMW], tw[kMW], lw[kMW];
            uint32_t c_nl = 0, c_tab = 0, c_let = 0, halo_first = 0xFFFFFFFFu;
            {
                const uint32_t w0 = lane * kMW;
                uint32_t carry_n = w0 ? nlbits[w0 - 1] >> 31 : 0u, carry_c = (exact && w0) ? crbits[w0 - 1] >> 31 : 0u;
#pragma unroll
                for (uint32_t k = 0; k < kMW; ++k) {
                    const uint32_t n = nlbits[w0 + k];
                    tw[k] = tabbits[w0 + k];
                    lw[k] = letbits[w0 + k];
                    uint32_t st = (n << 1) | carry_n;
                    carry_n = n >> 31;
                    if (exact) { const uint32_t cr = crbits[w0 + k]; st |= ((cr << 1) | carry_c) & ~n; carry_c = cr >> 31; }
                    // bits of this word inside [16, own_end) / [own_end, halo_end)
                    const uint32_t b = (w0 + k) * 32u;
                    uint32_t m_own = own_mask[k], m_halo = halo_mask[k];
                    if (hi_slot != 0xFFFFFFFFu) {                                       // (a tile the file ends in: its own limits)
                        m_own = below(own_end, b) & ~below(16u, b);
                        m_halo = below(halo_end, b) & ~below(own_end, b);
                    }
                    const uint32_t halo = st & m_halo;
                    if (halo && halo_first == 0xFFFFFFFFu) halo_first = b + (uint32_t)__ffs((int)halo) - 1u;
                    S[k] = st & m_own;
                    c_nl += (uint32_t)__popc(S[k]);
                    c_tab += (uint32_t)__popc(tw[k]);
                    c_let += (uint32_t)__popc(lw[k]);
                }
            }
            const uint32_t packed = c_nl | (c_tab << 13);
            const uint32_t i1 = wave_inclusive_sum(packed), i2 = wave_inclusive_sum(c_let);
            const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63) & 0x1FFFu;
            uint32_t idx0 = (i1 - packed) & 0x1FFFu;
            {
                uint32_t ptab = (i1 - packed) >> 13, plet = i2 - c_let;
#pragma unroll
                for (uint32_t k = 0; k < kMW; ++k) {
                    pre[lane * kMW + k] = (ptab & 0xFFFFu) | (plet << 16);
                    ptab += (uint32_t)__popc(tw[k]);
                    plet += (uint32_t)__popc(lw[k]);
                }
            }
            // where the line after the tile's last one starts: the first start in the halo, else the end of the file, else unknown
            uint32_t next = hi_slot;                                                    // 0xFFFFFFFF: not in the slot
            {
                const unsigned long long any = __ballot(halo_first != 0xFFFFFFFFu);
                if (any) next = (uint32_t)__builtin_amdgcn_readlane((int)halo_first, (int)(__ffsll((long long)any) - 1));
            }
            // ---- passes of up to VS_LIST_CAP lines: the list, then one lane per line ---------------------------------------
#pragma unroll 1
            for (uint32_t base = 0; base < T; base += VS_LIST_CAP) {
                {
                    // The fast form took every byte in 0x0A..0x0D for LF: on the way through the starts (first pass), is the byte in
                    // front of each one?  The first that is not — CR, VT, FF — switches the wave to the exact form, before anything
                    // of this tile has been used.
                    const bool check = !exact && base == 0;
                    bool odd = check && halo_first != 0xFFFFFFFFu && halo_first == next && ((lds32[(next - 1u) >> 2] >> (((next - 1u) & 3u) * 8u)) & 0xFFu) != 10u;
                    uint32_t idx = idx0;
#pragma unroll
                    for (uint32_t k = 0; k < kMW; ++k) {
                        uint32_t mm = S[k];
                        const uint32_t b = (lane * kMW + k) * 32u;
                        while (mm) {
                            const uint32_t q = b + (uint32_t)__ffs((int)mm) - 1u;
                            mm &= mm - 1u;
                            if (idx - base <= VS_LIST_CAP) lstart[idx - base] = (uint16_t)q;            // (unsigned: also false for idx < base)
                            if (check) odd |= ((lds32[(q - 1u) >> 2] >> (((q - 1u) & 3u) * 8u)) & 0xFFu) != 10u;
                            ++idx;
                        }
                    }
                    if (check && __ballot(odd)) { exact = true; redo = true; break; }
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t n_here = T - base < VS_LIST_CAP ? T - base : VS_LIST_CAP;
#pragma unroll 1
                for (uint32_t r = 0; r < n_here; r += 64) {
                    const uint32_t i = r + lane;
                    bool is_cand = false;
                    uint4 e = make_uint4(0, 0, 0, 0);
                    if (i < n_here) {
                        const uint32_t p0 = lstart[i];
                        const uint32_t end = base + i + 1u < T ? (uint32_t)lstart[i + 1u] : next;
                        const uint64_t off = t0 - 16 + p0 - lo;                         // file offset of the line
                        e.x = (uint32_t)off; e.y = (uint32_t)(off >> 32);
                        if (end == 0xFFFFFFFFu) { is_cand = true; e.w = VS_W_LONG; }    // runs past the slot: the walk finds its end
                        else {
                            e.z = end - p0;
                            auto byte_at = [&](uint32_t p) -> uint32_t { return (lds32[p >> 2] >> ((p & 3u) * 8u)) & 0xFFu; };
                            // the line without its terminator
                            uint32_t le = end;
                            if (!exact) {
                                if (byte_at(end - 1u) == 10u) le = end - 1u;           // (else: the file's last line, without a terminator)
                            } else {
                                while (le > p0) { const uint32_t cb = byte_at(le - 1u); if (cb != 10u && cb != 13u) break; --le; }
                            }
                            if (le > p0) {
                                bool plain = false;
                                // the first four TABs out of the 64 bits of the TAB string from the line's first byte on
                                const uint32_t wq = p0 >> 5, sh = p0 & 31u;
                                const uint32_t a0 = tabbits[wq], a1 = tabbits[wq + 1u];
                                const uint32_t span = le - p0;
                                uint32_t m32 = __builtin_amdgcn_alignbit(a1, a0, sh);
                                if (span < 32u) m32 &= (1u << span) - 1u;
                                uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
                                bool four = false;
                                if (__popc(m32) >= 4) {                                 // the usual case: all four within 32 bytes
                                    four = true;
                                    r0 = (uint32_t)__ffs((int)m32) - 1u; m32 &= m32 - 1u;
                                    r1 = (uint32_t)__ffs((int)m32) - 1u; m32 &= m32 - 1u;
                                    r2 = (uint32_t)__ffs((int)m32) - 1u; m32 &= m32 - 1u;
                                    r3 = (uint32_t)__ffs((int)m32) - 1u;
                                } else if (span > 32u) {                                // long contig names: 64 bytes
                                    const uint32_t a2 = tabbits[wq + 2u];
                                    uint64_t M = (uint64_t)m32 | ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, sh) << 32);
                                    if (span < 64u) M &= (1ull << span) - 1ull;
                                    if (__popcll(M) >= 4) {
                                        four = true;
                                        r0 = (uint32_t)__ffsll((long long)M) - 1u; M &= M - 1ull;
                                        r1 = (uint32_t)__ffsll((long long)M) - 1u; M &= M - 1ull;
                                        r2 = (uint32_t)__ffsll((long long)M) - 1u; M &= M - 1ull;
                                        r3 = (uint32_t)__ffsll((long long)M) - 1u;
                                    }
                                }
                                if (four) {
                                    const uint32_t nd = r3 - r2 - 1u;                   // digits of the depth
                                    if (r0 > 0u && r1 > r0 + 1u && r2 == r1 + 2u && nd >= 1u && nd <= 4u) {
                                        // the depth: four bytes from its first digit on, most significant first
                                        const uint32_t da = p0 + r2 + 1u;
                                        const uint32_t x = __builtin_amdgcn_alignbyte(lds32[(da >> 2) + 1u], lds32[da >> 2], da & 3u);
                                        const uint32_t keep = nd >= 4u ? 0xFFFFFFFFu : (1u << (8u * nd)) - 1u;
                                        const uint32_t z = (x & keep) | (0x30303030u & ~keep);      // the bytes behind the digits read as '0'
                                        // every byte in '0'..'9': bit 7 clear, z + 0x46 below 0x80, z + 0x50 at or above it
                                        const bool digits = ((z | (z + 0x46464646u) | ~(z + 0x50505050u)) & 0x80808080u) == 0u;
                                        const uint32_t ys = (z - 0x30303030u) << (8u * (4u - nd));  // digit k in byte 4 - nd + k: weights 1000, 100, 10, 1 by byte
                                        const uint32_t depth = __builtin_amdgcn_udot4(ys, 0x010A6400u, 0u, false) + (ys & 0xFFu) * 1000u;
                                        const uint32_t b0 = p0 + r3 + 1u;
                                        if (digits && depth >= 1u && b0 + 1u + depth < le) {
                                            const uint32_t t4 = le - depth - 1u;
                                            const uint32_t t4w = tabbits[t4 >> 5];
                                            // TABs in [p0, le), letters in [b0, t4): differences of running counts
                                            auto upto = [&](uint32_t at, uint32_t word, uint32_t shift) -> uint32_t {
                                                return ((pre[at >> 5] >> shift) & 0xFFFFu) + (uint32_t)__popc(word & ((1u << (at & 31u)) - 1u));
                                            };
                                            const uint32_t tabs = (upto(le, tabbits[le >> 5], 0) - upto(p0, a0, 0)) & 0xFFFFu;
                                            const uint32_t letters = (upto(t4, letbits[t4 >> 5], 16) - upto(b0, letbits[b0 >> 5], 16)) & 0xFFFFu;
                                            if (((t4w >> (t4 & 31u)) & 1u) && tabs == 5u) {
                                                plain = true;
                                                // (min_reads2 0: an allele without reads is skipped by the walk, so one letter is still needed)
                                                is_cand = depth >= prm.min_coverage && letters >= (prm.min_reads2 > 1u ? prm.min_reads2 : 1u);
                                                e.w = VS_W_PLAIN | depth | (r1 << 14) | (r3 << 20);
                                            }
                                        }
                                    }
                                }
                                if (!plain) is_cand = true;                              // the walk looks at it in full (format errors included)
                            }
                        }
                    }
                    const unsigned long long mk = __ballot(is_cand);
                    if (mk) {
                        if (is_cand) cand_local[n_local + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = e;
                        n_local += (uint32_t)__popcll(mk);
                        __builtin_amdgcn_wave_barrier();
                        if (n_local + 64u > VS_CAND_LOCAL) { flush(); __builtin_amdgcn_wave_barrier(); }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!redo) lines_seen += T;
        } while (redo);
        // this slot is free: every LDS read of it has returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        dma_next = tt + 2 < t_end ? request(tt + 2, cur) : false;
        uint4 *t_ = cur; cur = other; other = t_;
    }
    __builtin_amdgcn_wave_barrier();
    flush();
    if (lane == 0) wave_lines[gwave] = lines_seen;
}

// One lane per candidate line: the 64 lines of a wave are copied into LDS back to back (16-byte chunks; a wave prefix sum of
// the chunk counts gives every lane its place) and every lane walks its own there.  What does not fit the wave's LDS (a few
// lines several times the mean length) and what the scan could not measure (VS_W_LONG) goes on a second list.  The entries are
// fetched two rounds ahead of their use.
__global__ __launch_bounds__(64) void k_varscan_walk(const uint8_t *__restrict__ buf, uint64_t nbytes, snpgpu_varscan_params prm, snpgpu_varscan_site *out,
                                                     uint32_t capacity, uint32_t *out_n, unsigned long long *status, uint32_t lds_bytes,
                                                     const uint4 *__restrict__ cand, const uint32_t *__restrict__ cand_n, uint32_t cand_cap,
                                                     uint32_t *long_idx, uint32_t *long_n) {
    extern __shared__ uint4 vs_lds[];
    const uint32_t n = *cand_n < cand_cap ? *cand_n : cand_cap;
    const uint64_t stride = (uint64_t)gridDim.x * 64;
    uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint4 NONE = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
    auto entry = [&](uint64_t k) -> uint4 { return k < n ? cand[k] : NONE; };
    uint4 e = entry(i), e1 = entry(i + stride);
    for (uint64_t i0 = (uint64_t)blockIdx.x * 64; i0 < n; i0 += stride, i += stride) {
        const uint4 e2 = entry(i + 2 * stride);                      // two rounds ahead
        const bool have = !(e.x == 0xFFFFFFFFu && e.y == 0xFFFFFFFFu);
        const bool is_long = have && (e.w & VS_W_LONG);
        const uint64_t p0 = (uint64_t)e.x | ((uint64_t)e.y << 32), end = p0 + e.z;
        const uint64_t a0 = ((uintptr_t)buf + p0) & ~(uint64_t)15, a1 = (have && !is_long) ? (((uintptr_t)buf + end) + 15) & ~(uint64_t)15 : a0;
        const uint32_t chunks = a1 - a0 > 0xFFFFFFull ? 0xFFFFFFu / 16u : (uint32_t)((a1 - a0) / 16);
        // exclusive prefix sum of the chunk counts over the wave (a line that does not fit takes no room)
        const bool alone_fits = (uint64_t)chunks * 16 <= lds_bytes;
        const uint32_t mine_chunks = alone_fits ? chunks : 0u;
        uint32_t incl = mine_chunks;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up_ = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up_;
        }
        const uint32_t first_chunk = incl - mine_chunks;
        const bool fits = alone_fits && (uint64_t)incl * 16 <= lds_bytes;
        if (have && !is_long && fits) {
            uint4 *mine = vs_lds + first_chunk;
            const uint4 *src = (const uint4 *)a0;
            for (uint32_t c = 0; c < chunks; ++c) mine[c] = src[c];
            const uint32_t lane0 = first_chunk * 16u;                                   // LDS offset of this lane's bytes
            const uint64_t a0_off = a0 - (uintptr_t)buf;                                // their file offset
            const uint32_t *lds32 = (const uint32_t *)vs_lds;
            const uint32_t l0 = lane0 + (uint32_t)(p0 - a0_off);
            uint32_t l1 = lane0 + (uint32_t)(end - a0_off);
            LineCols cols;
            bool ok;
            if (e.w & VS_W_PLAIN) {
                // the columns are where k_varscan_scan found them
                while (l1 > l0) { const uint32_t c = (lds32[(l1 - 1u) >> 2] >> (((l1 - 1u) & 3u) * 8u)) & 0xFFu; if (c != 10u && c != 13u) break; --l1; }
                const uint32_t depth = e.w & 0x3FFFu, t1 = l0 + ((e.w >> 14) & 63u), t3 = l0 + ((e.w >> 20) & 63u);
                const uint32_t t4 = l1 - depth - 1u;
                cols = LineCols{t1 + 1u, depth, t3 + 1u, t4, t4 + 1u, l1};
                ok = true;
            } else {
                ok = varscan_parse_lds(lds32, l0, l1, a0_off - lane0, status, cols);
            }
            if (ok) varscan_core_lds(lds32, l0, cols.ref_at, cols.depth, cols.b0, cols.b1, cols.q0, cols.q1, a0_off - lane0, prm, out, capacity, out_n);
        } else if (have) {
            long_idx[atomicAdd(long_n, 1u)] = (uint32_t)i;                              // (rare)
        }
        e = e1; e1 = e2;
    }
}

// The candidates that did not fit a strip or have no known end: one lane per line, bytes straight from global memory.  The blocks
// also add up the line counts of the scan's waves (the one number besides the records that the host is told; ctl[4..5] start at 0).
__global__ __launch_bounds__(64) void k_varscan_walk_long(const uint8_t *__restrict__ buf, uint64_t nbytes, snpgpu_varscan_params prm, snpgpu_varscan_site *out,
                                                          uint32_t capacity, uint32_t *ctl, unsigned long long *status, const uint4 *__restrict__ cand,
                                                          const uint32_t *__restrict__ long_idx, const uint32_t *__restrict__ wave_lines, uint32_t n_waves) {
    const uint32_t n = ctl[2];
    for (uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 64)
        walk_entry_global(buf, nbytes, cand[long_idx[i]], prm, out, capacity, ctl, status);
    {   // every block a slice of the waves' line counts (up to 16 384 of them: one block alone would take longer than the scan's tail)
        unsigned long long s = 0;
        for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < n_waves; k += gridDim.x * 64u) s += wave_lines[k];
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if (threadIdx.x == 0 && s) atomicAdd((unsigned long long *)(ctl + 4), s);
    }
}

}  // namespace

// Device scratch the site calling of one file needs besides its records: the candidate list (16 bytes per entry, one entry per 64
// bytes of text: twelve times what a 30x pileup puts on it; a list that is full anyway costs speed, not answers), the indices of
// the long candidates, one line count per scan wave.
static inline uint32_t varscan_cand_cap(uint64_t nbytes) {
    const uint64_t c = nbytes / 64 + 4096;
    return (uint32_t)(c < 0x7FFFFFFFull ? c : 0x7FFFFFFFull);
}
static const uint32_t VARSCAN_MAX_WAVES = 256 * 64;
size_t snpgpu_varscan_scratch_bytes(uint64_t nbytes) {
    return (size_t)varscan_cand_cap(nbytes) * 20u + 4u * VARSCAN_MAX_WAVES + 1024;
}

// d_ctl: 8 zeroed words — [0] records found, [1] candidate lines, [2] long candidates, [4..5] lines of the file (written by the
// last kernel); d_status: one u64 preset to UINT64_MAX (becomes the offset of the first malformed line); d_scratch:
// snpgpu_varscan_scratch_bytes(nbytes) bytes, 16-byte aligned.
int snpgpu_varscan_halo_class(const uint8_t *head, uint64_t n) {
    // from the first bytes of the file (on the host): 0 for lines of up to ~110 bytes on average, 1 up to ~480, else 2
    uint64_t lines = 0;
    for (uint64_t i = 0; i < n; ++i) lines += head[i] == '\n' || head[i] == '\r';
    const uint64_t mean = n / (lines ? lines : 1);
    return mean <= 110 ? 0 : mean <= 480 ? 1 : 2;
}

int snpgpu_enqueue_varscan(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, const snpgpu_varscan_params *prm, snpgpu_varscan_site *d_sites,
                           uint32_t capacity, uint32_t *d_ctl, uint64_t *d_status, void *d_scratch, int halo_class) {
    if (nbytes == 0) return SNPGPU_OK;
    const uint32_t cand_cap = varscan_cand_cap(nbytes);
    uint4 *d_cand = (uint4 *)d_scratch;
    uint32_t *d_long = (uint32_t *)((char *)d_scratch + (size_t)cand_cap * 16u);
    uint32_t *d_wave_lines = d_long + cand_cap;
    const uint64_t shift = (uintptr_t)d_buf & 15u;
    const uint8_t *abase = d_buf - shift;
    const uint64_t lo = shift, hi = shift + nbytes;
    // (halo_class comes from snpgpu_varscan_halo_class over the file's first bytes: how long the lines are decides the geometry)
    const uint32_t cpl = halo_class == 0 ? 4u : halo_class == 1 ? 6u : 8u;             // 16-byte chunks per lane: the slot is 64 of that
    const uint32_t tile_bytes = halo_class == 0 ? 3840u : halo_class == 1 ? 5120u : 4096u;
    const uint64_t n_tiles = (hi + tile_bytes - 1) / tile_bytes;
    const uint32_t slot_bytes = cpl * 1024u, kw = slot_bytes / 32u;
    const uint32_t lds = 2u * slot_bytes + 5u * (kw + 4u) * 4u + VS_CAND_LOCAL * 16u + (VS_LIST_CAP + 2u) * 2u + 12u;
    const uint32_t lds_wave = (lds + 15u) / 16u * 16u;                                 // a wave's stretch of the workgroup's LDS
    uint32_t waves_per_cu = 160u * 1024u / lds_wave;
    if (waves_per_cu > 16u) waves_per_cu = 16u;                                        // (its registers allow 4 per SIMD)
    // Eight workgroups (of one wave) for every place a CU has: the dispatcher hands out the next one when a wave ends, which evens
    // out what a grid of exactly the resident size leaves to chance — how many waves share a SIMD, the oldest of them taking most
    // issue slots (tools/vs_sweep.sh: 250 us for 432 MB with 12 waves per CU and one workgroup each, 174 us with eight each; 30x).
    // One workgroup per CU, a multiple of four waves (the same number on every SIMD; 13 waves: 240 us where 12 take 170), every
    // wave one contiguous run of tiles weighted by its age on its SIMD.  (Round 4's first form — one-wave workgroups, eight per
    // resident place, balanced by the dispatcher — took 182 us in the same session, each short-lived wave paying its prologue.)
    uint32_t wg_waves = waves_per_cu >= 4u ? waves_per_cu / 4u * 4u : 1u;
    uint64_t grid = (uint64_t)ctx->n_cu * (wg_waves == 1u ? waves_per_cu : wg_waves);
    if (grid > n_tiles / 4u) grid = n_tiles / 4u ? n_tiles / 4u : 1u;                   // (at least four tiles per wave)
    uint32_t share[4] = {120, 100, 82, 70};                                             // tools/vs_share_sweep.sh: 181 -> 165 us at 30x, 513 -> 480 at 100x, 110 -> 100 at 8x
#ifdef SNPGPU_TUNING                                            // development builds only (tools/)
    if (const char *e = getenv("SNPGPU_VS_WAVES")) if (atoi(e) > 0) grid = (uint64_t)ctx->n_cu * (uint32_t)atoi(e);
    if (const char *e = getenv("SNPGPU_VS_GRID_MUL")) if (atoi(e) > 0) grid *= (uint32_t)atoi(e);
    if (const char *e = getenv("SNPGPU_VS_WG_WAVES")) if (atoi(e) > 0 && atoi(e) <= 16 && (uint32_t)atoi(e) * lds_wave <= 160u * 1024u) wg_waves = (uint32_t)atoi(e);
    if (const char *e = getenv("SNPGPU_VS_SHARE")) { int v[4]; if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) for (int k = 0; k < 4; ++k) share[k] = v[k] > 0 ? (uint32_t)v[k] : 1u; }
#endif
    grid = (grid + wg_waves - 1) / wg_waves;                                            // (counted in waves so far; the launch counts workgroups)
    if (grid * wg_waves > VARSCAN_MAX_WAVES) grid = VARSCAN_MAX_WAVES / wg_waves;
    if (grid * wg_waves > n_tiles) grid = (n_tiles + wg_waves - 1) / wg_waves;
    if (!ctx->varscan_lds_attr) {                                                       // (more than 64 KiB of dynamic LDS needs the permission, per device)
        for (auto f : {(const void *)k_varscan_scan<4>, (const void *)k_varscan_scan<6>, (const void *)k_varscan_scan<8>})
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ctx->varscan_lds_attr = true;
    }
    hipEvent_t ta = snpgpu_time_begin(ctx);
#define VS_SCAN(C) k_varscan_scan<C><<<(unsigned)grid, 64u * wg_waves, lds_wave * wg_waves, ctx->stream>>>(abase, lo, hi, tile_bytes, n_tiles, *prm, d_cand, cand_cap, \
                                                                            d_ctl, (unsigned long long *)d_status, d_sites, capacity, d_wave_lines, lds_wave / 16u, make_uint4(share[0], share[1], share[2], share[3]))
    if (halo_class == 0) VS_SCAN(4); else if (halo_class == 1) VS_SCAN(6); else VS_SCAN(8);
#undef VS_SCAN
    // walk: LDS for 64 candidate lines — the deeper lines of the file — 2 .. 60 KiB
    const uint32_t walk_bytes = halo_class == 0 ? 16u * 1024u : halo_class == 1 ? 32u * 1024u : 60u * 1024u;
    const uint32_t walk_lds = walk_bytes + 16;                   // (+ one chunk: the walk may read the word after a line)
    const uint32_t walk_waves_per_cu = 160 * 1024 / walk_lds < 16 ? 160 * 1024 / walk_lds : 16;
    const unsigned walk_grid = (unsigned)((uint64_t)ctx->n_cu * (walk_waves_per_cu ? walk_waves_per_cu : 1));
    k_varscan_walk<<<walk_grid, 64, walk_lds, ctx->stream>>>(d_buf, nbytes, *prm, d_sites, capacity, d_ctl, (unsigned long long *)d_status, walk_bytes, d_cand,
                                                              d_ctl + 1, cand_cap, d_long, d_ctl + 2);
    k_varscan_walk_long<<<ctx->n_cu, 64, 0, ctx->stream>>>(d_buf, nbytes, *prm, d_sites, capacity, d_ctl, (unsigned long long *)d_status, d_cand, d_long,
                                                           d_wave_lines, (uint32_t)(grid * wg_waves));
    snpgpu_time_end(ctx, SNPGPU_K_VARSCAN, ta);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}
