// Phase-1 site calling over every line of a pileup: the counting and selection half of `VarScan mpileup2snp`.
//
// Replaces what snppipeline/call_sites.py:89-108 obtains from the VarScan v2.3.9 jar (net.sf.varscan:
// VarScan.qualityDepth, VarScan.getReadCounts, the selection tests of VarScan.callPosition): per pileup line, the raw
// depth, the number of qualities >= min-avg-qual, reads per allele and strand at that quality with their quality sums,
// the indel-carrying reads (they count in the frequency's denominator), and the tests min-coverage / min-reads2 /
// min-avg-qual / min-var-freq.  Lines that pass leave one 48-byte record per passing allele; Fisher's exact test, the
// strand filter and the VCF text are host work on those few records (snp_pipeline_amd/varscan.py).
//
// k_varscan_scan reads the text once (one launch over any number of resident pileups; every wave a run of 4 KiB tiles of one file,
// streamed through a ring of three LDS slots by LDS-DMA), decides per line in constant time whether it can call anything, and walks
// the few lines that can with the exact read-base automaton when its tiles are done; k_varscan_finish takes what is left over and adds
// up the line counts (see the comment above k_varscan_scan).  A file arrives over PCIe at ~50 GB/s, so from files the pass is bounded
// by that copy; over resident pileups it is bounded by HBM and the VALU.  The read-base automaton follows the restatement in
// oracle/varscan_oracle.py (which tests compare it with); see its header for what the reference's fixtures pin.
#include <string.h>

#include <algorithm>
#include <vector>

#include "internal.h"
#include "prims.h"

namespace {

struct Acc { uint32_t f, r, q; };

__device__ __forceinline__ bool is_digit(uint32_t c) { return c - 0x30u < 10u; }


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

// ---- one pass over the text: scan + select every line, walk the few that can call something -----------------------------------
// Nearly every line of a real pileup cannot reach min-reads2 for any allele, and the byte-wise walk over its read bases — 90 % of
// this step's instructions in round 2 — computes nothing that is used.  So the scan (k_varscan_scan) reads the text once, has no
// loop over the bytes of a line, and only the lines that might call something ("candidates") are walked with the exact automaton.
// Round 5: the scan has no list of line starts and no window geometries any more (rounds 3-4 built the list with a wave prefix sum
// in LDS and re-classified a halo of 6-100 % of every tile); a launch serves any number of pileups; a wave walks its own
// candidates when its tiles are done, so a file costs one launch and a fixed-size epilogue, not three dependent ones.
//
// A wave owns a contiguous run of 4 KiB tiles of ONE file and two tile slots in LDS: tile k+1 streams into its slot with LDS-DMA
// (global_load_lds_dwordx4, four wave instructions, no register round trip, counted s_waitcnt) while tile k is classified and its lines
// are looked at.  In front of each slot lie the last 64 bytes of the tile before it (a line whose first columns straddle the tile
// edge is read in one piece); the bit strings and running counts below cover both slots, so a line that started in tile k-1 and ends
// in tile k is answered from them and nothing is classified twice.  Per tile:
//   A  a lane owns 64 CONTIGUOUS bytes (four conflict-free ds_read_b128, chunk order i ^ ((lane >> 2) & 3)) and classifies them with
//      SWAR adds on 8 bytes at a time into three 64-bit masks: terminators (bytes 0x0A..0x0D: w + 0x76 carries into bit 7, w + 0x72
//      does not), TABs (w + 0x77 carries, w + 0x76 does not: the sums are shared), "letters" (bit 6 set, bit 3 clear: every
//      ACGTacgt, no N / '^' / digit / '.' / ','), gathered with v_dot4_u32_u8.  The TAB and letter masks go to LDS as bit strings
//      over the ring together with running counts per 32 bytes (one wave prefix sum of packed popcounts); the terminator mask
//      stays in registers: the terminators a lane found are the lines it looks at.  Every flagged terminator is read back once:
//      the first that is not LF (CR, VT, FF) switches the wave to exact LF / CR masks and Java's readLine() rules for good.
//   B  one lane per line, one line per lane and round (30x: one round per tile): the line behind the lane's lowest terminator
//      left; it ends at the lane's next terminator, else at the first one of the next lane that has any (one ballot, one
//      ds_bpermute), else in the next tile: then what its first columns said travels there in scalars, and the first lane without
//      a line of its own takes it.  The lane answers in constant time from the strings: the first four TABs
//      (find-first-set over 32, for long contig names 64, bits), the depth (one unaligned load, SWAR digit test, v_dot4 decimal),
//      the fifth TAB where a quality column of exactly `depth` bytes puts it, five TABs in all and the letters of the read-base
//      column (differences of running counts).  A well-formed ("plain") line with fewer letters than min-reads2 cannot call
//      anything and is done; with at least that many it becomes a 16-byte candidate entry (file offset, length, depth, TAB
//      places); a line the shortcut cannot vouch for becomes an entry as it is, and the walk parses it in full (format errors
//      included).  A line that does not end by tile k (> 4 KiB) is left to the epilogue kernel with its end unknown.
// Candidates collect in the wave's LDS (what does not fit there until the end: in the wave's own stretch of a spill list in global
// memory).  When the wave's tiles are done its slots are free: the wave packs its candidates' lines back to back into them (LDS-DMA,
// mostly L2 hits) and every lane runs the exact read-base automaton over its own line (varscan_core_lds).  Lines that fit no strip
// or have no known end go to a shared list for k_varscan_finish, which also adds up the line counts of the waves per file.

// A list entry (16 bytes): x, y = file offset of the line's first byte (48 bits) and, in the upper half of y, the index of the file
// in the launch; z = its length in bytes (terminator included; 0 for a line whose end the scan does not know: VS_W_LONG, the walk
// finds it); w = what the walk need not find out again for a line whose shape the scan has checked (VS_W_PLAIN): depth (bits
// 0-13), and where, counted from the line's first byte, its second and fourth TAB are (bits 14-19, 20-25)
constexpr uint32_t VS_W_PLAIN = 1u << 31, VS_W_LONG = 1u << 30;

#define VS_TILE 4096u
#define VS_RING (2u * VS_TILE)                // the two tile slots of a wave
#define VS_PAD 64u                            // in front of each slot: the last bytes of the tile before the one in it
#define VS_DATA (2u * (VS_PAD + VS_TILE))     // the slots with their pads
#define VS_STR_WORDS (VS_RING / 32u)          // dwords of a bit string (and entries of the running counts) over the ring
#define VS_STR_PAD 4u                         // ... and room for the reads that run over its end (their bits are masked away)
#define VS_CAND_LOCAL 136u                    // candidate entries a wave holds in LDS (100x: a dozen per wave; what LDS is left at twelve waves per CU)
constexpr uint32_t VS_LDS_WAVE = VS_DATA + 3u * (VS_STR_WORDS + VS_STR_PAD) * 4u + VS_CAND_LOCAL * 16u;   // 13 616 bytes: twelve waves per CU (163 392 of 163 840 bytes)
static_assert(12u * VS_LDS_WAVE <= 160u * 1024u && VS_LDS_WAVE % 16u == 0u, "twelve waves of k_varscan_scan no longer fit a CU's LDS");
#define VS_NONE 0xFFFFFFFFu

// One pileup of a launch: an entry of the device table, or (a launch over one file) a kernel argument.
struct VsFile {
    const uint8_t *abase;       // 16-byte aligned, 16..31 bytes below the file's first byte
    uint64_t lo, hi;            // the file is abase[lo, hi); position lo - 1 and position hi count as '\n'
    uint64_t n_tiles;           // tiles of VS_TILE bytes over abase[0, hi]
    uint32_t wave0, n_waves;    // its waves in the launch
    snpgpu_varscan_site *out;   // records
    uint32_t capacity, pad;
    uint32_t *ctl;              // [0] records found, [4..5] lines of the file (u64)
    unsigned long long *status; // offset of the first malformed line (preset to UINT64_MAX)
};
// where the walk of one file's lines reads and writes
struct VsOut {
    const uint8_t *buf; uint64_t nbytes; snpgpu_varscan_site *out; uint32_t capacity; uint32_t *out_n; unsigned long long *status;
};
__device__ __forceinline__ VsOut vs_out(const VsFile &f) { return VsOut{f.abase + f.lo, f.hi - f.lo, f.out, f.capacity, f.ctl, f.status}; }

// A candidate straight from global memory, byte by byte (the generic form of the walk; also what a full list falls back on).
__device__ __noinline__ void walk_entry_global(VsOut o, uint4 e, const snpgpu_varscan_params &prm) {
    const uint64_t p0 = (uint64_t)e.x | ((uint64_t)(e.y & 0xFFFFu) << 32);
    uint64_t end = p0 + e.z;
    if (e.w & VS_W_LONG) {                                                              // ends at the first line terminator
        end = p0;
        while (end < o.nbytes && o.buf[end] != 10u && o.buf[end] != 13u) ++end;
    }
    varscan_line<uint64_t>(GlobalBytes{o.buf}, p0, end < o.nbytes ? end : o.nbytes, 0, prm, o.out, o.capacity, o.out_n, o.status);
}

// The candidates of a wave's lanes (`have`: this lane has one) packed back to back into `strips` (16-byte chunks; a wave prefix sum
// of the chunk counts gives every lane its place) and walked there, every lane its own.  Returns whether this lane's candidate has
// been dealt with; false: it found no room this time (or has no known end: VS_W_LONG).
__device__ __forceinline__ bool walk_in_strips(uint4 *strips, uint32_t strip_bytes, bool have, uint4 e, const VsOut &o, const snpgpu_varscan_params &prm) {
    const bool is_long = have && (e.w & VS_W_LONG);
    const uint64_t p0 = (uint64_t)e.x | ((uint64_t)(e.y & 0xFFFFu) << 32), end = p0 + e.z;
    const uint64_t a0 = ((uintptr_t)o.buf + p0) & ~(uint64_t)15, a1 = (have && !is_long) ? (((uintptr_t)o.buf + end) + 15) & ~(uint64_t)15 : a0;
    const uint32_t chunks = a1 - a0 > 0xFFFFFFull ? 0xFFFFFFu / 16u : (uint32_t)((a1 - a0) / 16);
    const bool alone_fits = (uint64_t)chunks * 16 <= strip_bytes;
    const uint32_t mine_chunks = alone_fits ? chunks : 0u;
    const uint32_t incl = wave_inclusive_sum(mine_chunks);
    const uint32_t first_chunk = incl - mine_chunks;
    const bool fits = alone_fits && (uint64_t)incl * 16 <= strip_bytes;
    const bool mine_fits = have && !is_long && fits;
    // the lines that fit, one after the other, each by LDS-DMA (64 lanes x 16 bytes per instruction, nothing waits before the last one
    // is under way: a lane copying its own line chunk by chunk pays the memory latency once per chunk)
    {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)strips);
        const uint32_t voff = lane * 16u;
        uint64_t todo = __builtin_amdgcn_ballot_w64(mine_fits);
        while (todo) {
            const uint32_t c = (uint32_t)__ffsll((long long)todo) - 1u;
            todo &= todo - 1;
            const uint32_t a_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a0, (int)c), a_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(a0 >> 32), (int)c);
            const uint32_t nch = (uint32_t)__builtin_amdgcn_readlane((int)chunks, (int)c), dst = (uint32_t)__builtin_amdgcn_readlane((int)first_chunk, (int)c);
            for (uint32_t o = 0; o < nch; o += 64) {
                const uint64_t ga = (((uint64_t)a_hi << 32) | a_lo) + (uint64_t)o * 16u;
                const uint64_t gr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ga >> 32)) << 32) |
                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ga);
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (dst + o) * 16u);
                if (o + lane < nch)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gr), "s"(m0v) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (!mine_fits) return false;
    const uint32_t lane0 = first_chunk * 16u;                                           // LDS offset of this lane's bytes
    const uint64_t a0_off = a0 - (uintptr_t)o.buf;                                      // their file offset (may be "negative": the chunk starts below the file)
    const uint32_t *lds32 = (const uint32_t *)strips;
    const uint32_t l0 = lane0 + (uint32_t)(p0 - a0_off);
    uint32_t l1 = lane0 + (uint32_t)(end - a0_off);
    LineCols cols;
    bool ok;
    if (e.w & VS_W_PLAIN) {                                                             // the columns are where the scan found them
        while (l1 > l0) { const uint32_t c = (lds32[(l1 - 1u) >> 2] >> (((l1 - 1u) & 3u) * 8u)) & 0xFFu; if (c != 10u && c != 13u) break; --l1; }
        const uint32_t depth = e.w & 0x3FFFu, t1 = l0 + ((e.w >> 14) & 63u), t3 = l0 + ((e.w >> 20) & 63u);
        const uint32_t t4 = l1 - depth - 1u;
        cols = LineCols{t1 + 1u, depth, t3 + 1u, t4, t4 + 1u, l1};
        ok = true;
    } else {
        ok = varscan_parse_lds(lds32, l0, l1, a0_off - lane0, o.status, cols);
    }
    if (ok) varscan_core_lds(lds32, l0, cols.ref_at, cols.depth, cols.b0, cols.b1, cols.q0, cols.q1, a0_off - lane0, prm, o.out, o.capacity, o.out_n);
    return true;
}

// 16 byte flags (bit `shift` of every byte: 7, or 6 for the letters) of a 16-byte chunk -> 16 bits, byte k -> bit k
__device__ __forceinline__ uint32_t gather16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3, uint32_t shift) {
    uint32_t a = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
    a = __builtin_amdgcn_udot4(f1, 0x80402010u, a, false);
    uint32_t b = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    b = __builtin_amdgcn_udot4(f3, 0x80402010u, b, false);
    return (a >> shift) | ((b >> shift) << 8);
}
// The 16-bit groups of a 64-bit mask from the order the lane read its chunks in (i ^ xq) into address order (and back: the same
// exchange): neighbours swap where xq & 1 (both dwords rotate by 16), the dwords swap where xq & 2.
__device__ __forceinline__ uint64_t regroup16(uint32_t lo, uint32_t hi, uint32_t xq) {
    const uint32_t rot = (xq & 1u) << 4;
    const uint32_t a_lo = __builtin_amdgcn_alignbit(lo, lo, rot), a_hi = __builtin_amdgcn_alignbit(hi, hi, rot);
    return (xq & 2u) ? ((uint64_t)a_hi | ((uint64_t)a_lo << 32)) : ((uint64_t)a_lo | ((uint64_t)a_hi << 32));
}
// The exact, carry-safe masks of a lane's 64 bytes (any byte value): LF, CR, TAB, bytes 0x0A..0x0D, letters.  The slow side of
// phase A: tiles with a byte >= 0x80 in them (the sums of the fast form would carry into the neighbour), and LF / CR of every tile
// once a wave has seen a terminator that is not LF.
struct SafeMasks { uint64_t lf, cr, tab, nl, let; };
__device__ __noinline__ SafeMasks classify_safe(const uint4 *src, uint32_t xq) {
    uint32_t b_lf[4], b_cr[4], b_tab[4], b_nl[4], b_let[4];
#pragma unroll 1
    for (uint32_t i = 0; i < 4; ++i) {
        const uint4 v = src[i ^ xq];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t f_lf[4], f_cr[4], f_tab[4], f_nl[4], f_let[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f_lf[k] = eq4(w[k], 0x0A0A0A0Au);
            f_cr[k] = eq4(w[k], 0x0D0D0D0Du);
            f_tab[k] = eq4(w[k], 0x09090909u);
            const uint32_t m = w[k] & 0x7F7F7F7Fu;
            f_nl[k] = (m + 0x76767676u) & ~(m + 0x72727272u) & ~w[k] & 0x80808080u;
            f_let[k] = w[k] & ~(w[k] << 3) & ~(w[k] >> 1) & 0x40404040u;                 // bit 6 set, bits 3 and 7 clear
        }
        b_lf[i] = gather16(f_lf[0], f_lf[1], f_lf[2], f_lf[3], 7);
        b_cr[i] = gather16(f_cr[0], f_cr[1], f_cr[2], f_cr[3], 7);
        b_tab[i] = gather16(f_tab[0], f_tab[1], f_tab[2], f_tab[3], 7);
        b_nl[i] = gather16(f_nl[0], f_nl[1], f_nl[2], f_nl[3], 7);
        b_let[i] = gather16(f_let[0], f_let[1], f_let[2], f_let[3], 6);
    }
    SafeMasks r;
    r.lf = regroup16(b_lf[0] | (b_lf[1] << 16), b_lf[2] | (b_lf[3] << 16), xq);
    r.cr = regroup16(b_cr[0] | (b_cr[1] << 16), b_cr[2] | (b_cr[3] << 16), xq);
    r.tab = regroup16(b_tab[0] | (b_tab[1] << 16), b_tab[2] | (b_tab[3] << 16), xq);
    r.nl = regroup16(b_nl[0] | (b_nl[1] << 16), b_nl[2] | (b_nl[3] << 16), xq);
    r.let = regroup16(b_let[0] | (b_let[1] << 16), b_let[2] | (b_let[3] << 16), xq);
    return r;
}

__device__ __forceinline__ uint32_t ffs64(uint64_t x) { return (uint32_t)__ffsll((long long)x) - 1u; }     // (x != 0)

__global__ __launch_bounds__(768) void k_varscan_scan(const VsFile *__restrict__ files, uint32_t n_files, VsFile one, uint32_t n_waves_total,
                                                      snpgpu_varscan_params prm, uint4 *cand, uint32_t cand_cap, uint32_t *cand_n, uint32_t *wave_lines,
                                                      uint4 share, uint4 *spill, uint32_t spill_per_wave) {
    extern __shared__ uint4 vs_lds_all[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const uint32_t gwave = blockIdx.x * wpb + wave_in_wg;
    if (gwave >= n_waves_total) return;
    // my file: the host dealt the waves of the launch to the files in proportion to their sizes
    uint32_t fi = 0;
    if (n_files > 1) {
        uint32_t lo_i = 0, hi_i = n_files;                                              // the last file whose first wave is <= gwave
        while (hi_i - lo_i > 1) { const uint32_t mid = (lo_i + hi_i) >> 1; if (files[mid].wave0 <= gwave) lo_i = mid; else hi_i = mid; }
        fi = __builtin_amdgcn_readfirstlane(lo_i);
    }
    const VsFile f = n_files > 1 ? files[fi] : one;
    const VsOut fo = vs_out(f);
    // the waves of a workgroup are independent (no barrier anywhere): each has its own stretch of the workgroup's LDS
    uint8_t *ring = (uint8_t *)vs_lds_all + (size_t)wave_in_wg * VS_LDS_WAVE;
    const uint32_t *ring32 = (const uint32_t *)ring;
    uint32_t *tabs32 = (uint32_t *)(ring + VS_DATA);
    uint32_t *lets32 = tabs32 + VS_STR_WORDS + VS_STR_PAD;
    uint32_t *pre = lets32 + VS_STR_WORDS + VS_STR_PAD;                                // (tabs before) | (letters before) << 16, per 32 bytes, modulo 2^16
    uint4 *cand_local = (uint4 *)(pre + VS_STR_WORDS + VS_STR_PAD);
    const uint32_t xq = (lane >> 2) & 3u;
    uint32_t n_local = 0, lines_seen = 0;
    uint32_t exact = 0;                                                                 // wave-uniform (kept in a scalar), sticky: exact LF / CR masks, readLine() rules

    // my run of tiles, in proportion to my share: wave w of a workgroup sits on SIMD w % 4 and is the (w / 4)-th oldest there; the
    // SIMD issues oldest first, so with equal shares the oldest wave finishes early and the youngest runs on alone (share.x: oldest)
    uint64_t t_first, t_end;
    {
        const uint32_t sh[4] = {share.x, share.y, share.z, share.w};
        uint64_t blk = 0;
        for (uint32_t w = 0; w < wpb; ++w) blk += sh[(w >> 2) & 3u];
        auto cum = [&](uint64_t x) -> uint64_t {                                        // total weight of the launch's waves [0, x)
            const uint32_t r = (uint32_t)(x % wpb);
            uint64_t c = 0;
            for (uint32_t w = 0; w < r; ++w) c += sh[(w >> 2) & 3u];
            return (x / wpb) * blk + c;
        };
        const uint64_t c_lo = cum(f.wave0), c_span = cum((uint64_t)f.wave0 + f.n_waves) - c_lo;
        t_first = (uint64_t)((unsigned __int128)f.n_tiles * (cum(gwave) - c_lo) / c_span);
        t_end = (uint64_t)((unsigned __int128)f.n_tiles * (cum((uint64_t)gwave + 1) - c_lo) / c_span);
    }
    if (t_first >= t_end) { if (lane == 0) wave_lines[gwave] = 0; return; }
    const uint64_t t_last = t_end < f.n_tiles ? t_end : f.n_tiles - 1;                  // the last tile I read: the one after my run (the end of my last line)

    // My candidates so far out of LDS: into my own stretch of the spill (no atomics, nobody else writes there; I walk them myself
    // when my tiles are done), and only when that is full onto the shared list for the epilogue kernel.
    uint32_t n_spilled = 0;
    uint4 *my_spill = spill + (size_t)gwave * spill_per_wave;
    auto flush = [&]() {
        if (n_local == 0) return;
        if (n_spilled + n_local <= spill_per_wave) {
            for (uint32_t k = lane; k < n_local; k += 64) my_spill[n_spilled + k] = cand_local[k];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (the requests in flight are counted from zero again)
            n_spilled += n_local;
            n_local = 0;
            return;
        }
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cand_n, n_local);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t k = lane; k < n_local; k += 64) {
            const uint4 e = cand_local[k];
            if (base + k < cand_cap) cand[base + k] = e;
            else walk_entry_global(fo, e, prm);                                         // the list is full: looked at on the spot
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // (the requests in flight are counted from zero again)
        n_local = 0;
    };
    auto interior = [&](uint64_t tt) { const uint64_t x0 = tt * VS_TILE; return x0 >= f.lo && x0 + VS_TILE <= f.hi; };
    // request tile tt into ring slot `slot`; returns whether it travels by DMA (else it has been staged synchronously)
    auto request = [&](uint64_t tt, uint32_t slot) -> bool {
        if (interior(tt)) {
            const uint64_t ga = (uint64_t)(uintptr_t)(f.abase + tt * VS_TILE);
            const uint64_t gr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ga >> 32)) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ga);
            const uint32_t voff = lane * 16u;
            const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)ring) + VS_PAD + slot * (VS_PAD + VS_TILE));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:0\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(gr), "s"(m0v) : "memory");
            return true;
        }
        // first / last tiles of the file: byte loads; the byte before the file and the byte behind it read as '\n' (byte 0 starts a
        // line, the last line ends with the file), the other bytes outside it as NUL
        const int64_t x0 = (int64_t)(tt * VS_TILE);
        uint4 *dst = (uint4 *)(ring + VS_PAD + slot * (VS_PAD + VS_TILE));
#pragma nounroll
        for (uint32_t c = lane; c < VS_TILE / 16u; c += 64) {
            uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t idx = x0 + (int64_t)c * 16 + 4 * k + j;
                    const uint32_t bb = (idx >= (int64_t)f.lo && idx < (int64_t)f.hi) ? (uint32_t)f.abase[idx]
                                        : ((idx + 1 == (int64_t)f.lo || idx == (int64_t)f.hi) ? 10u : 0u);
                    d[k] |= bb << (8 * j);
                }
            dst[c] = make_uint4(d[0], d[1], d[2], d[3]);
        }
        return false;
    };
    uint32_t dma_mask = 0;                                                              // bit s: the tile now in slot s travels by DMA and has not been waited for
    if (request(t_first, 0)) dma_mask |= 1u;
    if (t_first + 1 <= t_last && request(t_first + 1, 1)) dma_mask |= 2u;
    uint32_t slot = 0;                                                                  // ring slot of tile k
    uint32_t base_t = 0, base_l = 0;                                                    // running counts at the start of tile k (wave-uniform)
    // the line that started in the tile before and had not ended there (all wave-uniform): where it starts, counted from the start of
    // THAT tile, and what its first bytes said (bit 0: shape and depth vouched for; r1 << 1, r3 << 7, depth << 13)
    uint32_t c_valid = 0, c_p0 = 0, c_pack = 0;
    for (uint64_t k = t_first; k <= t_last; ++k) {
        const bool own = k < t_end;                                                     // (wave-uniform) one of my tiles; else the one behind them: only the line carried into it
        const uint32_t base = slot * VS_TILE;                                           // offset of tile k in the ring of the strings
        const uint32_t dbase = VS_PAD + slot * (VS_PAD + VS_TILE);                      // ... and of its bytes in LDS (in front of them: the last 64 of tile k - 1)
        const uint64_t t0k = k * VS_TILE;                                               // its aligned coordinate
        // the current tile's request has landed when only the next tile's is outstanding
        if (dma_mask & ~(1u << slot)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dma_mask = 0; }
        dma_mask &= ~(1u << slot);
        __builtin_amdgcn_wave_barrier();
#if defined(SNPGPU_TUNING) && defined(VS_EXP) && VS_EXP >= 2         // (experiment builds: the stream alone — wait, touch, request)
        {
            lines_seen += ((const uint32_t *)(ring + dbase))[lane] & 1u;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (k + 2 <= t_last && request(k + 2, slot)) dma_mask |= 1u << slot;
            slot ^= 1u;
            continue;
        }
#endif
        // ---- A: my 64 bytes of tile k -> terminator / TAB / letter masks ---------------------------------------------------
        const uint4 *src = (const uint4 *)(ring + dbase) + 4u * lane;
        uint64_t m_nl, m_tab, m_let, cur_lf = 0, cur_cr = 0;
        {
            uint32_t bn[4], bt[4], bl[4], hi_any = 0;
            uint4 keepv[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint4 v = src[i ^ xq];
                keepv[i] = v;
                hi_any |= v.x | v.y | v.z | v.w;
                const uint64_t q0 = (uint64_t)v.x | ((uint64_t)v.y << 32), q1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
                const uint64_t a0 = q0 + 0x7777777777777777ull, b0 = q0 + 0x7676767676767676ull, c0 = q0 + 0x7272727272727272ull;
                const uint64_t a1 = q1 + 0x7777777777777777ull, b1 = q1 + 0x7676767676767676ull, c1 = q1 + 0x7272727272727272ull;
                bt[i] = gather16((uint32_t)a0 & ~(uint32_t)b0 & 0x80808080u, (uint32_t)(a0 >> 32) & ~(uint32_t)(b0 >> 32) & 0x80808080u,
                                 (uint32_t)a1 & ~(uint32_t)b1 & 0x80808080u, (uint32_t)(a1 >> 32) & ~(uint32_t)(b1 >> 32) & 0x80808080u, 7);
                bn[i] = gather16((uint32_t)b0 & ~(uint32_t)c0 & 0x80808080u, (uint32_t)(b0 >> 32) & ~(uint32_t)(c0 >> 32) & 0x80808080u,
                                 (uint32_t)b1 & ~(uint32_t)c1 & 0x80808080u, (uint32_t)(b1 >> 32) & ~(uint32_t)(c1 >> 32) & 0x80808080u, 7);
                // (x & ~(x << 3) & 0x40404040 as shift + one v_bitop3: left to itself the compiler moves the complement in front of the shift)
                bl[i] = gather16(__builtin_amdgcn_bitop3_b32(v.x, v.x << 3, 0x40404040u, 0x20), __builtin_amdgcn_bitop3_b32(v.y, v.y << 3, 0x40404040u, 0x20),
                                 __builtin_amdgcn_bitop3_b32(v.z, v.z << 3, 0x40404040u, 0x20), __builtin_amdgcn_bitop3_b32(v.w, v.w << 3, 0x40404040u, 0x20), 6);
            }
            if (lane == 63) {                                                           // the tile's last 64 bytes in front of the other slot (lane 63 read chunk i ^ 3 in turn i)
                uint4 *pad = (uint4 *)(ring + (slot ? 0u : VS_PAD + VS_TILE));
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) pad[i ^ 3u] = keepv[i];
            }
            m_nl = regroup16(bn[0] | (bn[1] << 16), bn[2] | (bn[3] << 16), xq);
            m_tab = regroup16(bt[0] | (bt[1] << 16), bt[2] | (bt[3] << 16), xq);
            m_let = regroup16(bl[0] | (bl[1] << 16), bl[2] | (bl[3] << 16), xq);
            const bool high = __builtin_amdgcn_ballot_w64((hi_any & 0x80808080u) != 0u) != 0;
            if (high || __builtin_amdgcn_readfirstlane(exact)) {                        // (wave-uniform) the slow, exact side
                const SafeMasks sm = classify_safe(src, xq);
                if (high) { m_nl = sm.nl; m_tab = sm.tab; m_let = sm.let; }
                cur_lf = sm.lf; cur_cr = sm.cr;
            }
        }
        if (!__builtin_amdgcn_readfirstlane(exact)) {
            // every byte the fast form flagged: is it LF?  The first that is not — CR, VT, FF — switches the wave to the exact form
            // before anything of this tile has been used (the tiles before it hold nothing but LF)
            bool odd = false;
            uint64_t v = m_nl;
            const uint8_t *mine = ring + dbase + lane * 64u;
            while (__builtin_amdgcn_ballot_w64(v != 0)) {
                if (v) { odd |= mine[ffs64(v)] != 10u; v &= v - 1; }
            }
            if (__builtin_amdgcn_ballot_w64(odd)) {
                exact = 1u;
                const SafeMasks sm = classify_safe(src, xq);
                cur_lf = sm.lf; cur_cr = sm.cr;
            } else { cur_lf = m_nl; cur_cr = 0; }
        }
        // the strings and the running counts of this tile
        {
            const uint32_t ct0 = (uint32_t)__popc((uint32_t)m_tab), ct1 = (uint32_t)__popc((uint32_t)(m_tab >> 32));
            const uint32_t cl0 = (uint32_t)__popc((uint32_t)m_let), cl1 = (uint32_t)__popc((uint32_t)(m_let >> 32));
            const uint32_t mine = (ct0 + ct1) | ((cl0 + cl1) << 16);                   // (<= 64 each: the fields of the sums stay apart, <= 4096)
            const uint32_t incl = wave_inclusive_sum(mine), excl = incl - mine;
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t t_run = base_t + (excl & 0xFFFFu), l_run = base_l + (excl >> 16);
            const uint32_t blk = slot * 64u + lane;                                     // my 64-byte block of the ring
            ((uint2 *)tabs32)[blk] = make_uint2((uint32_t)m_tab, (uint32_t)(m_tab >> 32));
            ((uint2 *)lets32)[blk] = make_uint2((uint32_t)m_let, (uint32_t)(m_let >> 32));
            ((uint2 *)pre)[blk] = make_uint2((t_run & 0xFFFFu) | (l_run << 16), ((t_run + ct0) & 0xFFFFu) | ((l_run + cl0) << 16));
            base_t = (base_t + (tot & 0xFFFFu)) & 0xFFFFu;
            base_l = (base_l + (tot >> 16)) & 0xFFFFu;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- B: the lines that start behind the terminators of tile k, and the one carried into it --------------------------------
        const uint64_t hi_rel64 = f.hi - t0k;                                           // (t0k <= hi: the tile is one of the file's)
        const uint32_t hi_rel = hi_rel64 < 0x7FFFFFFFull ? (uint32_t)hi_rel64 : 0x7FFFFFFFu;
        // terminators: LF, and (exact form) a CR that no LF follows
        uint64_t T = cur_lf;
        const uint32_t is_exact = __builtin_amdgcn_readfirstlane(exact);
        if (is_exact) {
            uint32_t nb = (uint32_t)__shfl_down((int)(uint32_t)cur_lf, 1) & 1u;
            if ((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cur_cr >> 32), 63) >> 31) {                                                              // (rare) a CR in the tile's last byte: what follows it is in the next tile
                const uint64_t at = t0k + VS_TILE;
                const uint32_t nx = at < f.hi ? (uint32_t)f.abase[at] : (at == f.hi ? 10u : 0u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                dma_mask = 0;
                if (lane == 63) nb = nx == 10u ? 1u : 0u;
            }
            T = cur_lf | (cur_cr & ~((cur_lf >> 1) | ((uint64_t)nb << 63)));
        }
        // the first terminator of the tile: where the carried line ends
        uint32_t first_cur = VS_NONE;
        const uint64_t t_lanes = __builtin_amdgcn_ballot_w64(T != 0);
        {
            if (t_lanes) {
                const uint32_t j0 = ffs64(t_lanes);
                first_cur = j0 * 64u + (uint32_t)__builtin_amdgcn_readlane((int)(T ? ffs64(T) : 0u), (int)j0);
            }
        }
        if (c_valid && first_cur == VS_NONE) {                                          // the carried line runs on past this tile (> 4 KiB): the walk finds its end
            if (lane == 0) {
                const uint64_t off = t0k - VS_TILE + c_p0 - f.lo;
                cand_local[n_local] = make_uint4((uint32_t)off, (uint32_t)(off >> 32) | (fi << 16), 0u, VS_W_LONG);
            }
            ++n_local;
            c_valid = 0;
            __builtin_amdgcn_wave_barrier();
            if (n_local + 64u > VS_CAND_LOCAL) { flush(); dma_mask = 0; __builtin_amdgcn_wave_barrier(); }
        }
        // for every lane: the first terminator behind its own last one (VS_NONE: none in this tile)
        uint32_t nxt = VS_NONE;
        {
            const uint32_t first_t = T ? lane * 64u + ffs64(T) : VS_NONE;
            const uint64_t above = (t_lanes >> 1) >> lane;                              // lanes lane + 1 ..
            const uint32_t j = above ? lane + 1u + ffs64(above) : lane;
            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(j << 2), (int)first_t);
            if (above) nxt = got;
        }
        uint64_t pend = own ? T : 0ull;
#if defined(SNPGPU_TUNING) && defined(VS_EXP) && VS_EXP >= 1         // (experiment builds: what the rounds of phase B cost — they are left out)
        pend = 0; c_valid = 0; lines_seen += (uint32_t)__popcll(T);
#endif
        uint32_t n_valid = 0, n_pack = 0, n_p0 = 0;                                     // the line this tile hands on
        // One line per lane and round, in straight-line code: every test lands in a flag, every LDS read is issued whether its
        // line needs it or not (a read outside the wave's LDS returns nothing and costs nothing), so a round is three LDS
        // round trips — the TAB window; the depth and what hangs on the fourth TAB; what hangs on the depth — and no branch.  The
        // carried line takes the first lane that has no line of its own in a round.
        for (;;) {
            const uint64_t busy = __builtin_amdgcn_ballot_w64(pend != 0);
            if (!busy && !c_valid) break;
            uint32_t taker = 64u;
            if (c_valid && ~busy) { taker = ffs64(~busy); c_valid = 0; }
            const bool take = lane == taker;
            const bool active = (pend != 0) | take;
            const uint32_t bpos = ffs64(pend);                                          // (no terminator left: 0xFFFFFFFF)
            pend &= pend - 1;                                                           // (0 stays 0)
            // the line's first byte and the terminator that ends it, counted from the start of tile k (the carried line: p0 < 0)
            const uint32_t p0 = take ? c_p0 - VS_TILE : lane * 64u + bpos + 1u;
            const uint32_t e = take ? first_cur : (pend ? lane * 64u + ffs64(pend) : nxt);
            const bool starts = active & (take | (p0 < hi_rel));                        // (the virtual terminator behind the file starts no line)
            lines_seen += (starts & !take) ? 1u : 0u;                                   // (the carried line was counted where it started)
            const bool defer = starts & (e == VS_NONE);                                 // the last terminator of the tile: its line ends in a later one
            const uint32_t a0 = (base + p0) & (VS_RING - 1u);                           // ring offset of the line's first byte
            uint32_t le = e;                                                            // the line without its terminator: [p0, le)
            if (is_exact) {                                                             // (wave-uniform) CR LF: both go
                const uint32_t ae = (dbase + e) & 0x3FFFu;                              // (e == VS_NONE: anywhere, but inside the wave's LDS)
                const bool pair = (e != VS_NONE) & ((int32_t)(e - p0) > 0) & (ring[ae] == 10u) & (ring[(ae - 1u) & 0x3FFFu] == 13u);
                le = pair ? e - 1u : e;
            }
            const uint32_t span = le - p0;
            // the first four TABs out of the 64 bits of the TAB string from the line's first byte on
            const uint32_t wq = a0 >> 5, sh = a0 & 31u;
            const uint32_t w0 = tabs32[wq], w1 = tabs32[(wq + 1u) & (VS_STR_WORDS - 1u)], w2 = tabs32[(wq + 2u) & (VS_STR_WORDS - 1u)];
            const uint32_t pre_p0 = pre[wq];
            const uint32_t a_le = (base + le) & (VS_RING - 1u);
            const uint32_t tw_le = tabs32[a_le >> 5], pre_le = pre[a_le >> 5];
            uint32_t m32 = __builtin_amdgcn_alignbit(w1, w0, sh);
            m32 = span < 32u ? m32 & ((1u << (span & 31u)) - 1u) : m32;
            bool four = __popc(m32) >= 4;
            uint32_t r0 = (uint32_t)__ffs((int)m32) - 1u; m32 &= m32 - 1u;
            uint32_t r1 = (uint32_t)__ffs((int)m32) - 1u; m32 &= m32 - 1u;
            uint32_t r2 = (uint32_t)__ffs((int)m32) - 1u; m32 &= m32 - 1u;
            uint32_t r3 = (uint32_t)__ffs((int)m32) - 1u;
            if (__builtin_amdgcn_ballot_w64(starts & !four & ((span > 32u) | defer))) {   // (wave-uniform) long contig names: 64 bytes
                uint64_t M = (uint64_t)__builtin_amdgcn_alignbit(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
                M = span < 64u ? M & ((1ull << (span & 63u)) - 1ull) : M;
                four = __popcll(M) >= 4;
                r0 = ffs64(M); M &= M - 1ull;
                r1 = ffs64(M); M &= M - 1ull;
                r2 = ffs64(M); M &= M - 1ull;
                r3 = ffs64(M);
            }
            r0 &= 63u; r1 &= 63u; r2 &= 63u; r3 &= 63u;                                // (four == false: anything, but small)
            const uint32_t nd = r3 - r2 - 1u;                                           // digits of the depth
            // the depth: four bytes from its first digit on, most significant first
            const uint32_t da = (dbase + p0 + r2 + 1u) & 0x3FFFu;                       // (the carried line: p0 >= -63 where its first bytes matter, they are in the pad)
            const uint32_t d_lo = ring32[da >> 2], d_hi = ring32[(da >> 2) + 1u];
            const uint32_t x = __builtin_amdgcn_alignbyte(d_hi, d_lo, da & 3u);
            const uint32_t keep = nd >= 4u ? 0xFFFFFFFFu : (1u << (8u * (nd & 3u))) - 1u;
            const uint32_t z = (x & keep) | (0x30303030u & ~keep);                      // the bytes behind the digits read as '0'
            // every byte in '0'..'9': bit 7 clear, z + 0x46 below 0x80, z + 0x50 at or above it
            const bool digits = ((z | (z + 0x46464646u) | ~(z + 0x50505050u)) & 0x80808080u) == 0u;
            const uint32_t ys = (z - 0x30303030u) << ((8u * (4u - nd)) & 31u);          // digit k in byte 4 - nd + k: weights 1000, 100, 10, 1 by byte
            uint32_t depth = (__builtin_amdgcn_udot4(ys, 0x010A6400u, 0u, false) + (ys & 0xFFu) * 1000u) & 0x3FFFu;
            // what the line's first bytes say: four TABs where chrom / position / a one-byte reference / 1-4 digits of depth >= 1 put them.
            // A line that is handed on is vouched for here when all four lie in this tile (the strings of the next one are not there
            // yet); else the lane that takes it in the next tile looks again: the tile's last 64 bytes are in front of the next one.
            bool head = four & (r0 > 0u) & (r1 > r0 + 1u) & (r2 == r1 + 2u) & (nd - 1u < 4u) & digits & (depth >= 1u) & (!defer | (p0 + r3 < VS_TILE));
            // hand the line on ...
            const uint64_t dm = __builtin_amdgcn_ballot_w64(defer);
            if (dm) {
                const uint32_t dl = ffs64(dm);
                n_valid = 1u;
                n_p0 = (uint32_t)__builtin_amdgcn_readlane((int)p0, (int)dl);
                n_pack = (uint32_t)__builtin_amdgcn_readlane((int)((head ? 1u : 0u) | (r1 << 1) | (r3 << 7) | (depth << 13)), (int)dl);
            }
            // ... or take the one handed to this tile
            const bool taken = take & ((c_pack & 1u) != 0u);                            // (else: what this lane has just read itself, ...
            head = taken | (head & (!take | (c_p0 + 63u >= VS_TILE)));                  //  which is the line's when it starts in the pad)
            r1 = taken ? (c_pack >> 1) & 63u : r1;
            r3 = taken ? (c_pack >> 7) & 63u : r3;
            depth = taken ? c_pack >> 13 : depth;
            const uint32_t a_b0 = (a0 + r3 + 1u) & (VS_RING - 1u);
            const uint32_t lw_b0 = lets32[a_b0 >> 5], pre_b0 = pre[a_b0 >> 5];
            const bool fits = r3 + 2u + depth < span;                                   // (b0 + 1 + depth < le)
            const uint32_t t4 = le - depth - 1u;
            const uint32_t a_t4 = (base + t4) & (VS_RING - 1u);
            const uint32_t tw_t4 = tabs32[a_t4 >> 5], lw_t4 = lets32[a_t4 >> 5], pre_t4 = pre[a_t4 >> 5];
            // TABs in [p0, le), letters in [b0, t4): differences of running counts
            const uint32_t tabs = ((pre_le & 0xFFFFu) + (uint32_t)__popc(tw_le & ((1u << (a_le & 31u)) - 1u)) -
                                   (pre_p0 & 0xFFFFu) - (uint32_t)__popc(w0 & ((1u << sh) - 1u))) & 0xFFFFu;
            const uint32_t letters = ((pre_t4 >> 16) + (uint32_t)__popc(lw_t4 & ((1u << (a_t4 & 31u)) - 1u)) -
                                      (pre_b0 >> 16) - (uint32_t)__popc(lw_b0 & ((1u << (a_b0 & 31u)) - 1u))) & 0xFFFFu;
            const bool plain = head & fits & (((tw_t4 >> (a_t4 & 31u)) & 1u) != 0u) & (tabs == 5u);
            // (min_reads2 0: an allele without reads is skipped by the walk, so one letter is still needed)
            const bool calls = (depth >= prm.min_coverage) & (letters >= (prm.min_reads2 > 1u ? prm.min_reads2 : 1u));
            // an empty line: nothing; a line the shortcut cannot vouch for: the walk looks at it in full (format errors included)
            const bool is_cand = starts & !defer & ((int32_t)span > 0) & (!plain | calls);
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(is_cand);
            if (mk) {
                const uint64_t off = t0k + (uint64_t)(int64_t)(int32_t)p0 - f.lo;      // file offset of the line
                uint4 ent;
                ent.x = (uint32_t)off; ent.y = (uint32_t)(off >> 32) | (fi << 16);
                ent.z = (e + 1u < hi_rel ? e + 1u : hi_rel) - p0;                       // (the virtual terminator behind the file is no byte of the line)
                ent.w = plain ? (VS_W_PLAIN | depth | (r1 << 14) | (r3 << 20)) : 0u;
                if (is_cand) cand_local[n_local + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = ent;
                n_local += (uint32_t)__popcll(mk);
                __builtin_amdgcn_wave_barrier();
                if (n_local + 64u > VS_CAND_LOCAL) { flush(); dma_mask = 0; __builtin_amdgcn_wave_barrier(); }
            }
        }
        c_valid = n_valid; c_p0 = n_p0; c_pack = n_pack;
        // this slot is free: every LDS read of it has returned (the line handed on travels in registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (k + 2 <= t_last && request(k + 2, slot)) dma_mask |= 1u << slot;
        slot ^= 1u;
    }
    // ---- my candidates: their lines into the ring (free now), every lane walks its own ------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const uint32_t n_mine = n_local + n_spilled;                                        // first what is still in LDS, then what went to my spill
    for (uint32_t r = 0; r < n_mine; r += 64) {
        bool have = r + lane < n_mine;
        const uint4 e = have ? (r + lane < n_local ? cand_local[r + lane] : my_spill[r + lane - n_local]) : make_uint4(0, 0, 0, 0);
        for (int round = 0; round < 4 && __builtin_amdgcn_ballot_w64(have); ++round) {
            const bool done = walk_in_strips((uint4 *)ring, VS_DATA + 3u * (VS_STR_WORDS + VS_STR_PAD) * 4u - 16u, have, e, fo, prm);   // (- one chunk: the walk may read the word after a line)
            const bool progress = __builtin_amdgcn_ballot_w64(done && have) != 0;
            have = have && !done;
            __builtin_amdgcn_wave_barrier();
            if (!progress) break;
        }
        // what found no room (a line longer than the ring, many deep ones, an unknown end): onto the global list
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(have);
        if (mk) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cand_n, (uint32_t)__popcll(mk));
            base = __builtin_amdgcn_readfirstlane(base);
            if (have) {
                const uint32_t at = base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
                if (at < cand_cap) cand[at] = e;
                else walk_entry_global(fo, e, prm);
            }
        }
    }
    for (int o = 32; o; o >>= 1) lines_seen += __shfl_xor(lines_seen, o);
    if (lane == 0) wave_lines[gwave] = lines_seen;
}

// The epilogue of a launch: the candidates on the global list (a wave's overflow, lines that found no room in a ring, lines of
// unknown end) — one lane per candidate, lines packed into LDS as the scan's waves do it, the rest byte-wise from global memory —
// and the line counts of the scan's waves added up per file.
__global__ __launch_bounds__(64) void k_varscan_finish(const VsFile *__restrict__ files, uint32_t n_files, VsFile one, snpgpu_varscan_params prm,
                                                       const uint4 *__restrict__ cand, const uint32_t *__restrict__ cand_n, uint32_t cand_cap,
                                                       const uint32_t *__restrict__ wave_lines, uint32_t n_waves_total, uint32_t strip_bytes) {
    extern __shared__ uint4 vs_strips[];
    const uint32_t n = *cand_n < cand_cap ? *cand_n : cand_cap;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 64; i0 < n; i0 += (uint64_t)gridDim.x * 64) {
        const uint64_t i = i0 + threadIdx.x;
        const bool have = i < n;
        const uint4 e = have ? cand[i] : make_uint4(0, 0, 0, 0);
        const VsFile f = (n_files > 1 && have) ? files[e.y >> 16] : one;
        const VsOut o = vs_out(f);
        const bool done = walk_in_strips(vs_strips, strip_bytes, have, e, o, prm);
        if (have && !done) walk_entry_global(o, e, prm);
        __builtin_amdgcn_wave_barrier();
    }
    // the line counts: 64 waves of the scan per block and turn; a file's waves are neighbours, so nearly always the 64 belong to one
    // file and the block adds one sum to that file's count (ctl[4..5] start at 0)
    for (uint32_t w0 = blockIdx.x * 64u; w0 < n_waves_total; w0 += gridDim.x * 64u) {
        const uint32_t w = w0 + threadIdx.x;
        const bool in = w < n_waves_total;
        unsigned long long v = in ? wave_lines[w] : 0ull;
        uint32_t fidx = 0;
        if (n_files > 1 && in) {
            uint32_t lo_i = 0, hi_i = n_files;
            while (hi_i - lo_i > 1) { const uint32_t mid = (lo_i + hi_i) >> 1; if (files[mid].wave0 <= w) lo_i = mid; else hi_i = mid; }
            fidx = lo_i;
        }
        const uint32_t f_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)fidx);
        if (__builtin_amdgcn_ballot_w64(in && fidx != f_first) == 0) {                  // one file
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            if (threadIdx.x == 0 && v) atomicAdd((unsigned long long *)((n_files > 1 ? files[f_first].ctl : one.ctl) + 4), v);
        } else if (in && v) {
            atomicAdd((unsigned long long *)(files[fidx].ctl + 4), v);
        }
    }
}

}  // namespace

// Device scratch the site calling of a launch needs besides its records: the candidate list (16 bytes per entry: a shared half for
// the epilogue kernel and a stretch per wave for what a wave's LDS cannot hold until its tiles are done — a list that is full costs
// speed, not answers) and one line count per scan wave.
static const uint32_t VARSCAN_MAX_WAVES = 256 * 64;
static inline uint32_t varscan_cand_cap(uint64_t nbytes) {
    const uint64_t c = nbytes / 128 + 2 * VARSCAN_MAX_WAVES * 64;                      // (a shared half, and a stretch of >= 64 per wave)
    return (uint32_t)(c < (1ull << 23) ? c : (1ull << 23));
}
size_t snpgpu_varscan_scratch_bytes(uint64_t nbytes) {
    return (size_t)varscan_cand_cap(nbytes) * 16u + 4u * VARSCAN_MAX_WAVES + 1024;
}

// Deal the waves of a launch to its files in proportion to their sizes (every file at least one, every wave at least four tiles)
// and launch the scan and its epilogue.  `h_files`: abase / lo / hi / out / capacity / ctl / status set by the caller; n_tiles,
// wave0, n_waves are set here.  `d_files`: where the table is to go in device memory (n_files > 1; copied on the stream).
// d_cand_n: one zeroed word.
static int varscan_launch(snpgpu_ctx *ctx, VsFile *h_files, uint32_t n_files, VsFile *d_files, const snpgpu_varscan_params *prm, uint4 *d_cand,
                          uint32_t cand_cap, uint32_t *d_cand_n, uint32_t *d_wave_lines) {
    uint64_t total_tiles = 0;
    for (uint32_t i = 0; i < n_files; ++i) {
        h_files[i].n_tiles = h_files[i].hi / VS_TILE + 1;                               // (the tile of position hi, the virtual terminator, included)
        total_tiles += h_files[i].n_tiles;
    }
    uint32_t wg_waves = 12;                                                             // 12 x 13 616 bytes of LDS: one workgroup per CU, three waves per SIMD
    uint64_t resident = (uint64_t)ctx->n_cu * wg_waves;
    uint32_t mult = 1;                                                                  // (more workgroups than the CUs hold at once were measured and lose: every wave pays its prologue and a tile behind its run)
    uint32_t share[4] = {140, 100, 66, 66};                                             // tools/vs_share_sweep.sh: 123 -> 111 us per 30x sample from 118 : 100 : 84; steeper loses again
#ifdef SNPGPU_TUNING                                            // development builds only (tools/)
    if (const char *e = getenv("SNPGPU_VS_GRID_MUL")) if (atoi(e) > 0) mult = (uint32_t)atoi(e);
    uint32_t lds_waves = 12;                                    // (fewer waves per workgroup with the LDS of twelve: that many waves per CU)
    if (const char *e = getenv("SNPGPU_VS_WG_WAVES")) if (atoi(e) > 0 && atoi(e) <= 12) wg_waves = (uint32_t)atoi(e);
    if (const char *e = getenv("SNPGPU_VS_SHARE")) { int v[4]; if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) for (int k = 0; k < 4; ++k) share[k] = v[k] > 0 ? (uint32_t)v[k] : 1u; }
#endif
#ifdef SNPGPU_TUNING
    resident = (uint64_t)ctx->n_cu * wg_waves;
    const uint32_t lds_bytes = VS_LDS_WAVE * (wg_waves < lds_waves ? lds_waves : wg_waves);
#else
    const uint32_t lds_bytes = VS_LDS_WAVE * wg_waves;
#endif
    uint64_t want = resident * mult;
    if (want > VARSCAN_MAX_WAVES) want = VARSCAN_MAX_WAVES;
    if (want > total_tiles / 4) want = total_tiles / 4 ? total_tiles / 4 : 1;
    if (want < n_files) want = n_files;
    // Exactly `want` waves in all (a launch of one workgroup more than the CUs hold runs twice as long): every file the whole part
    // of its share, at least one; what is left over goes, one each, to the files in the order of their fractional parts.
    {
        std::vector<std::pair<long double, uint32_t>> frac(n_files);
        uint64_t given = 0;
        for (uint32_t i = 0; i < n_files; ++i) {
            const long double sh = (long double)want * (long double)h_files[i].n_tiles / (long double)total_tiles;
            uint64_t w = (uint64_t)sh;
            frac[i] = {sh - (long double)w, i};
            if (w < 1) { w = 1; frac[i].first = 0; }
            if (w > h_files[i].n_tiles) { w = h_files[i].n_tiles; frac[i].first = 0; }
            h_files[i].n_waves = (uint32_t)w;
            given += w;
        }
        std::sort(frac.begin(), frac.end(), [](const std::pair<long double, uint32_t> &a, const std::pair<long double, uint32_t> &b) {
            return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        for (uint32_t k = 0; k < n_files && given < want; ++k) {
            VsFile &f = h_files[frac[k].second];
            if (frac[k].first > 0 && f.n_waves < f.n_tiles) { ++f.n_waves; ++given; }
        }
    }
    uint32_t wave0 = 0;
    for (uint32_t i = 0; i < n_files; ++i) { h_files[i].wave0 = wave0; wave0 += h_files[i].n_waves; }
    const uint32_t n_waves_total = wave0;
    if (n_waves_total > 2 * VARSCAN_MAX_WAVES) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "too many pileups in one site-calling launch");
    const uint32_t grid = (n_waves_total + wg_waves - 1) / wg_waves;
    if (!ctx->varscan_lds_attr) {                                                       // (more than 64 KiB of dynamic LDS needs the permission, per device)
        (void)hipFuncSetAttribute((const void *)k_varscan_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_varscan_finish, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        ctx->varscan_lds_attr = true;
    }
    if (n_files > 1) HIP_TRY(ctx, hipMemcpyAsync(d_files, h_files, sizeof(VsFile) * n_files, hipMemcpyHostToDevice, ctx->stream));
    hipEvent_t ta = snpgpu_time_begin(ctx);
    // the list: its first half shared (the epilogue's), its second half one stretch per wave of this launch
    const uint32_t shared_cap = cand_cap / 2, spill_per_wave = (cand_cap - shared_cap) / n_waves_total;
    k_varscan_scan<<<grid, 64u * wg_waves, lds_bytes, ctx->stream>>>(d_files, n_files, h_files[0], n_waves_total, *prm, d_cand, shared_cap, d_cand_n,
                                                                      d_wave_lines, make_uint4(share[0], share[1], share[2], share[3]), d_cand + shared_cap,
                                                                      spill_per_wave);
    const uint32_t strip_bytes = 32u * 1024u;
    const uint32_t fin_grid = (uint32_t)ctx->n_cu * 4u;
    k_varscan_finish<<<fin_grid, 64, strip_bytes + 16u, ctx->stream>>>(d_files, n_files, h_files[0], *prm, d_cand, d_cand_n, shared_cap, d_wave_lines, n_waves_total, strip_bytes);
    snpgpu_time_end(ctx, SNPGPU_K_VARSCAN, ta);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

static inline void varscan_file_coords(VsFile &f, const uint8_t *d_buf, uint64_t nbytes) {
    const uint64_t shift = ((uintptr_t)d_buf & 15u) + 16u;                              // the aligned coordinates start 16..31 bytes below the file
    f.abase = d_buf - shift;
    f.lo = shift;
    f.hi = shift + nbytes;
}

// d_ctl: 8 zeroed words — [0] records found, [1] candidates on the global list, [4..5] lines of the file (written by the last
// kernel); d_status: one u64 preset to UINT64_MAX (becomes the offset of the first malformed line); d_scratch:
// snpgpu_varscan_scratch_bytes(nbytes) bytes, 16-byte aligned.
int snpgpu_enqueue_varscan(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, const snpgpu_varscan_params *prm, snpgpu_varscan_site *d_sites,
                           uint32_t capacity, uint32_t *d_ctl, uint64_t *d_status, void *d_scratch) {
    if (nbytes == 0) return SNPGPU_OK;
    if (nbytes >> 47) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "pileup too large");
    const uint32_t cand_cap = varscan_cand_cap(nbytes);
    VsFile f;
    memset(&f, 0, sizeof f);
    varscan_file_coords(f, d_buf, nbytes);
    f.out = d_sites; f.capacity = capacity; f.ctl = d_ctl; f.status = (unsigned long long *)d_status;
    return varscan_launch(ctx, &f, 1, nullptr, prm, (uint4 *)d_scratch, cand_cap, d_ctl + 1, (uint32_t *)((char *)d_scratch + (size_t)cand_cap * 16u));
}

// Many resident pileups in ONE launch.  d_ctl: n_files x 8 zeroed words (as above, [1] of file 0 is the launch's list counter);
// d_status: n_files u64 preset to UINT64_MAX; d_sites: n_files x capacity records; d_scratch: snpgpu_varscan_batch_scratch_bytes.
size_t snpgpu_varscan_batch_scratch_bytes(uint64_t total_bytes, uint32_t n_files) {
    return (size_t)varscan_cand_cap(total_bytes) * 16u + 4u * 2u * VARSCAN_MAX_WAVES + sizeof(VsFile) * (size_t)n_files + 1024;
}
size_t snpgpu_varscan_table_bytes(uint32_t n_files) { return sizeof(VsFile) * (size_t)(n_files ? n_files : 1); }
int snpgpu_enqueue_varscan_batch(snpgpu_ctx *ctx, const uint8_t *const *d_bufs, const uint64_t *nbytes, uint32_t n_files, const snpgpu_varscan_params *prm,
                                 snpgpu_varscan_site *d_sites, uint32_t capacity, uint32_t *d_ctl, uint64_t *d_status, void *d_scratch, void *h_table) {
    // (h_table: sizeof(VsFile) * n_files bytes of host memory that stay valid until the stream has copied them)
    if (n_files == 0) return SNPGPU_OK;
    if (n_files > 0xFFFFu) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "too many pileups in one site-calling launch");
    VsFile *h = (VsFile *)h_table;
    uint64_t total = 0;
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_files; ++i) {
        memset(&h[i], 0, sizeof h[i]);
        if (nbytes[i] >> 47) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "pileup too large");
        // (an empty file takes part with one tile: its only "line" starts behind its end and is not counted)
        varscan_file_coords(h[i], d_bufs[i] ? d_bufs[i] : (const uint8_t *)d_scratch, nbytes[i]);
        h[i].out = d_sites + (size_t)i * capacity; h[i].capacity = capacity; h[i].ctl = d_ctl + 8u * i; h[i].status = (unsigned long long *)(d_status + i);
        total += nbytes[i];
        ++n;
    }
    const uint32_t cand_cap = varscan_cand_cap(total);
    char *s = (char *)d_scratch;
    uint4 *d_cand = (uint4 *)s;
    uint32_t *d_wave_lines = (uint32_t *)(s + (size_t)cand_cap * 16u);
    VsFile *d_files = (VsFile *)(s + (size_t)cand_cap * 16u + 4u * 2u * VARSCAN_MAX_WAVES);
    // One launch per stretch of files of about 400 tiles per wave (twelve 30x samples of 5 Mbp): measured per sample, 110 us at 12
    // files per launch, 121 at 30, 135 at 100 — the longer a wave's run, the further the static shares of the waves drift from what
    // the SIMDs give them.  The launches follow each other on the stream and share the list and the line counts (a stretch's
    // epilogue has run before the next scan starts); each has its own list counter (word 1 of its first file's control words).
    uint64_t per_wave = 400;
#ifdef SNPGPU_TUNING
    if (const char *e = getenv("SNPGPU_VS_TILES_PER_WAVE")) if (atoi(e) > 0) per_wave = (uint64_t)atoi(e);
#endif
    const uint64_t per_launch = per_wave * (uint64_t)ctx->n_cu * 12ull;
    for (uint32_t i0 = 0; i0 < n;) {
        uint64_t tiles = 0;
        uint32_t i1 = i0;
        while (i1 < n && (i1 == i0 || tiles + h[i1].hi / VS_TILE + 1 <= per_launch + per_launch / 8)) { tiles += h[i1].hi / VS_TILE + 1; ++i1; }
        const int rc = varscan_launch(ctx, h + i0, i1 - i0, d_files + i0, prm, d_cand, cand_cap, d_ctl + 8u * i0 + 1, d_wave_lines);
        if (rc) return rc;
        i0 = i1;
    }
    return SNPGPU_OK;
}
