// K2: the consensus caller (one wavefront per site, and one lane per site for the throughput path), and the
// call_consensus entry points (K1, the scan, is scan.hip).
//
// Replaces, for a batch of samples:
//   pileup.Reader.__iter__            snppipeline/pileup.py:408-429   -> k_scan_wave (scan.hip)
//   pileup.Record._init_from_split_line  pileup.py:209-274            -> k_call_lanes / k_call_sites
//   pileup.Record._strip_unwanted_base_patterns  pileup.py:276-325    -> k_call_lanes / k_call_sites (mask algebra, see below)
//   pileup.ConsensusCaller.call_consensus  pileup.py:492-590          -> k_call_lanes / k_call_sites (tail)
//   call_consensus.py:161-188 (Region filter, '-' mapping, last line wins, missing -> '-')
//
// The scan leaves, per (sample, site), the offset + 1 of the last pileup line of that position (0: none).  k_call_sites
// runs one 64-lane wavefront per site over that line: the three regex passes of the reference become 64-bit mask
// algebra on wave ballots (scalar ALU work on gfx950: one ballot == one SGPR pair).  It is the complete caller
// (per-site counts for consensus.vcf, any symbol, any line length).  k_call_lanes gives every LANE a site and does the
// same algebra on per-lane bit masks; it serves the FASTA-only path and hands what it cannot do to k_call_sites.
#include <stdlib.h>

#include "internal.h"

// ------------------------------------------------------------------------------------------------
//                                   K2: one wavefront per site
// ------------------------------------------------------------------------------------------------
#define CALL_WAVES 4
#define CALL_LBUF 4224          // bytes of a line staged in LDS (lines longer than this take the serial path)
#define CALL_LMAX 2048          // longest bases field handled by the wave-parallel path
#define HIST_BINS 384           // [fwd | rev | neither] x 128 upper-cased symbols

struct CallArgs {
    const SampleDev *samples;   // the batch; outputs and site_line are [n_samples][n_sites] row-major
    uint32_t n_samples;
    const uint64_t *site_line;
    const uint8_t *site_flags;  // [n_sites], or one row per sample when flags_stride != 0 (per-sample exclude lists)
    uint32_t flags_stride;
    uint32_t n_sites;
    snpgpu_caller_params prm;
    uint8_t *out_base;
    uint8_t *out_filters;
    snpgpu_site_counts *out_counts;   // nullable
    uint64_t *todo;             // k_call_lanes: sites it leaves to the next kernel; k_call_sites: nullable, work only on these
    uint32_t *todo_n;
    const uint64_t *in_todo;    // k_call_lanes: nullable, work only on these sites
    const uint32_t *in_todo_n;
    const uint32_t *deep;       // k_call_lanes: *deep != 0: the batch's lines average more than 100 bytes — the 128-byte pass
                                // returns at once and the 256-byte pass takes every site (see k_call_mode)
    snpgpu_symbol_spill *spill; // k_call_sites, with out_counts: where ranks 8.. of a position with more symbols go (nullable)
    uint32_t *spill_n;          // records taken so far
    uint32_t spill_cap;         // records the arena holds
};

struct WaveLds {
    uint8_t line[CALL_LBUF];
    uint8_t s1[CALL_LMAX + 64];
    uint32_t hist[HIST_BINS];
    uint32_t fs[6], fe[6];
};

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint64_t low_mask(uint32_t n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }   // bits [0,n)
__device__ __forceinline__ uint32_t sat_add(uint32_t x, uint32_t y) { uint32_t s = x + y; return (s < x || s > (1u << 30)) ? (1u << 30) : s; }
__device__ __forceinline__ uint32_t sat_mul10_add(uint32_t x, uint32_t d) { return x >= (1u << 26) ? (1u << 30) : x * 10u + d; }

// histogram bin of a surviving base byte (after '.'/',' substitution): strand class * 128 + upper(byte)
__device__ __forceinline__ uint32_t hist_bin(uint32_t c) {
    uint32_t strand = c <= 0x5Au ? 0u : (c >= 0x61u ? 1u : 2u);      // pileup.py:269-270
    return strand * 128u + (to_upper(c) & 127u);             // bytes >= 0x80 are rejected by the scan
}

// Serial, streaming restatement of the same automaton for one lane: used for lines too long for the LDS path.
__device__ void call_serial(const uint8_t *g, uint64_t bs, uint64_t be, uint64_t qs, uint64_t qe, int minq,
                            uint32_t ref_up, uint32_t ref_lo, uint32_t *hist, uint32_t &good) {
    bool caret_skip = false, in_run = false;
    uint32_t pending = 0, acc = 0, debt = 0;
    uint64_t kept = 0, qlen = qe - qs;
    auto ordinary = [&](uint32_t c) {
        if (debt) { --debt; return; }
        if (c == '$') return;
        uint64_t i = kept++;
        if (i >= qlen) return;
        if ((int)g[qs + i] - 33 < minq) return;
        if (c == '.') c = ref_up; else if (c == ',') c = ref_lo;
        hist[hist_bin(c)] += 1;
        ++good;
    };
    for (uint64_t p = bs; p < be; ++p) {
        uint32_t c = g[p];
        if (caret_skip) { caret_skip = false; continue; }
        if (c == '^' && p + 1 < be) { caret_skip = true; continue; }
        if (pending) {
            uint32_t ps = pending;
            pending = 0;
            if (is_digit(c)) { in_run = true; acc = c - 48u; continue; }
            ordinary(ps);
        }
        if (in_run) {
            if (is_digit(c)) { acc = sat_mul10_add(acc, c - 48u); continue; }
            debt = sat_add(debt, acc);
            in_run = false;
        }
        if (c == '+' || c == '-') { pending = c; continue; }
        ordinary(c);
    }
    if (pending) ordinary(pending);
}

__global__ __launch_bounds__(CALL_WAVES * 64) void k_call_sites(CallArgs a) {
    __shared__ WaveLds lds[CALL_WAVES];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    WaveLds &L = lds[wave];
    const uint64_t lt = low_mask(lane);                     // lanes below me

    // Software pipeline over the wave's sites: the kernel is a chain of dependent memory round trips per site
    // (line offset -> line bytes), so the offset of the site after next and the first 192 bytes of the next site's line
    // (one round trip covers a typical 30x line) are requested before the current site is worked on.
    const uint64_t n_work = a.todo ? (uint64_t)*a.todo_n : (uint64_t)a.n_samples * a.n_sites;
    const uint64_t stride = (uint64_t)gridDim.x * CALL_WAVES;
    struct SiteRef { uint64_t site; uint64_t lv; uint32_t sflags; const uint8_t *buf; uint64_t nbytes; };
    auto fetch_ref = [&](uint64_t idx) -> SiteRef {
        SiteRef r{0, 0, 0, nullptr, 0};
        if (idx < n_work) {
            const uint64_t s = a.todo ? a.todo[idx] : idx;
            const uint32_t sample = (uint32_t)(s / a.n_sites);
            r.site = s;
            r.lv = a.site_line[s];
            r.sflags = a.site_flags[s - (uint64_t)sample * a.n_sites + (uint64_t)sample * a.flags_stride];
            r.buf = a.samples[sample].buf;
            r.nbytes = a.samples[sample].nbytes;
        }
        return r;
    };
    auto fetch_head = [&](const SiteRef &r, uint32_t &h0, uint32_t &h1, uint32_t &h2) {
        h0 = h1 = h2 = 10u;
        if (r.lv != 0) {
            const uint64_t ls = r.lv - 1;
            if (ls + lane < r.nbytes) h0 = r.buf[ls + lane];
            if (ls + 64 + lane < r.nbytes) h1 = r.buf[ls + 64 + lane];
            if (ls + 128 + lane < r.nbytes) h2 = r.buf[ls + 128 + lane];
        }
    };
    uint64_t idx = (uint64_t)blockIdx.x * CALL_WAVES + wave;
    SiteRef cur = fetch_ref(idx), nxt = fetch_ref(idx + stride);
    uint32_t pre0, pre1, pre2;
    fetch_head(cur, pre0, pre1, pre2);
    for (; idx < n_work; idx += stride) {
        const SiteRef after = fetch_ref(idx + 2 * stride);
        const uint64_t site = cur.site;
        uint32_t nh0, nh1, nh2;
        fetch_head(nxt, nh0, nh1, nh2);
        const uint8_t *buf = cur.buf;
        const uint64_t nbytes = cur.nbytes;
        const uint64_t lv = cur.lv;
        const uint32_t sflags = cur.sflags;
        uint32_t status = SNPGPU_ST_NO_LINE, filters = 0, cons = '-', out_b = '-';
        uint32_t raw_depth = 0, good = 0, nfwd = 0, nrev = 0, nsym = 0, ref = 0, ref_len = 1, ref_at = 0;
        long long depth64 = 0;
        bool have_hist = false;
        if (lv != 0) {
            const uint64_t ls = lv - 1;
            // ---- stage the line in LDS and tokenise it (fields 0..5) with ballots --------------
            for (uint32_t i = lane; i < HIST_BINS; i += 64) L.hist[i] = 0;
            if (lane < 6) { L.fs[lane] = 0xFFFFFFFFu; L.fe[lane] = 0xFFFFFFFFu; }
            uint32_t nfields = 0, line_len = 0;
            bool prev_ws = true;
            for (uint64_t k = 0;; k += 64) {
                uint64_t p = ls + k + lane;
                uint32_t c = k == 0 ? pre0 : (k == 64 ? pre1 : (k == 128 ? pre2 : (p < nbytes ? buf[p] : 10u)));
                if (k + lane < CALL_LBUF) L.line[k + lane] = (uint8_t)c;
                uint64_t T = __ballot(is_term(c));
                uint64_t valid = T ? low_mask((uint32_t)__ffsll((long long)T)) : ~0ull;   // up to and incl. the terminator
                uint64_t W = __ballot(is_ws(c)) & valid;
                uint64_t prevW = (W << 1) | (prev_ws ? 1ull : 0ull);
                uint64_t starts = ~W & prevW & valid;
                uint64_t ends = W & ~prevW & valid;
                uint32_t before = nfields + __popcll(starts & lt);
                if ((starts >> lane) & 1) { if (before < 6) L.fs[before] = (uint32_t)(k + lane); }
                if ((ends >> lane) & 1) { if (before >= 1 && before <= 6) L.fe[before - 1] = (uint32_t)(k + lane); }
                nfields += __popcll(starts);
                prev_ws = (W >> 63) & 1;
                if (T) { line_len = (uint32_t)(k + __ffsll((long long)T) - 1); break; }
                if (k + 64 > 0xFFFFFF00ull) break;
            }
            __builtin_amdgcn_wave_barrier();               // LDS is in order within a wave: no counter wait needed
            status = SNPGPU_ST_OK;
            const uint8_t *gl = buf + ls;                 // line bytes in global memory
            auto lb = [&](uint32_t off) -> uint32_t { return off < CALL_LBUF ? L.line[off] : (uint32_t)gl[off]; };
            if (nfields < 4) status = SNPGPU_ST_SHORT_LINE;
            else {
                ref_at = L.fs[2];
                ref_len = L.fe[2] - L.fs[2];                   // (a field has at least one byte)
                ref = lb(ref_at);
                uint32_t ds = L.fs[3], de = L.fe[3];
                PyInt di;                                    // int(depth), pileup.py:225 ("+30" and "3_0" are integers too)
                for (uint32_t q = ds; q < de; ++q) di.feed(lb(q));   // uniform loop, a handful of bytes
                // (a negative depth, or one past 2^32, is an integer for the reference as well — it only ever prints it and compares it
                // with 0: such a value travels in the position's spill record; past 2^62 it is refused)
                if (!di.ok() || di.v >= (1ull << 62)) status = SNPGPU_ST_BAD_DEPTH;
                else {
                    raw_depth = di.v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)di.v;
                    if (di.v > 0xFFFFFFFFull || (di.neg && di.v != 0)) depth64 = di.neg ? -(long long)di.v : (long long)di.v;
                    if (raw_depth != 0 && nfields == 5) status = SNPGPU_ST_NO_QUALS;
                }
            }
            if (status == SNPGPU_ST_OK && raw_depth != 0 && nfields >= 6) {
                have_hist = true;
                const uint32_t bs = L.fs[4], be = L.fe[4], qs = L.fs[5], qe = L.fe[5];
                const uint32_t L0 = be - bs, qlen = qe - qs;
                // A reference field of several bytes (pileup.py:223 takes any string): '.' and ',' are counted as themselves
                // here and spelled out afterwards
                const uint32_t ref_up = ref_len == 1 ? to_upper(ref) : (uint32_t)'.', ref_lo = ref_len == 1 ? to_lower(ref) : (uint32_t)',';
                const int minq = a.prm.min_base_quality;
                if (line_len <= CALL_LBUF && L0 <= CALL_LMAX) {
                    // ---- pass A: '^' + next byte (pileup.py:312).  Openers are the carets at even distance from
                    //      the start of their caret run; a run start is never itself consumed except through the
                    //      carry at a chunk boundary.  Run membership by parity class uses the carry chain of a
                    //      64-bit add. -----------------------------------------------------------------------
                    uint32_t n1 = 0;
                    bool carry_consumed = false;
                    for (uint32_t k = 0; k < L0; k += 64) {
                        uint32_t i = k + lane;
                        bool v = i < L0;
                        uint32_t c = v ? L.line[bs + i] : 0u;
                        uint64_t V = low_mask(L0 - k);
                        uint64_t C = __ballot(v && c == '^');
                        uint64_t S = C & ~(C << 1);
                        const uint64_t EVEN = 0x5555555555555555ull;
                        uint64_t Se = S & EVEN, So = S & ~EVEN;
                        if (carry_consumed && (C & 1ull)) { Se &= ~1ull; So |= 1ull; }   // run continues from the previous chunk with flipped parity
                        uint64_t Me = C & ~(C + Se), Mo = C & ~(C + So);
                        uint64_t openers = (Me & EVEN) | (Mo & ~EVEN);
                        if (carry_consumed && (C & 1ull) == 0) { /* byte 0 consumed, not a caret: nothing else changes */ }
                        uint64_t consumed = (openers << 1) | (carry_consumed ? 1ull : 0ull);
                        carry_consumed = (openers >> 63) & 1ull;
                        if (k + 64 >= L0) openers &= ~(1ull << (L0 - 1 - k));          // a trailing lone '^' stays
                        uint64_t keep = V & ~(openers | consumed);
                        if ((keep >> lane) & 1) L.s1[n1 + __popcll(keep & lt)] = (uint8_t)c;
                        n1 += __popcll(keep);
                    }
                    __builtin_amdgcn_wave_barrier();
                    // ---- pass B: indel markers with additive debt (pileup.py:315-320), '$' (pileup.py:323),
                    //      quality pairing (pileup.py:248-250), substitution + histogram (pileup.py:255-274) ----
                    uint32_t kept = 0, debt = 0, acc = 0;
                    bool in_run = false;
                    uint64_t carry_md = 0;
                    for (uint32_t k = 0; k < n1; k += 64) {
                        uint32_t i = k + lane;
                        bool v = i < n1;
                        uint32_t c = v ? L.s1[i] : 0u;
                        uint32_t nx = (i + 1 < n1) ? L.s1[i + 1] : 0u;
                        uint32_t nvalid = n1 - k < 64 ? n1 - k : 64;
                        uint64_t V = low_mask(nvalid);
                        uint64_t Ms = __ballot(v && (c == '+' || c == '-') && is_digit(nx));
                        uint64_t Dg = __ballot(v && is_digit(c));
                        uint64_t Del = 0, Md = 0;
                        if (Ms || in_run || debt) {
                            uint64_t St = ((Ms << 1) | carry_md) & Dg;
                            Md = Dg & ~(Dg + St);
                            uint32_t cur = 0;
                            while (true) {
                                if (in_run) {
                                    uint64_t rest = cur < 64 ? (~Md >> cur) : 1ull;
                                    uint32_t run_len = rest ? (uint32_t)__ffsll((long long)rest) - 1 : 64 - cur;
                                    for (uint32_t q = 0; q < run_len; ++q) acc = sat_mul10_add(acc, (uint32_t)L.s1[k + cur + q] - 48u);
                                    cur += run_len;
                                    if (cur >= nvalid && k + 64 < n1) break;            // run may continue in the next chunk
                                    debt = sat_add(debt, acc);
                                    acc = 0;
                                    in_run = false;
                                    if (cur >= nvalid) break;
                                }
                                uint64_t ms_rest = cur < 64 ? (Ms >> cur) : 0ull;
                                uint32_t m = ms_rest ? cur + (uint32_t)__ffsll((long long)ms_rest) - 1 : 64;
                                uint32_t seg_end = m < nvalid ? m : nvalid;
                                uint32_t seg_len = seg_end - cur;
                                uint32_t eat = debt < seg_len ? debt : seg_len;
                                Del |= low_mask(cur + eat) & ~low_mask(cur);
                                debt -= eat;
                                cur = m;
                                if (cur >= nvalid) break;
                                cur += 1;                                               // the sign
                                in_run = true;
                                acc = 0;
                            }
                        }
                        carry_md = ((Ms | Md) >> 63) & 1ull;
                        uint64_t keep = V & ~Ms & ~Md & ~Del & ~__ballot(c == '$');
                        uint32_t gi = kept + __popcll(keep & lt);
                        bool is_good = ((keep >> lane) & 1) && gi < qlen && ((int)L.line[qs + gi] - 33 >= minq);
                        kept += __popcll(keep);
                        if (c == '.') c = ref_up; else if (c == ',') c = ref_lo;
                        uint32_t bin = is_good ? hist_bin(c) : 0xFFFFu;
                        uint64_t rem = __ballot(is_good);
                        good += __popcll(rem);
                        while (rem) {                                                   // one LDS update per distinct bin
                            uint32_t leader = (uint32_t)__ffsll((long long)rem) - 1;
                            uint32_t kb = __builtin_amdgcn_readlane(bin, leader);
                            uint64_t same = __ballot(bin == kb);
                            if (lane == leader) L.hist[kb] += __popcll(same);
                            rem &= ~same;
                        }
                    }
                } else {
                    if (lane == 0) call_serial(gl, bs, be, qs, qe, minq, ref_up, ref_lo, L.hist, good);
                    good = __builtin_amdgcn_readfirstlane(good);
                }
                __builtin_amdgcn_wave_barrier();
                if (ref_len > 1) {
                    // bases_str.replace('.', ref.upper()).replace(',', ref.lower()) (pileup.py:255-258): every good '.' stands
                    // for all characters of the upper-cased field — and a ',' among THOSE is hit by the second replace too —,
                    // every good ',' for all characters of the lower-cased field.  good_depth stays the number of reads.
                    if (lane == 0) {
                        const uint32_t nd = L.hist[(uint32_t)'.'], nc = L.hist[(uint32_t)','];      // both bytes are <= 'Z': strand class 0
                        L.hist[(uint32_t)'.'] = 0; L.hist[(uint32_t)','] = 0;
                        // (one pass whatever the field's length: every ',' IN the field sends the nd '.' reads once more through the
                        // whole lower-cased field)
                        uint32_t commas = 0;
                        for (uint32_t i = 0; i < ref_len; ++i) commas += lb(ref_at + i) == (uint32_t)',' ? 1u : 0u;
                        for (uint32_t i = 0; i < ref_len; ++i) {
                            const uint32_t ch = lb(ref_at + i), u = to_upper(ch);
                            if (u != ',') L.hist[hist_bin(u)] += nd;
                            L.hist[hist_bin(to_lower(ch))] += nc + nd * commas;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }

        // ---- rank the histogram and apply the caller's filters (pileup.py:550-588) ---------------
        uint32_t t0 = 0, t1 = 0, f0 = 0, f1 = 0, r0 = 0, r1 = 0;
        if (have_hist) {
            f0 = L.hist[lane]; f1 = L.hist[lane + 64];
            r0 = L.hist[128 + lane]; r1 = L.hist[192 + lane];
            t0 = f0 + r0 + L.hist[256 + lane]; t1 = f1 + r1 + L.hist[320 + lane];
            nfwd = 0; nrev = 0;
            uint32_t sf = f0 + f1, sr = r0 + r1;
            for (int o = 32; o; o >>= 1) { sf += __shfl_xor(sf, o); sr += __shfl_xor(sr, o); }
            nfwd = sf; nrev = sr;
            nsym = __popcll(__ballot(t0 != 0)) + __popcll(__ballot(t1 != 0));
        }
        uint8_t top_sym[SNPGPU_MAX_SYMS];
        uint32_t top_t[SNPGPU_MAX_SYMS], top_f[SNPGPU_MAX_SYMS], top_r[SNPGPU_MAX_SYMS];
        const int want_top = a.out_counts ? SNPGPU_MAX_SYMS : 1;
#pragma unroll
        for (int r = 0; r < SNPGPU_MAX_SYMS; ++r) { top_sym[r] = 0; top_t[r] = top_f[r] = top_r[r] = 0; }
        if (status == SNPGPU_ST_OK) {
            if (good == 0) { filters = SNPGPU_F_RAWDPTH; cons = '-'; }
            else {
#pragma unroll
                for (int r = 0; r < SNPGPU_MAX_SYMS; ++r) {
                    if (r >= want_top) break;
                    // key: count descending, then byte ascending (pileup.py:265)
                    uint64_t k0 = t0 ? ((uint64_t)t0 << 8) | (255u - lane) : 0ull;
                    uint64_t k1 = t1 ? ((uint64_t)t1 << 8) | (255u - (lane + 64)) : 0ull;
                    uint64_t best = k0 > k1 ? k0 : k1;
                    for (int o = 32; o; o >>= 1) { uint64_t other = __shfl_xor((unsigned long long)best, o); best = other > best ? other : best; }
                    if (best == 0) break;
                    uint32_t sym = 255u - (uint32_t)(best & 255u);
                    uint32_t owner = sym & 63u;
                    uint32_t tf = sym < 64 ? f0 : f1, tr = sym < 64 ? r0 : r1;
                    top_sym[r] = (uint8_t)sym;
                    top_t[r] = (uint32_t)(best >> 8);
                    top_f[r] = __builtin_amdgcn_readlane(tf, owner);
                    top_r[r] = __builtin_amdgcn_readlane(tr, owner);
                    if (lane == owner) { if (sym < 64) t0 = 0; else t1 = 0; }
                }
                cons = top_sym[0];
                const uint32_t n = top_t[0], nf = top_f[0], nr = top_r[0];
                // CPython compares int < float exactly; one IEEE double multiply, no contraction (pileup.py:564, 580-582)
                if ((double)n < (double)good * a.prm.min_cons_freq) filters |= SNPGPU_F_VARFREQ;
                if ((int64_t)n < (int64_t)a.prm.min_cons_depth) filters |= SNPGPU_F_DEPTH;
                if ((int64_t)nf < (int64_t)a.prm.min_cons_strand_depth || (int64_t)nr < (int64_t)a.prm.min_cons_strand_depth) filters |= SNPGPU_F_STRDPTH;
                const double bias = (double)n * a.prm.min_cons_strand_bias;
                if ((double)nf < bias || (double)nr < bias) filters |= SNPGPU_F_STRBIAS;
                if (ref_len == 1 && cons == to_upper(ref)) cons = ref;                     // (a one-character base never equals a longer field)
            }
            // more symbols than the record keeps (consensus.vcf lists every one as an ALT allele), or a reference field of several
            // bytes (with or without good reads): ranks 8, 9, ... in the same order and the field go to a spill record of the
            // context; its index + 1 travels in the upper bits of n_symbols
            if (a.out_counts && (nsym > SNPGPU_MAX_SYMS || ref_len > 1 || depth64 != 0)) {
                // (a field of more than SNPGPU_SPILL_REF bytes goes on, raw, in the records that follow its own: claimed in one step)
                const uint32_t more_recs = ref_len > SNPGPU_SPILL_REF ? (ref_len - SNPGPU_SPILL_REF + (uint32_t)sizeof(snpgpu_symbol_spill) - 1) / (uint32_t)sizeof(snpgpu_symbol_spill) : 0u;
                uint32_t slot = 0xFFFFFFu;
                if (a.spill) {
                    if (lane == 0) slot = atomicAdd(a.spill_n, 1u + more_recs);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                }
                if (slot < a.spill_cap && more_recs < a.spill_cap - slot && slot + more_recs < 0xFFFFFEu) {
                    snpgpu_symbol_spill *sp = a.spill + slot;
                    uint32_t r = 0;
                    for (; r < SNPGPU_SPILL_SYMS; ++r) {
                        uint64_t k0 = t0 ? ((uint64_t)t0 << 8) | (255u - lane) : 0ull;
                        uint64_t k1 = t1 ? ((uint64_t)t1 << 8) | (255u - (lane + 64)) : 0ull;
                        uint64_t best = k0 > k1 ? k0 : k1;
                        for (int o = 32; o; o >>= 1) { uint64_t other = __shfl_xor((unsigned long long)best, o); best = other > best ? other : best; }
                        if (best == 0) break;
                        const uint32_t sym = 255u - (uint32_t)(best & 255u), owner = sym & 63u;
                        const uint32_t ef = __builtin_amdgcn_readlane(sym < 64 ? f0 : f1, owner), er = __builtin_amdgcn_readlane(sym < 64 ? r0 : r1, owner);
                        if (lane == 0) { sp->sym[r] = (uint8_t)sym; sp->total[r] = (uint32_t)(best >> 8); sp->fwd[r] = ef; sp->rev[r] = er; }
                        if (lane == owner) { if (sym < 64) t0 = 0; else t1 = 0; }
                    }
                    if (lane == 0) {
                        sp->n = r;
                        sp->ref_len = ref_len > 1 ? ref_len : 0u;
                        sp->depth64 = depth64;
                    }
                    if (ref_len > 1) {
                        // ref[] is the record's last member: the bytes from SNPGPU_SPILL_REF on land in the records claimed with it
                        uint8_t *dst = (uint8_t *)(sp + 1) - SNPGPU_SPILL_REF;
                        for (uint32_t i = lane; i < ref_len; i += 64)
                            dst[i] = (uint8_t)((ref_at + i) < CALL_LBUF ? (uint32_t)L.line[ref_at + i] : (uint32_t)buf[lv - 1 + ref_at + i]);
                    }
                    nsym |= (slot + 1u) << 8;
                } else {
                    nsym |= 0xFFFFFFu << 8;                   // no room (or no spill): the writer refuses this record
                }
            }
            if (sflags & SNPGPU_SITE_EXCLUDED) filters |= SNPGPU_F_REGION;
            out_b = (filters || cons == '*') ? '-' : cons;                                // call_consensus.py:169-176
        }
        if (lane == 0) {
            a.out_base[site] = (uint8_t)out_b;
            a.out_filters[site] = (uint8_t)filters;
            if (a.out_counts) {
                snpgpu_site_counts oc;
                oc.raw_depth = raw_depth; oc.good_depth = good; oc.fwd_good_depth = nfwd; oc.rev_good_depth = nrev;
                oc.n_symbols = nsym; oc.ref_base = (uint8_t)ref; oc.cons_base = (uint8_t)cons;
                oc.filters = (uint8_t)filters; oc.status = (uint8_t)status;
#pragma unroll
                for (int r = 0; r < SNPGPU_MAX_SYMS; ++r) { oc.sym[r] = top_sym[r]; oc.total[r] = top_t[r]; oc.fwd[r] = top_f[r]; oc.rev[r] = top_r[r]; }
                a.out_counts[site] = oc;
            } else if (status > SNPGPU_ST_OK) {
                a.out_filters[site] = (uint8_t)(0x80u | status);                          // error marker when no counts buffer
            }
        }
        cur = nxt;
        nxt = after;
        pre0 = nh0; pre1 = nh1; pre2 = nh2;
    }
}

// 64*N-bit per-lane masks for k_call_lanes (N = 2 or 4)
template <int N> struct BM { uint64_t w[N]; };
#define BM_FOR _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_)
template <int N> __device__ __forceinline__ BM<N> m_fill(uint64_t v) { BM<N> r; BM_FOR r.w[k_] = v; return r; }
template <int N> __device__ __forceinline__ BM<N> m_make(const uint32_t (&w)[2 * N]) { BM<N> r; BM_FOR r.w[k_] = (uint64_t)w[2 * k_] | ((uint64_t)w[2 * k_ + 1] << 32); return r; }
template <int N> __device__ __forceinline__ BM<N> m_and(BM<N> a, BM<N> b) { BM_FOR a.w[k_] &= b.w[k_]; return a; }
template <int N> __device__ __forceinline__ BM<N> m_or(BM<N> a, BM<N> b) { BM_FOR a.w[k_] |= b.w[k_]; return a; }
template <int N> __device__ __forceinline__ BM<N> m_andn(BM<N> a, BM<N> b) { BM_FOR a.w[k_] &= ~b.w[k_]; return a; }   // a & ~b
template <int N> __device__ __forceinline__ BM<N> m_not(BM<N> a) { BM_FOR a.w[k_] = ~a.w[k_]; return a; }
template <int N> __device__ __forceinline__ bool m_any(BM<N> a) { uint64_t o = 0; BM_FOR o |= a.w[k_]; return o != 0; }
template <int N> __device__ __forceinline__ BM<N> m_shl1(BM<N> a) { BM<N> r; BM_FOR r.w[k_] = (a.w[k_] << 1) | (k_ ? a.w[k_ ? k_ - 1 : 0] >> 63 : 0ull); return r; }
template <int N> __device__ __forceinline__ BM<N> m_add(BM<N> a, BM<N> b) {
    uint64_t c = 0;
    BM_FOR { const uint64_t s0 = a.w[k_] + b.w[k_], s1 = s0 + c; c = (s0 < a.w[k_] || s1 < s0) ? 1ull : 0ull; a.w[k_] = s1; }
    return a;
}
template <int N> __device__ __forceinline__ BM<N> m_below(uint32_t n) {                               // bits [0, n), n <= 64 N
    BM<N> r;
    BM_FOR r.w[k_] = n >= 64u * (k_ + 1) ? ~0ull : (n > 64u * k_ ? ((1ull << (n - 64u * k_)) - 1ull) : 0ull);
    return r;
}
template <int N> __device__ __forceinline__ BM<N> m_bit(uint32_t i) { BM<N> r; BM_FOR r.w[k_] = (i >> 6) == (uint32_t)k_ ? 1ull << (i & 63u) : 0ull; return r; }   // i >= 64 N: empty
template <int N> __device__ __forceinline__ uint64_t m_word(BM<N> a, uint32_t idx) { uint64_t r = a.w[0]; BM_FOR if (idx == (uint32_t)k_) r = a.w[k_]; return r; }
template <int N> __device__ __forceinline__ bool m_test(BM<N> a, uint32_t i) { return ((m_word(a, (i >> 6) & (N - 1)) >> (i & 63u)) & 1ull) != 0; }   // bit i mod 64 N
template <int N> __device__ __forceinline__ uint32_t m_ctz(BM<N> a) {                                 // 64 N: empty
    uint32_t r = 64u * N;
#pragma unroll
    for (int k = N - 1; k >= 0; --k) if (a.w[k]) r = 64u * k + (uint32_t)__builtin_ctzll(a.w[k]);
    return r;
}
template <int N> __device__ __forceinline__ uint32_t m_popc(BM<N> a) { uint32_t r = 0; BM_FOR r += (uint32_t)__popcll(a.w[k_]); return r; }
template <int N> __device__ __forceinline__ BM<N> m_clear_lowest(BM<N> a) {
    bool done = false;
    BM_FOR if (!done && a.w[k_]) { a.w[k_] &= a.w[k_] - 1; done = true; }
    return a;
}

// ------------------------------------------------------------------------------------------------
//                         K2 fast path: one LANE per site, 64 sites per wave
// ------------------------------------------------------------------------------------------------
// k_call_sites spends ~1000 wave instructions on one ~90-byte line.  Here every lane owns a site: the 256 aligned
// bytes around its line go to a private LDS slot (16-byte loads, slot stride 65 dwords: conflict free), then the lanes
// walk their lines in lockstep over aligned dwords — a serial tokeniser (str.split() field boundaries packed into two
// 64-bit registers) and the same left-to-right automaton as call_serial, with the per-symbol counts in byte lanes of
// two 64-bit registers ('*' A C G N T x forward/reverse; a line of <= 256 bytes cannot overflow them).  That is ~50
// wave instructions per site.  Whatever does not fit — a line longer than the window, a malformed line, any other
// symbol, a sign that is not an indel marker — is appended to a list for k_call_sites, which remains the reference
// implementation of the caller on the device (and the only one when per-site counts are requested).
// Two instantiations: a 256-byte window (lines of up to ~110x coverage; bases field <= 128 bytes, two mask words; two
// waves per workgroup) over all sites, then a 512-byte window (bases field <= 255 bytes, four mask words; one wave per
// workgroup because of the LDS it needs) over what the first one left, then k_call_sites over what that one left.
template <int kWin, int kWaves>
struct LanesLds {
    uint32_t slot[kWaves][64 * (kWin / 4 + 1)];   // slot stride kWin/4 + 1 dwords: odd, conflict free
    uint8_t cls[256];           // byte -> rank of its symbol among "*ACGNT" (6: '.' / ',') | reverse << 3; 0xFF: other
    uint4 hist[kWaves][64];     // per lane: sixteen byte counters, one per symbol class (bytes 0-7: forward strand, 8-15: reverse)
};

template <int kWin, int kWords, int kWaves, bool kCounts>
__global__ __launch_bounds__(kWaves * 64, kWin == 128 ? 3 : 1) void k_call_lanes(CallArgs a) {
    constexpr int LANES_WIN = kWin;               // bytes staged per site, from the 16-byte aligned address at or below the line start
    constexpr int LANES_STRIDE = kWin / 4 + 1;    // dwords between slots
    constexpr int LANES_WAVES = kWaves;
    constexpr uint32_t kBits = 64u * kWords;      // mask width = longest bases field in bytes ...
    constexpr uint32_t kMaxField = kBits > 255u ? 255u : kBits;   // ... but the counts live in byte lanes
    typedef BM<kWords> M;
    __shared__ LanesLds<kWin, kWaves> S;
    const bool deep = a.deep && *a.deep != 0;
    if (kWin == 128 && deep) return;                          // nearly every line would be pushed on after a wasted fetch
    if (kWin == 256 && deep) { a.in_todo = nullptr; a.in_todo_n = nullptr; }   // ... so this pass starts from all sites
    for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) {
        const uint32_t u = to_upper(c);
        uint32_t k = u == '*' ? 0u : u == 'A' ? 1u : u == 'C' ? 2u : u == 'G' ? 3u : u == 'N' ? 4u : u == 'T' ? 5u : 0xFFu;
        if (k != 0xFFu && c >= 0x61u) k |= 8u;               // pileup.py:269-270: >= 'a' is the reverse strand
        if (c == '.') k = 6u;                                // the reference base, forward strand (lane 6 of cnt_f)
        if (c == ',') k = 14u;                               // ... reverse strand (lane 6 of cnt_r)
        S.cls[c] = (uint8_t)k;
    }
    __syncthreads();
    const uint32_t lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *slot = &S.slot[wave][lane * LANES_STRIDE];
    const uint8_t *slot_b = (const uint8_t *)slot;
    const uint64_t n_work = a.in_todo ? (uint64_t)*a.in_todo_n : (uint64_t)a.n_samples * a.n_sites;
    const uint64_t n_groups = (n_work + 63) / 64;
    const int minq = a.prm.min_base_quality;

    // what a lane needs to know about its site; the next group's is requested before this group is worked on
    struct LaneRef { uint64_t site; uint64_t lv; uint32_t sflags; uintptr_t buf, end; };
    auto fetch_ref = [&](uint64_t g) -> LaneRef {
        LaneRef r{0, 0, 0, 0, 0};
        const uint64_t idx = g * 64 + lane;
        if (g < n_groups && idx < n_work) {
            const uint64_t s = a.in_todo ? a.in_todo[idx] : idx;
            const uint32_t sample = (uint32_t)(s / a.n_sites);
            r.site = s;
            r.lv = a.site_line[s];
            r.sflags = a.site_flags[s - (uint64_t)sample * a.n_sites + (uint64_t)sample * a.flags_stride];
            r.buf = (uintptr_t)a.samples[sample].buf;
            r.end = r.buf + a.samples[sample].nbytes;
        }
        return r;
    };
    const uint64_t g_stride = (uint64_t)gridDim.x * LANES_WAVES;
    uint64_t grp = (uint64_t)blockIdx.x * LANES_WAVES + wave;
    LaneRef nxt = fetch_ref(grp);
    for (; grp < n_groups; grp += g_stride) {
        const LaneRef cur = nxt;
        nxt = fetch_ref(grp + g_stride);
        const uint64_t site = cur.site;
        const bool valid = grp * 64 + lane < n_work;
        const uint64_t lv = cur.lv;
        const uint32_t sflags = cur.sflags;
        const uintptr_t addr = cur.buf + (lv - 1), end = cur.end;
        const bool has = lv != 0;
        // ---- stage the line: bytes [al, al + 128) of every lane first, [al + 128, al + 256) only for the lanes whose line
        //      does not end in the first half (a 30x line is ~100 bytes; the fetch is random 128-byte DRAM accesses, so
        //      bytes matter).  Bytes at or past the end of the file read as '\n'. ------------------------------------
#if defined(SNPGPU_TUNING) && defined(CALL_EXP_ALIGN128)       // (experiment build: every window inside ONE 128-byte cache line — what K2 would cost if
        const uintptr_t al = addr & ~(uintptr_t)127;           //  the matched lines lay in aligned compact slots; the results are wrong)
#else
        const uintptr_t al = addr & ~(uintptr_t)15;
#endif
        const uint32_t o = (uint32_t)(addr & 15);              // the line starts at byte o of the slot
        auto stage_half = [&](int half, bool want) {
            if (!__ballot(want && end - al < (uintptr_t)LANES_WIN + 16)) {
                // Eight neighbouring lanes fetch one site's 128 bytes, so that a wave instruction is 8 coalesced
                // 128-byte reads (2 cache lines each) instead of 64 scattered 16-byte ones: the fetch is bound by
                // the number of outstanding requests per CU, not by bytes.
                const uint64_t wm = __ballot(want);
                const uint32_t piece = lane & 7u;
                uint4 v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint32_t src = 8u * r + (lane >> 3);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)(uint32_t)al);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)(uint32_t)(al >> 32));
                    const uintptr_t x = (((uintptr_t)hi << 32) | lo) + 128u * half + 16u * piece;
                    v[r] = make_uint4(0, 0, 0, 0);
                    if ((wm >> src) & 1ull) v[r] = *(const uint4 *)x;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint32_t src = 8u * r + (lane >> 3);
                    if ((wm >> src) & 1ull) {
                        uint32_t *dst = &S.slot[wave][src * LANES_STRIDE + 32 * half + 4 * piece];
                        dst[0] = v[r].x; dst[1] = v[r].y; dst[2] = v[r].z; dst[3] = v[r].w;
                    }
                }
                return;
            }
            uint4 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uintptr_t x = al + 16u * (8 * half + r);
                v[r] = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
                if (want && x < end) v[r] = *(const uint4 *)x;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uintptr_t x = al + 16u * (8 * half + r);
                if (__ballot(want && x < end && x + 16 > end)) {           // the file ends inside this chunk (rare)
                    uint32_t w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
                    for (int d = 0; d < 4; ++d)
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb)
                            if (x + 4 * d + bb >= end) w[d] = (w[d] & ~(0xFFu << (8 * bb))) | (0x0Au << (8 * bb));
                    v[r] = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (want) {
                    uint32_t *dst = slot + 4 * (8 * half + r);
                    dst[0] = v[r].x; dst[1] = v[r].y; dst[2] = v[r].z; dst[3] = v[r].w;
                }
            }
        };
        stage_half(0, has);
        __builtin_amdgcn_wave_barrier();                       // LDS operations of a wave execute in order
#if defined(SNPGPU_TUNING) && defined(CALL_EXP) && CALL_EXP >= 2      // (... the staging alone)
        if (valid) a.out_filters[site] = (uint8_t)slot[0];
        continue;
#endif

        // ---- tokenise (pileup.py:206): per-lane 256-bit masks of the str.split() separators and of the terminator
        //      candidates, built 4 bytes per step with SWAR compares; the field boundaries are bit scans on the masks ----
        uint64_t Wm[LANES_WIN / 64], Tm[LANES_WIN / 64];
#pragma unroll
        for (int q = 0; q < LANES_WIN / 64; ++q) { Wm[q] = 0; Tm[q] = 0; }
        bool done = !has;
#pragma unroll
        for (int blk = 0; blk < LANES_WIN / 64; ++blk) {
            if (!__ballot(!done)) break;
            if (blk >= 2 && (blk & 1) == 0) { stage_half(blk / 2, !done); __builtin_amdgcn_wave_barrier(); }   // only the lines that are still open
            const bool open = !done;                             // (the slots of the others hold stale bytes up there)
            uint32_t mw[2] = {0, 0}, mt[2] = {0, 0};
#pragma unroll
            for (int jp = 0; jp < 8; ++jp) {                     // 8 bytes per step: the SWAR adds are 64-bit (v_lshl_add_u64)
                uint32_t w2[2] = {slot[16 * blk + 2 * jp], slot[16 * blk + 2 * jp + 1]};
                if (blk == 0 && jp < 2) {                        // bytes before the line start read as spaces
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t j = 2 * jp + h;
                        const uint32_t nb = o > 4u * j ? min(o - 4u * j, 4u) : 0u;
                        const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
                        w2[h] = (w2[h] & ~m) | (0x20202020u & m);
                    }
                }
                const uint64_t W = (uint64_t)w2[0] | ((uint64_t)w2[1] << 32);
                // bit 7 of a byte of W + (0x80 - t): byte >= t (input is ASCII)
                const uint64_t ge9 = W + 0x7777777777777777ull, ge14 = W + 0x7272727272727272ull, ge10 = W + 0x7676767676767676ull;
                const uint64_t ge28 = W + 0x6464646464646464ull, ge33 = W + 0x5F5F5F5F5F5F5F5Full;
                const uint64_t f_ws = ((ge9 & ~ge14) | (ge28 & ~ge33)) & 0x8080808080808080ull;        // 9..13, 28..32
                const uint64_t f_t = ge10 & ~ge14 & 0x8080808080808080ull;                              // 10..13
                const uint32_t bw = __builtin_amdgcn_udot4((uint32_t)(f_ws >> 32), 0x80402010u, __builtin_amdgcn_udot4((uint32_t)f_ws, 0x08040201u, 0u, false), false) >> 7;
                const uint32_t bt = __builtin_amdgcn_udot4((uint32_t)(f_t >> 32), 0x80402010u, __builtin_amdgcn_udot4((uint32_t)f_t, 0x08040201u, 0u, false), false) >> 7;
                mw[jp >> 2] |= bw << (8 * (jp & 3));
                mt[jp >> 2] |= bt << (8 * (jp & 3));
            }
            Wm[blk] = open ? (uint64_t)mw[0] | ((uint64_t)mw[1] << 32) : ~0ull;     // past the end of a line: separators
            Tm[blk] = open ? (uint64_t)mt[0] | ((uint64_t)mt[1] << 32) : 0ull;
            done = done || Tm[blk] != 0;
        }
        // first set bit at or after `pos` (LANES_WIN: none); inv: scan the complement
        auto scan_from = [&](const uint64_t (&Mk)[LANES_WIN / 64], bool inv, uint32_t pos) -> uint32_t {
            uint32_t r = LANES_WIN;
#pragma unroll
            for (int j = LANES_WIN / 64 - 1; j >= 0; --j) {
                uint64_t m = inv ? ~Mk[j] : Mk[j];
                const uint32_t pj = pos >> 6;
                m = pj > (uint32_t)j ? 0ull : (pj == (uint32_t)j ? m & (~0ull << (pos & 63u)) : m);
                if (m) r = 64u * j + (uint32_t)__builtin_ctzll(m);
            }
            return r;
        };
        const uint32_t line_end = scan_from(Tm, false, 0);      // candidates are bytes 10..13; 11 and 12 are checked below
        uint32_t nf = 0, f_s[6], f_e[6];
        {
            uint32_t pos = o;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t st = scan_from(Wm, true, pos);
                const uint32_t en = scan_from(Wm, false, st < (uint32_t)LANES_WIN ? st : LANES_WIN - 1u);
                const bool found = st < line_end;
                nf += found ? 1u : 0u;
                f_s[k] = st; f_e[k] = en;
                pos = found ? (en < (uint32_t)LANES_WIN ? en : LANES_WIN - 1u) : LANES_WIN - 1u;
            }
        }
        bool punt = has && line_end >= (uint32_t)LANES_WIN;    // no terminator inside the window: long line
        if (has && !punt) { const uint32_t tc = slot_b[line_end]; punt = tc == 11u || tc == 12u; }
        auto fs = [&](int k) -> uint32_t { return f_s[k]; };
        auto fe = [&](int k) -> uint32_t { return f_e[k]; };
        // reference base and depth (pileup.py:224-225); anything odd goes to k_call_sites, which knows the error codes
        uint32_t ref = 0, raw_depth = 0;
        if (has && !punt) {
            if (nf < 4 || nf == 5 || fe(2) - fs(2) != 1 || fe(3) - fs(3) > 9 || fe(3) <= fs(3)) punt = true;
            else {
                ref = slot_b[fs(2)];
                for (uint32_t q = fs(3); q < fe(3); ++q) {
                    const uint32_t c = slot_b[q];
                    if (!is_digit(c)) punt = true;
                    raw_depth = raw_depth * 10u + (c - 48u);
                }
            }
        }
        // ---- bases + qualities (pileup.py:237-274) -------------------------------------------------------------------
        // The bases field (<= kMaxField bytes here) is handled as per-lane 64*kWords-bit masks in field coordinates: byte classes by
        // SWAR compares + dot4, caret pairs by the carry chains of two 128-bit adds (as k_call_sites does on ballots),
        // indel markers by a short serial walk over the '+'/'-' bytes (none on most lines), and only the surviving bytes
        // are then looked at one by one for the counts.
        uint64_t cnt_f = 0, cnt_r = 0;                         // byte lanes: '*' A C G N T, lane 6: '.' (fwd) / ',' (rev)
        uint32_t good = 0;
        bool parse = has && !punt && raw_depth != 0 && nf >= 6;
        if (parse && fe(4) - fs(4) > kMaxField) { punt = true; parse = false; }
        {
            const uint32_t bs = parse ? fs(4) : 0u, L0 = parse ? fe(4) - fs(4) : 0u, qs = fs(5), qlen = parse ? fe(5) - fs(5) : 0u;
            uint32_t maxL = L0;
            for (int off = 32; off; off >>= 1) maxL = max(maxL, (uint32_t)__shfl_xor((int)maxL, off));
            maxL = __builtin_amdgcn_readfirstlane(maxL);
            const uint32_t nd = (maxL + 3) >> 2;                 // dwords of the longest field in the wave
            const M V = m_below<kWords>(L0);
            // -- byte classes: NOT '^', NOT sign, NOT '$' (a cleared bit = match), digits
            uint32_t nC[2 * kWords], nP[2 * kWords], nS[2 * kWords], dG[2 * kWords];
#pragma unroll
            for (int q = 0; q < 2 * kWords; ++q) { nC[q] = 0; nP[q] = 0; nS[q] = 0; dG[q] = 0; }
            {
                const uint32_t sh = bs & 3u;
                uint32_t lo = slot[bs >> 2];
#pragma unroll
                for (uint32_t j = 0; j < 16u * kWords; ++j) {
                    if (j >= nd) break;
                    const uint32_t hi = slot[(bs >> 2) + j + 1];
                    const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    lo = hi;
                    const uint32_t ne_c = ((w ^ 0x5E5E5E5Eu) + 0x7F7F7F7Fu) & 0x80808080u;
                    const uint32_t ne_p = ((w ^ 0x2B2B2B2Bu) + 0x7F7F7F7Fu) & ((w ^ 0x2D2D2D2Du) + 0x7F7F7F7Fu) & 0x80808080u;
                    const uint32_t ne_s = ((w ^ 0x24242424u) + 0x7F7F7F7Fu) & 0x80808080u;
                    const uint32_t dg = ((w + 0x50505050u) ^ (w + 0x46464646u)) & 0x80808080u;     // '0'..'9'
                    const uint32_t pos = (4 * j) & 31u;
                    nC[j >> 3] |= (__builtin_amdgcn_udot4(ne_c, 0x08040201u, 0u, false) >> 7) << pos;
                    nP[j >> 3] |= (__builtin_amdgcn_udot4(ne_p, 0x08040201u, 0u, false) >> 7) << pos;
                    nS[j >> 3] |= (__builtin_amdgcn_udot4(ne_s, 0x08040201u, 0u, false) >> 7) << pos;
                    dG[j >> 3] |= (__builtin_amdgcn_udot4(dg, 0x08040201u, 0u, false) >> 7) << pos;
                }
            }
            const M C = m_andn(V, m_make<kWords>(nC)), PM = m_andn(V, m_make<kWords>(nP)), DL = m_andn(V, m_make<kWords>(nS)), D = m_and(V, m_make<kWords>(dG));
            // -- '^' + next byte (pileup.py:312): openers are the carets at even distance from the start of their run
            const M EVEN = m_fill<kWords>(0x5555555555555555ull), ZERO = m_fill<kWords>(0ull);
            const M S0 = m_andn(C, m_shl1(C));
            const M Me = m_andn(C, m_add(C, m_and(S0, EVEN))), Mo = m_andn(C, m_add(C, m_andn(S0, EVEN)));
            M openers = m_or(m_and(Me, EVEN), m_andn(Mo, EVEN));
            openers = m_andn(openers, m_bit<kWords>(L0 - 1));            // a trailing lone '^' stays (L0 == 0: no bit)
            const M K1 = m_andn(V, m_or(openers, m_shl1(openers)));
            // -- indel markers with additive debt (pileup.py:315-320), walked sign by sign
            M Ms = ZERO, Md = ZERO, Del = ZERO;
            M P = m_and(PM, K1);
            if (__ballot(m_any(P))) {
                uint32_t debt = 0, prev_end = 0;
                auto settle = [&](uint32_t upto) {                 // the debt eats ordinary bytes in [prev_end, upto)
                    M O = m_and(m_andn(m_andn(K1, Ms), Md), m_andn(m_below<kWords>(upto), m_below<kWords>(prev_end)));
                    const uint32_t avail = m_popc(O), eat = debt < avail ? debt : avail;
                    M rest = O;
                    for (uint32_t t = 0; __ballot(t < eat); ++t) if (t < eat) rest = m_clear_lowest(rest);
                    Del = m_or(Del, m_andn(O, rest));
                    debt -= eat;
                };
                while (__ballot(m_any(P))) {
                    const bool on = m_any(P);
                    const uint32_t i = on ? m_ctz(P) : 0u;
                    if (on) P = m_clear_lowest(P);
                    const uint32_t nk = m_ctz(m_andn(K1, m_below<kWords>(i + 1 < kBits ? i + 1 : kBits)));            // the next byte that survived the carets
                    const bool marker = on && nk < kBits && m_test(D, nk);
                    const uint32_t e = m_ctz(m_andn(m_andn(K1, D), m_below<kWords>(nk)));      // kBits: the digits run to the end
                    const M run = m_and(m_and(K1, m_below<kWords>(e)), marker ? m_not(m_below<kWords>(nk)) : ZERO);
                    uint32_t count = 0;
                    {
                        M r = run;
                        while (__ballot(m_any(r))) {
                            if (m_any(r)) { count = sat_mul10_add(count, (uint32_t)slot_b[(bs + m_ctz(r)) & (LANES_WIN - 1u)] - 48u); r = m_clear_lowest(r); }
                        }
                    }
                    if (__ballot(marker && debt != 0)) { if (marker) settle(i); }
                    if (marker) {
                        Ms = m_or(Ms, m_bit<kWords>(i));
                        Md = m_or(Md, run);
                        // ordinary bytes between the previous marker and this sign were settled above (or there was no debt)
                        debt = sat_add(debt, count);
                        prev_end = e;
                    }
                }
                if (__ballot(debt != 0)) settle(kBits);
                punt = punt || m_any(m_andn(m_andn(m_and(PM, K1), Ms), Del));         // a sign that survives as a symbol: not ours
            }
            const M K = m_andn(m_andn(m_andn(m_andn(K1, Ms), Md), Del), DL);
            // -- pair with qualities (zip truncation, pileup.py:248-250) and count
            const int thr = 33 + minq;
            const uint32_t kept_total = m_popc(K);
            // all qualities pass and none is missing?  Then every kept byte is a good base.
            bool allq = kept_total <= qlen;
            if (thr > 0) {
                if (thr > 128) allq = allq && kept_total == 0;
                else {
                    const uint32_t addc = (uint32_t)(128 - thr) * 0x01010101u;       // bit 7 of a byte after the add: byte >= thr
                    const uint32_t shq = qs & 3u;
                    uint32_t lo = slot[(qs >> 2) & (LANES_WIN / 4 - 1u)];
                    uint32_t maxQ = parse ? (kept_total < qlen ? kept_total : qlen) : 0u;
                    for (int off = 32; off; off >>= 1) maxQ = max(maxQ, (uint32_t)__shfl_xor((int)maxQ, off));
                    const uint32_t nq = (__builtin_amdgcn_readfirstlane(maxQ) + 3) >> 2;
                    const uint32_t need = kept_total < qlen ? kept_total : qlen;      // qualities that matter
                    for (uint32_t j = 0; j < nq; ++j) {
                        const uint32_t hi = slot[((qs >> 2) + j + 1) & (LANES_WIN / 4 - 1u)];
                        const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, shq);
                        lo = hi;
                        const uint32_t nb = need > 4 * j ? min(need - 4 * j, 4u) : 0u;
                        const uint32_t m = nb >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nb)) - 1u));
                        allq = allq && ((~(w + addc)) & m) == 0;
                    }
                }
            }
            const bool pair_any = __ballot(parse && !allq) != 0;
            {
                const uint32_t sh = bs & 3u;
                uint32_t lo = slot[bs >> 2];
                uint32_t kept = 0;
                uint32_t qd = qs >> 2, qw = 0, qw_next = 0;
                if (pair_any) { qw = slot[qd & (LANES_WIN / 4 - 1u)]; qw_next = slot[(qd + 1) & (LANES_WIN / 4 - 1u)]; }
                // The counts live in LDS while the bases are walked: byte `cl` of the lane's sixteen takes one ds_add per read base
                // (a lane only ever touches its own sixteen bytes; <= 255 bases per field: no carry between them) instead of two
                // 64-bit shift-and-add pairs in registers — the loop is the kernel's largest block of vector instructions.
                uint32_t *my_hist = (uint32_t *)&S.hist[wave][lane];
                S.hist[wave][lane] = make_uint4(0u, 0u, 0u, 0u);
#if defined(SNPGPU_TUNING) && defined(CALL_EXP) && CALL_EXP >= 1      // (experiment builds: what the phases of the lane kernel cost; the results are wrong)
                for (uint32_t j = nd; j < nd; ++j) {
#else
                for (uint32_t j = 0; j < nd; ++j) {
#endif
                    const uint32_t hi = slot[(bs >> 2) + j + 1];
                    const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    lo = hi;
                    const uint32_t k4 = (uint32_t)(m_word(K, j >> 4) >> ((4 * j) & 63u)) & 15u;
                    uint32_t cl4[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) cl4[k] = S.cls[(w >> (8 * k)) & 0xFFu];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const bool emit = (k4 >> k) & 1u;
                        bool goodb = emit;
                        if (pair_any) {
                            const uint32_t qi = qs + kept;
                            if (__ballot((qi >> 2) != qd)) {
                                const bool adv = (qi >> 2) != qd;
                                qw = adv ? qw_next : qw;
                                qd += adv ? 1u : 0u;
                                qw_next = slot[(qd + 1) & (LANES_WIN / 4 - 1u)];
                            }
                            const uint32_t qv = (qw >> (8 * (qi & 3u))) & 0xFFu;
                            goodb = emit && kept < qlen && (int)qv >= thr;
                            kept += emit ? 1u : 0u;
                        }
                        const uint32_t cl = cl4[k];
                        // (cl 0xFF — any other symbol — lands in byte 15, which no class uses: such a site is handed on below)
                        (void)__hip_atomic_fetch_add(my_hist + ((cl >> 2) & 3u), (goodb ? 1u : 0u) << (8u * (cl & 3u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
                const uint4 h = S.hist[wave][lane];
                punt = punt || (h.w >> 24) != 0u;
                cnt_f = (uint64_t)h.x | ((uint64_t)h.y << 32);
                cnt_r = (uint64_t)h.z | ((uint64_t)(h.w & 0x00FFFFFFu) << 32);
                // good bases = all that were counted (pileup.py:252)
                good = __builtin_amdgcn_udot4(h.x, 0x01010101u, __builtin_amdgcn_udot4(h.y, 0x01010101u, 0u, false), false) +
                       __builtin_amdgcn_udot4(h.z, 0x01010101u, __builtin_amdgcn_udot4(h.w, 0x00010101u, 0u, false), false);
            }
            // '.' and ',' stand for the reference base on the forward / reverse strand (pileup.py:255-258)
            const uint32_t ndot = (uint32_t)(cnt_f >> 48) & 0xFFu, ncom = (uint32_t)(cnt_r >> 48) & 0xFFu;
            const uint32_t kr = S.cls[to_upper(ref) & 0xFFu];
            const bool ref_ok = kr >= 1u && kr <= 5u;             // A C G N T in either case
            punt = punt || (!ref_ok && (ndot | ncom) != 0);
            cnt_f = (cnt_f & 0x0000FFFFFFFFFFFFull) + ((uint64_t)ndot << (8 * (kr & 7u)));
            cnt_r = (cnt_r & 0x0000FFFFFFFFFFFFull) + ((uint64_t)ncom << (8 * (kr & 7u)));
        }
        // ---- the caller (pileup.py:550-588) --------------------------------------------------------------------------
        uint32_t filters = 0, cons = '-';
        if (has && !punt) {
            if (good == 0) filters = SNPGPU_F_RAWDPTH;
            else {
                const uint64_t tot = cnt_f + cnt_r;            // per byte lane <= 253: no carry between lanes
                uint32_t n = 0, bk = 0;
#pragma unroll
                for (uint32_t k = 0; k < 6; ++k) {             // count descending, byte ascending (pileup.py:265)
                    const uint32_t tk = (uint32_t)(tot >> (8 * k)) & 0xFFu;
                    if (tk > n) { n = tk; bk = k; }
                }
                cons = bk == 0 ? '*' : bk == 1 ? 'A' : bk == 2 ? 'C' : bk == 3 ? 'G' : bk == 4 ? 'N' : 'T';
                const uint32_t nfw = (uint32_t)(cnt_f >> (8 * bk)) & 0xFFu, nrv = (uint32_t)(cnt_r >> (8 * bk)) & 0xFFu;
                if ((double)n < (double)good * a.prm.min_cons_freq) filters |= SNPGPU_F_VARFREQ;
                if ((int64_t)n < (int64_t)a.prm.min_cons_depth) filters |= SNPGPU_F_DEPTH;
                if ((int64_t)nfw < (int64_t)a.prm.min_cons_strand_depth || (int64_t)nrv < (int64_t)a.prm.min_cons_strand_depth) filters |= SNPGPU_F_STRDPTH;
                const double bias = (double)n * a.prm.min_cons_strand_bias;
                if ((double)nfw < bias || (double)nrv < bias) filters |= SNPGPU_F_STRBIAS;
                if (cons == to_upper(ref)) cons = ref;
            }
            if (sflags & SNPGPU_SITE_EXCLUDED) filters |= SNPGPU_F_REGION;
        }
        if (valid && !punt) {
            a.out_base[site] = (uint8_t)((filters || cons == '*') ? '-' : cons);           // call_consensus.py:169-176
            a.out_filters[site] = (uint8_t)filters;
            if (kCounts) {
                // the per-site record (consensus.vcf): everything pileup.Record exposes, ranked by count descending, byte
                // ascending (pileup.py:263-266) — the byte lanes are already in byte order: '*' A C G N T
                const uint64_t tot = cnt_f + cnt_r;
                uint32_t syms[2] = {0, 0}, tt[8], tf[8], tr[8], nsym = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) { tt[r] = 0; tf[r] = 0; tr[r] = 0; }
                uint64_t left = tot;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    uint32_t n = 0, bk = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 6; ++k) {
                        const uint32_t tk = (uint32_t)(left >> (8 * k)) & 0xFFu;
                        if (tk > n) { n = tk; bk = k; }
                    }
                    if (n != 0) {
                        const uint32_t sym = bk == 0 ? '*' : bk == 1 ? 'A' : bk == 2 ? 'C' : bk == 3 ? 'G' : bk == 4 ? 'N' : 'T';
                        syms[r >> 2] |= sym << (8 * (r & 3));
                        tt[r] = n;
                        tf[r] = (uint32_t)(cnt_f >> (8 * bk)) & 0xFFu;
                        tr[r] = (uint32_t)(cnt_r >> (8 * bk)) & 0xFFu;
                        left &= ~(0xFFull << (8 * bk));
                        ++nsym;
                    }
                }
                const uint32_t nfwd = __builtin_amdgcn_udot4((uint32_t)cnt_f, 0x01010101u, 0u, false) + __builtin_amdgcn_udot4((uint32_t)(cnt_f >> 32), 0x00000101u, 0u, false);
                const uint32_t nrev = __builtin_amdgcn_udot4((uint32_t)cnt_r, 0x01010101u, 0u, false) + __builtin_amdgcn_udot4((uint32_t)(cnt_r >> 32), 0x00000101u, 0u, false);
                uint4 *rec = (uint4 *)&a.out_counts[site];
                const uint32_t st = has ? (uint32_t)SNPGPU_ST_OK : (uint32_t)SNPGPU_ST_NO_LINE;
                rec[0] = make_uint4(has ? raw_depth : 0u, good, nfwd, nrev);
                rec[1] = make_uint4(nsym, (has ? ref : 0u) | (cons << 8) | (filters << 16) | (st << 24), syms[0], syms[1]);
                rec[2] = make_uint4(tt[0], tt[1], tt[2], tt[3]); rec[3] = make_uint4(tt[4], tt[5], tt[6], tt[7]);
                rec[4] = make_uint4(tf[0], tf[1], tf[2], tf[3]); rec[5] = make_uint4(tf[4], tf[5], tf[6], tf[7]);
                rec[6] = make_uint4(tr[0], tr[1], tr[2], tr[3]); rec[7] = make_uint4(tr[4], tr[5], tr[6], tr[7]);
            }
        }
        // ---- leftovers: one atomic per wave ---------------------------------------------------------------------------
        const uint64_t pm = __ballot(punt);
        if (pm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(a.todo_n, (uint32_t)__popcll(pm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (punt) a.todo[base + __popcll(pm & low_mask(lane))] = site;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// One workgroup, before the call kernels: is this a batch of deep pileups?  The scan has just counted the lines of every
// sample; mean line length = bytes / lines.  (Every lane kernel pushes what it cannot do on a list with one atomic per
// 64 sites — same-address atomics cost ~12 ns each — so a pass that would push nearly everything on is skipped as a whole.)
__global__ __launch_bounds__(256) void k_call_mode(const SampleDev *samples, uint32_t n, uint32_t *deep) {
    __shared__ unsigned long long part[2][4];
    unsigned long long bytes = 0, lines = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { bytes += samples[i].nbytes; lines += samples[i].status[1]; }
    for (int o = 32; o; o >>= 1) { bytes += __shfl_xor(bytes, o); lines += __shfl_xor(lines, o); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = bytes; part[1][threadIdx.x >> 6] = lines; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bytes = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        lines = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        *deep = bytes > 100ull * lines ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------------
//                                           host API
// ------------------------------------------------------------------------------------------------
struct SampleIO { const uint8_t *d_pileup; size_t nbytes; uint64_t *d_status; };

// The call kernels over n_sites "sites" of n samples: d_table describes the (complete) files, d_site_line the line of every
// (sample, site); default_flags [n_sites] unless d_site_flags gives every sample its own row.
static int enqueue_call_n(snpgpu_ctx *ctx, uint32_t n_sites, const uint8_t *default_flags, const SampleDev *d_table, uint32_t n,
                          const snpgpu_caller_params *prm, const uint64_t *d_site_line, uint8_t *d_out_base,
                          uint8_t *d_out_filters, snpgpu_site_counts *d_out_counts, uint32_t *d_todo_n, uint64_t *d_todo,
                          uint64_t *d_todo2, const uint8_t *d_site_flags, uint32_t flags_stride) {
    hipStream_t st = ctx->stream;
    if (!n_sites || !n) return SNPGPU_OK;
    CallArgs ca;
    ca.samples = d_table;
    ca.n_samples = n;
    ca.site_line = d_site_line;
    ca.site_flags = d_site_flags ? d_site_flags : default_flags;
    ca.flags_stride = d_site_flags ? flags_stride : 0;
    ca.n_sites = n_sites;
    ca.prm = *prm;
    ca.out_base = d_out_base;
    ca.out_filters = d_out_filters;
    ca.out_counts = d_out_counts;
    ca.spill = d_out_counts ? ctx->d_spill : nullptr;
    ca.spill_n = ctx->d_spill_n;
    ca.spill_cap = ctx->spill_cap;
    const uint64_t n_work = (uint64_t)n_sites * n;
    const uint64_t blocks = (n_work + CALL_WAVES - 1) / CALL_WAVES;
    const uint64_t max_blocks = (uint64_t)ctx->n_cu * 16;
    const unsigned grid = (unsigned)(blocks < max_blocks ? blocks : max_blocks);
    ca.todo = nullptr;
    ca.todo_n = nullptr;
    ca.in_todo = nullptr;
    ca.in_todo_n = nullptr;
    ca.deep = nullptr;
    hipEvent_t ta = snpgpu_time_begin(ctx);
    {
        // One lane per site, in three window sizes: 128 bytes (64-bit masks: half the registers and a quarter of the LDS of
        // the next one, so twice the waves per SIMD — a 30x line is ~90 bytes) for every site, 256 bytes (bases field <= 128)
        // for what that left, 512 bytes (bases field <= 255) for what THAT left, then one wave per site for the rest.
        // With per-site records (consensus.vcf) the same chain writes them too.  d_todo_n: three leftover counts.
        const uint64_t groups = (n_work + 63) / 64;
        k_call_mode<<<1, 256, 0, st>>>(d_table, n, d_todo_n + 3);
        ca.deep = d_todo_n + 3;
        ca.todo = d_todo;
        ca.todo_n = d_todo_n;
        const uint64_t b0 = (groups + 3) / 4, m0 = (uint64_t)ctx->n_cu * 16;
        const unsigned g0 = (unsigned)(b0 < m0 ? b0 : m0);
        if (d_out_counts) k_call_lanes<128, 1, 4, true><<<g0, 256, 0, st>>>(ca); else k_call_lanes<128, 1, 4, false><<<g0, 256, 0, st>>>(ca);
        ca.in_todo = d_todo;
        ca.in_todo_n = d_todo_n;
        ca.todo = d_todo2;
        ca.todo_n = d_todo_n + 1;
        const uint64_t lblocks = (groups + 1) / 2, lmax = (uint64_t)ctx->n_cu * 8;
        const unsigned g1 = (unsigned)(lblocks < lmax ? lblocks : lmax);
        if (d_out_counts) k_call_lanes<256, 2, 2, true><<<g1, 128, 0, st>>>(ca); else k_call_lanes<256, 2, 2, false><<<g1, 128, 0, st>>>(ca);
        ca.in_todo = d_todo2;
        ca.in_todo_n = d_todo_n + 1;
        ca.todo = d_todo;                                   // the first list has been consumed
        ca.todo_n = d_todo_n + 2;
        const uint64_t l2max = (uint64_t)ctx->n_cu * 4;
        const unsigned g2 = (unsigned)(groups < l2max ? groups : l2max);
        if (d_out_counts) k_call_lanes<512, 4, 1, true><<<g2, 64, 0, st>>>(ca); else k_call_lanes<512, 4, 1, false><<<g2, 64, 0, st>>>(ca);
        k_call_sites<<<grid, CALL_WAVES * 64, 0, st>>>(ca);
    }
    snpgpu_time_end(ctx, SNPGPU_K_CALL, ta);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// The call kernels over a scanned batch: d_table describes the (complete) files, d_site_line the scan's result.
int snpgpu_enqueue_call(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n,
                        const snpgpu_caller_params *prm, const uint64_t *d_site_line, uint8_t *d_out_base,
                        uint8_t *d_out_filters, snpgpu_site_counts *d_out_counts, uint32_t *d_todo_n, uint64_t *d_todo,
                        uint64_t *d_todo2, const uint8_t *d_site_flags, uint32_t flags_stride) {
    return enqueue_call_n(ctx, ss->n_sites, ss->dev.flags, d_table, n, prm, d_site_line, d_out_base, d_out_filters, d_out_counts, d_todo_n, d_todo, d_todo2,
                          d_site_flags, flags_stride);
}

// Every line of one pileup as a "site" (--vcfAllPos, call_consensus.py:148 / pileup.py:418-421): the caller over a list of line
// offsets — the same chain as over a site list (one lane per line in three window sizes, one wave per line for what is left; up to
// round 4 every line took a wave of its own: 9.9 ms for 5 M lines).  d_todo / d_todo2: 8 * n_lines bytes each; d_todo_n: four words.
int snpgpu_enqueue_call_lines(snpgpu_ctx *ctx, const SampleDev *d_sample, const uint64_t *d_line_off, const uint8_t *d_flags,
                              uint32_t n_lines, const snpgpu_caller_params *prm, uint8_t *d_out_base, uint8_t *d_out_filters,
                              snpgpu_site_counts *d_out_counts, uint32_t *d_todo_n, uint64_t *d_todo, uint64_t *d_todo2) {
    if (!n_lines) return SNPGPU_OK;
    if (d_todo_n && d_todo && d_todo2) {
        HIP_TRY(ctx, hipMemsetAsync(d_todo_n, 0, 16, ctx->stream));
        return enqueue_call_n(ctx, n_lines, d_flags, d_sample, 1, prm, d_line_off, d_out_base, d_out_filters, d_out_counts, d_todo_n, d_todo, d_todo2, nullptr, 0);
    }
    CallArgs ca;
    ca.samples = d_sample;
    ca.n_samples = 1;
    ca.site_line = d_line_off;
    ca.site_flags = d_flags;
    ca.flags_stride = 0;
    ca.n_sites = n_lines;
    ca.prm = *prm;
    ca.out_base = d_out_base;
    ca.out_filters = d_out_filters;
    ca.out_counts = d_out_counts;
    ca.spill = d_out_counts ? ctx->d_spill : nullptr;
    ca.spill_n = ctx->d_spill_n;
    ca.spill_cap = ctx->spill_cap;
    ca.todo = nullptr; ca.todo_n = nullptr; ca.in_todo = nullptr; ca.in_todo_n = nullptr; ca.deep = nullptr;
    const uint64_t blocks = ((uint64_t)n_lines + CALL_WAVES - 1) / CALL_WAVES, max_blocks = (uint64_t)ctx->n_cu * 16;
    hipEvent_t ta = snpgpu_time_begin(ctx);
    k_call_sites<<<(unsigned)(blocks < max_blocks ? blocks : max_blocks), CALL_WAVES * 64, 0, ctx->stream>>>(ca);
    snpgpu_time_end(ctx, SNPGPU_K_CALL, ta);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// Scan + call for up to SNPGPU_SCAN_MAX_BATCH resident samples: one scan launch and one call launch for all of them.
// d_site_line == nullptr: the rows live in the context's scratch; outputs are [n][n_sites] row-major.
static int enqueue_group(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleIO *io, uint32_t n,
                         const snpgpu_caller_params *prm, uint8_t *d_out_base, uint8_t *d_out_filters,
                         snpgpu_site_counts *d_out_counts, uint64_t *d_site_line, int want_depth,
                         const uint8_t *d_site_flags = nullptr, uint32_t flags_stride = 0) {
    const uint32_t n_sites = ss->n_sites;
    const size_t ws_bytes = (snpgpu_scan_workspace_bytes(ctx, n) + 255) / 256 * 256;
    const size_t rows_bytes = d_site_line ? 0 : 8ull * n_sites * n;
    const size_t list_bytes = (8ull * n_sites * n + 255) / 256 * 256;
    const size_t todo_bytes = 2 * list_bytes + 256;                             // leftovers of the two lane kernels + their counts
    void *ws = nullptr;
    {
        int rc = snpgpu_scratch(ctx, ws_bytes + rows_bytes + todo_bytes + 256, &ws);
        if (rc) return rc;
    }
    if (!d_site_line) d_site_line = (uint64_t *)((char *)ws + ws_bytes);
    uint32_t *d_todo_n = (uint32_t *)((char *)ws + ws_bytes + rows_bytes);      // leftovers after the 128-, 256- and 512-byte passes
    uint64_t *d_todo = (uint64_t *)((char *)ws + ws_bytes + rows_bytes + 256);
    uint64_t *d_todo2 = (uint64_t *)((char *)ws + ws_bytes + rows_bytes + 256 + list_bytes);
    std::vector<SampleDev> samples(n);
    for (uint32_t i = 0; i < n; ++i) {
        samples[i] = SampleDev{};
        samples[i].buf = io[i].d_pileup;
        samples[i].nbytes = io[i].nbytes;
        samples[i].status = io[i].d_status;
    }
    // the scan's prepare kernel also zeroes the line-offset rows and the two leftover counters of the lane kernels
    int rc = snpgpu_enqueue_scan(ctx, ss, samples, ws, d_site_line, want_depth, d_todo_n, 3);
    if (rc) return rc;
    return snpgpu_enqueue_call(ctx, ss, (const SampleDev *)ws, n, prm, d_site_line, d_out_base, d_out_filters, d_out_counts,
                               d_todo_n, d_todo, d_todo2, d_site_flags, flags_stride);
}

static int enqueue_sample(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const uint8_t *d_pileup, size_t nbytes,
                          const snpgpu_caller_params *prm, uint8_t *d_out_base, uint8_t *d_out_filters,
                          snpgpu_site_counts *d_out_counts, uint64_t *d_status, int want_depth) {
    SampleIO io{d_pileup, nbytes, d_status};
    // the single-sample forms keep the line offsets in the site set, where snpgpu_siteset_line_offsets finds them
    return enqueue_group(ctx, ss, &io, 1, prm, d_out_base, d_out_filters, d_out_counts, ss->site_line, want_depth);
}

extern "C" {

int snpgpu_call_consensus_dev(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const void *d_pileup, size_t nbytes,
                              const snpgpu_caller_params *params, uint8_t *d_out_base, uint8_t *d_out_filters,
                              snpgpu_site_counts *d_out_counts, uint64_t *d_status, int want_depth_sum) {
    if (!ctx || !ss || !params || !d_status || (nbytes && !d_pileup)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (ss->n_sites && (!d_out_base || !d_out_filters)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    if (d_out_counts) { int rc = snpgpu_spill_begin(ctx); if (rc) return rc; }
    return enqueue_sample(ctx, ss, (const uint8_t *)d_pileup, nbytes, params, d_out_base, d_out_filters, d_out_counts, d_status, want_depth_sum);
}

int snpgpu_siteset_line_offsets(snpgpu_ctx *ctx, const snpgpu_siteset *ss, uint64_t *out_line_off) {
    if (!ctx || !ss || (ss->n_sites && !out_line_off)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (!ss->n_sites) return SNPGPU_OK;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(out_line_off, ss->site_line, 8ull * ss->n_sites, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SNPGPU_OK;
}

int snpgpu_call_consensus_batch_dev(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const void *d_pileups,
                                    const uint64_t *h_offsets, const uint64_t *h_sizes, uint32_t n_samples,
                                    const snpgpu_caller_params *params, uint8_t *d_out_base,
                                    uint8_t *d_out_filters, uint64_t *d_status) {
    if (!ctx || !ss || !params || !d_status || !h_offsets) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    const uint8_t *p = (const uint8_t *)d_pileups;
    if (!h_sizes)
        for (uint32_t i = 0; i < n_samples; ++i)
            if (h_offsets[i + 1] < h_offsets[i]) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "offsets must be non-decreasing");
    // groups of samples share one scan launch and one call launch; the line-offset rows of a group (8 bytes per
    // sample and site) live in scratch, so the group is also bounded by 1 GiB of rows
    uint32_t group = SNPGPU_SCAN_MAX_BATCH;
    if (ss->n_sites) {
        const uint64_t by_mem = (1ull << 30) / (8ull * ss->n_sites);
        if (by_mem < group) group = by_mem ? (uint32_t)by_mem : 1;
    }
    std::vector<SampleIO> io;
    for (uint32_t i0 = 0; i0 < n_samples; i0 += group) {
        const uint32_t n = n_samples - i0 < group ? n_samples - i0 : group;
        io.resize(n);
        for (uint32_t k = 0; k < n; ++k)
            io[k] = SampleIO{p + h_offsets[i0 + k], (size_t)(h_sizes ? h_sizes[i0 + k] : h_offsets[i0 + k + 1] - h_offsets[i0 + k]),
                             d_status + (size_t)(i0 + k) * SNPGPU_SCAN_STATUS_WORDS};
        const size_t off = (size_t)i0 * ss->n_sites;
        int rc = enqueue_group(ctx, ss, io.data(), n, params, d_out_base + off, d_out_filters + off, nullptr, nullptr, 0);
        if (rc) return rc;
    }
    return SNPGPU_OK;
}

// The same for samples that live anywhere in device memory (the resident pileups of the one-job pipeline), with everything
// the per-sample CLI gets from the streamed form: per-site records, the line offsets, and optionally one row of site flags
// per sample (a sample's own exclude list, call_consensus.py:117-123, 165-168).
int snpgpu_call_consensus_many_dev(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const void *const *d_pileups, const uint64_t *h_sizes,
                                   uint32_t n_samples, const snpgpu_caller_params *params, const uint8_t *d_site_flags,
                                   uint8_t *d_out_base, uint8_t *d_out_filters, snpgpu_site_counts *d_out_counts,
                                   uint64_t *d_out_line_off, uint64_t *d_status, int want_depth_sum) {
    if (!ctx || !ss || !params || !d_status || (n_samples && (!d_pileups || !h_sizes))) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null argument");
    if (ss->n_sites && n_samples && (!d_out_base || !d_out_filters)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null output");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    if (d_out_counts) { int rc = snpgpu_spill_begin(ctx); if (rc) return rc; }
    uint32_t group = SNPGPU_SCAN_MAX_BATCH;
    if (ss->n_sites) {                                          // per group in scratch: two leftover lists (+ the rows when not given)
        const uint64_t by_mem = (1ull << 30) / (8ull * ss->n_sites);
        if (by_mem < group) group = by_mem ? (uint32_t)by_mem : 1;
    }
    std::vector<SampleIO> io;
    for (uint32_t i0 = 0; i0 < n_samples; i0 += group) {
        const uint32_t n = n_samples - i0 < group ? n_samples - i0 : group;
        io.resize(n);
        for (uint32_t k = 0; k < n; ++k) {
            if (h_sizes[i0 + k] && !d_pileups[i0 + k]) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null pileup %u", i0 + k);
            io[k] = SampleIO{(const uint8_t *)d_pileups[i0 + k], (size_t)h_sizes[i0 + k], d_status + (size_t)(i0 + k) * SNPGPU_SCAN_STATUS_WORDS};
        }
        const size_t off = (size_t)i0 * ss->n_sites;
        int rc = enqueue_group(ctx, ss, io.data(), n, params, d_out_base + off, d_out_filters + off, d_out_counts ? d_out_counts + off : nullptr,
                               d_out_line_off ? d_out_line_off + off : nullptr, want_depth_sum, d_site_flags ? d_site_flags + off : nullptr, ss->n_sites);
        if (rc) return rc;
    }
    return SNPGPU_OK;
}

}  // extern "C"
