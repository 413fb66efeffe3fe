// K1: pileup scan — newline index, "chrom pos" parse of every line, site match.
//
// Replaces pileup.Reader.__iter__ with a position set (snppipeline/pileup.py:422-429): for every line of the
// genome-wide pileup, split on whitespace, take (chrom, int(pos)) and test membership in the site set.  A matching
// line publishes (file offset + 1) with atomicMax into site_line[site]; the maximum implements "the last duplicate
// line wins" (call_consensus.py:171-176).
//
// One launch serves a batch of pileups; every wavefront is an independent stream over a contiguous run of 4 KiB tiles
// of ONE sample (waves dealt by the host, shares weighted in the kernel by the wave's age rank on its SIMD and the sample's line density).  A wave owns two LDS slots,
// requests tile k+2 with LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B straight into LDS, no VGPR round trip) as
// soon as tile k is parsed, and waits for its own DMA with a counted s_waitcnt — no workgroup barrier, no flags, no
// polling.  The DMA goes through inline asm on purpose: hipcc orders every LDS read behind a *tracked* LDS-DMA with
// vmcnt(0), which would serialise fetch and parse; the steady-state parse issues no other vector-memory operation (the
// site bitmap is probed through a register window; matched lines collect in LDS and their atomics go out right after a
// top-of-tile wait — vmcnt counts loads and atomics alike, and an atomic issued in the middle of a tile made the next counted wait
// stand until it had been to memory and back), so the explicit counted wait is the only one on the path.
// Per tile:
//   B  each lane scans the four 16-byte chunks of ITS 64 contiguous bytes for bytes in 0x0A..0x0D (SWAR, v_dot4_u32_u8 gathers byte
//      flags into bits; the chunks are read in an order that keeps ds_read_b128 free of bank conflicts); two flagged neighbours
//      (CR LF) are told apart by reading them, so that the '\r' flags nothing.  There is no list of line starts and no prefix sum
//      (rounds 1-3 built one in LDS): the terminators a lane found are the lines it parses
//   C  one lane per line, one line per lane and round: the line behind the lowest terminator bit the lane has left (30x: one round
//      per tile, now and then a second; 8x / 15x: two).  One-window form: with the name length L and the digit count g of the positions known, the
//      24 bytes that end with the separator after the position hold "\n name SEP digits SEP" at fixed places — masked
//      compares (the '\n' in front is what makes the flagged start a line start), SWAR digit test, dot4 decimal
//      conversion; a round in which every line fits does no other bookkeeping.  A start that does not follow '\n' is
//      looked at bytewise: after "\r\n" split over two chunks, '\v' or '\f' there is no line; after a lone '\r' there is.  General form (calibrates g, handles long names): name
//      window, digit window, SWAR "<= 0x20" mask + ffs for the digit count.  Then the site probe by ds_bpermute into a
//      64-dword register window of the bitmap / rank directory.
// A line of ANOTHER CONTIG than the wave's hint waits (one 64-bit mask per lane) until the tile's rounds are through; then the hint
// moves to that contig and a second trip through the same LDS slot takes the lines that waited (round 6: up to then they went to the
// queue below, 6 000 lines of every 200-contig sample).
// A line that fits neither form (odd whitespace, > 10 digits, names > 44 bytes, a second contig change in one tile) is
// pushed on a device queue and finished by k_scan_queue with an exact byte-wise parser; if the queue overflows, the
// kExact instantiation (every line through the exact parser) redoes the batch.  kExact also serves the depth-column
// sum.  Line and match counts leave the kernel through per-wave slots (k_scan_finish adds them up per sample):
// same-address atomics from thousands of waves cost ~12 ns each and stall the loads behind them.
#include <stdlib.h>

#include "internal.h"
#include "prims.h"

#define SCAN_TILE SNPGPU_SCAN_TILE            // bytes per wave tile (4096)
#define SCAN_HALO SNPGPU_SCAN_HALO            // bytes staged past the tile for the first fields of its last lines (128)
#ifndef SCAN_NBUF
#define SCAN_NBUF 2                          // LDS slots per wave: tile k is parsed while tiles k+1 .. k+NBUF-1 stream in
#endif
#define SCAN_WTILE_CHUNKS ((16 + SCAN_TILE + SCAN_HALO) / 16)   // [t0-16, t0+TILE+HALO) in 16-byte chunks
#define SCAN_DMA_PER_TILE ((SCAN_WTILE_CHUNKS + 63) / 64)       // global_load_lds wave-instructions per tile
#define SCAN_LIST_CAP 192                    // line starts held in LDS per pass (a tile with more makes extra passes)
#define SCAN_HINT_WORDS 12                   // contig names up to 44 bytes take the fast compare
#define SCAN_HIT_CAP 128                     // matched lines a wave collects in LDS before it publishes them (what LDS is left at sixteen waves per CU)
#ifndef SCAN_HIT_FLUSH
#define SCAN_HIT_FLUSH 48                    // ... published at the top of the next tile once there are this many
#endif

struct ScanArgs {
    const SampleDev *samples; // one launch covers a batch of pileups; each wave works inside exactly one of them
    uint32_t n_samples;
    uint32_t n_sites;
    uint64_t *site_line;     // [n_samples][n_sites]
    uint64_t *queue;         // (sample << 40 | file offset) of lines left to the exact parser
    uint32_t *ctl;           // [0] queue length, [1] queue overflowed
    uint32_t q_cap;
    int want_depth;
    uint32_t share[4];       // relative tile share of a wave by its age rank on its SIMD (wave-in-block / 4): files of long lines
    uint32_t share_dense[4]; // ... files of short lines (a wave measures the line density of its sample itself, below)
    uint64_t *totals;        // per-wave {lines, matched, depth sum}: same-address atomics from thousands of waves
                             // serialise at ~12 ns each and stall the loads of the waves still running
    unsigned long long *dbg; // optional: per-wave timing records (tuning only)
};
// the part of a sample's description the parse loop works with (wave-uniform)
struct ScanFile {
    const uint8_t *base;     // 16-byte aligned pointer at or below the first byte of the file
    uint64_t lo, hi;         // the file is base[lo, hi)
    uint64_t *site_line;     // this sample's row
    uint64_t *status;        // this sample's SNPGPU_SCAN_STATUS_WORDS
    uint64_t sample;         // index << 40, OR-ed into queue entries
};
__device__ __forceinline__ ScanFile scan_file(const ScanArgs &a, uint32_t i) {
    const SampleDev sd = a.samples[i];
    ScanFile f;
    const uintptr_t addr = (uintptr_t)sd.buf;
    f.base = (const uint8_t *)(addr & ~(uintptr_t)15);
    f.lo = addr & 15;
    f.hi = f.lo + sd.nbytes;
    f.site_line = a.site_line + (size_t)i * a.n_sites;
    f.status = sd.status;
    f.sample = (uint64_t)i << 40;
    return f;
}

struct WaveSlots {
    uint4 slot[SCAN_NBUF][SCAN_WTILE_CHUNKS];
    uint16_t lstart[SCAN_LIST_CAP];
    uint32_t hint_w[SCAN_HINT_WORDS];        // the wave's current contig name, zero padded
    uint32_t hint_m[SCAN_HINT_WORDS];        // byte masks of the name (zero past its end)
    uint32_t lay[16];                        // scan_layout() of the current (name, digit count)
    uint32_t hit_site[SCAN_HIT_CAP];         // matched lines not yet published: the site ...
    uint32_t hit_off[SCAN_HIT_CAP];          // ... and the line's file offset + 1, counted from hit_base
};
struct ScanShared {                          // dynamic LDS: the table, then one WaveSlots per wave of the workgroup
    uint4 digit_mask[16];                    // [nd]: keeps the last nd bytes of window bytes 4..14
    WaveSlots w[1];
};
#if SCAN_NBUF == 2
static_assert(sizeof(ScanShared) + 15 * sizeof(WaveSlots) <= 160 * 1024, "sixteen waves of k_scan_wave no longer fit a CU's LDS");
#endif

__device__ __forceinline__ void report_scan_error(uint64_t *status, uint64_t file_off, uint32_t code) {
    atomicMin((unsigned long long *)&status[0], (unsigned long long)(((file_off + 1) << 8) | code));
}

struct TileView {
    const uint8_t *tile;     // LDS copy, valid for [-16, lds_limit)
    const uint8_t *base;     // global
    uint64_t t0, hi;
    int64_t lds_limit;
    __device__ __forceinline__ uint32_t get(int64_t p) const {
        if (p < lds_limit) return tile[p];
        uint64_t ab = t0 + (uint64_t)p;
        return ab < hi ? base[ab] : 10u;
    }
};

// The contig table of a site set, handed to the helpers below BY VALUE: a reference to the kernel's site-set argument would make
// every wave keep a copy of that argument in scratch memory (4.6 KB written per wave at its start).
struct ContigTab {
    const uint8_t *names;
    const uint32_t *name_off;
    const uint64_t *bit_off;
    const uint32_t *max_pos;
    uint32_t n_contigs;
};
__device__ __forceinline__ ContigTab contig_tab(const SiteSetDev &ss) { return ContigTab{ss.names, ss.name_off, ss.bit_off, ss.max_pos, ss.n_contigs}; }

// Bytewise lexicographic compare of a line field with contig name c.
__device__ int cmp_name(const ContigTab &ss, uint32_t c, const TileView &tv, int64_t p0, uint32_t len) {
    uint32_t a = ss.name_off[c], nl = ss.name_off[c + 1] - a;
    uint32_t m = len < nl ? len : nl;
    for (uint32_t k = 0; k < m; ++k) {
        int d = (int)tv.get(p0 + k) - (int)ss.names[a + k];
        if (d) return d;
    }
    return (int)len - (int)nl;
}

__device__ __noinline__ uint32_t find_contig(ContigTab ss, TileView tv, int64_t f0, uint32_t f0len) {
    int lo_i = 0, hi_i = (int)ss.n_contigs - 1;
    while (lo_i <= hi_i) {
        int mid = (lo_i + hi_i) >> 1;
        int d = cmp_name(ss, (uint32_t)mid, tv, f0, f0len);
        if (d == 0) return (uint32_t)mid;
        if (d < 0) hi_i = mid - 1; else lo_i = mid + 1;
    }
    return 0xFFFFFFFFu;
}

// Exact byte-wise parse of "chrom pos [ref depth]" (any whitespace, any length).
struct SlowLine {
    uint32_t err;            // 0 or SCAN_ERR_*
    uint32_t f0len;
    int64_t f0;
    uint64_t pos;
    unsigned long long depth;
};
__device__ __noinline__ SlowLine parse_line_slow(TileView tv, int64_t s, int want_depth) {
    SlowLine r;
    r.err = 0; r.f0len = 0; r.f0 = s; r.pos = 0; r.depth = 0;
    int64_t p = s;
    uint32_t c = tv.get(p);
    while (is_ws(c) && !is_term(c)) c = tv.get(++p);
    if (is_term(c)) { r.err = SCAN_ERR_FEW_FIELDS; return r; }
    const int64_t f0 = p;
    while (!is_ws(c)) c = tv.get(++p);
    const uint32_t f0len = (uint32_t)(p - f0);
    while (is_ws(c) && !is_term(c)) c = tv.get(++p);
    if (is_term(c)) { r.err = SCAN_ERR_FEW_FIELDS; return r; }
    PyInt pi;                                                   // int(pos), pileup.py:426: "+5" and "1_0" are integers too
    while (!is_ws(c)) { pi.feed(c); c = tv.get(++p); }
    if (!pi.ok()) { r.err = SCAN_ERR_BAD_POS; return r; }
    // a negative position, or one past 2^32 - 1, cannot be in the site set
    const uint64_t pos = (pi.neg && pi.v != 0) || pi.v > 0xFFFFFFFFull ? 0x100000000ull : pi.v;
    r.f0 = f0; r.f0len = f0len; r.pos = pos;
    if (want_depth) {                                       // 4th column, collect_metrics.py:325-340 by-product
        while (is_ws(c) && !is_term(c)) c = tv.get(++p);
        while (!is_ws(c)) c = tv.get(++p);                  // reference base field
        while (is_ws(c) && !is_term(c)) c = tv.get(++p);
        PyInt di;                                               // int(tokens[3]) or skip the line (collect_metrics.py:330-333)
        while (!is_ws(c)) { di.feed(c); c = tv.get(++p); }
        if (di.ok()) r.depth = di.neg ? 0ull - di.v : di.v;     // sums are taken modulo 2^64: a negative depth subtracts
    }
    return r;
}

__device__ __forceinline__ uint32_t four_digits(uint32_t x) {      // x: 4 bytes 0..9, lowest address most significant
    uint32_t t = (x * 10u + (x >> 8)) & 0x00FF00FFu;
    return (t & 0xFFu) * 100u + (t >> 16);
}

// 0x80 in every byte of w that is <= 0x20 (ASCII input)
__device__ __forceinline__ uint32_t le20_flags(uint32_t w) { return ~(w + 0x5F5F5F5Fu) & 0x80808080u; }
// 16 byte flags (0x80 per byte) of a 16-byte chunk -> 16 bits (v_dot4_u32_u8 with weights 1,2,4,8 / 16..128)
__device__ __forceinline__ uint32_t flags_to_bits16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3) {
    uint32_t lo = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(f1, 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(f3, 0x80402010u, hi, false);
    return (lo >> 7) | ((hi >> 7) << 8);
}

// 16 bytes of LDS starting at any byte offset, as four dwords.  A byte-misaligned ds_read_b128 is executed as slowly
// as sixteen byte reads, so read five ALIGNED dwords and funnel-shift them (v_alignbyte_b32).
__device__ __forceinline__ void lds_window16(const uint8_t *tile, int off, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
    const uint32_t *pw = (const uint32_t *)(tile + (off & ~3));
    const uint32_t a0 = pw[0], a1 = pw[1], a2 = pw[2], a3 = pw[3], a4 = pw[4];
    const uint32_t sh = (uint32_t)off & 3u;
    o0 = __builtin_amdgcn_alignbyte(a1, a0, sh);
    o1 = __builtin_amdgcn_alignbyte(a2, a1, sh);
    o2 = __builtin_amdgcn_alignbyte(a3, a2, sh);
    o3 = __builtin_amdgcn_alignbyte(a4, a3, sh);
}

// 24 bytes of LDS starting at any byte offset (may be negative), as six dwords
__device__ __forceinline__ void lds_window24(const uint8_t *tile, int off, uint32_t (&o)[6]) {
    const uint32_t *pw = (const uint32_t *)(tile + (off & ~3));
    uint32_t a[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) a[k] = pw[k];
    const uint32_t sh = (uint32_t)off & 3u;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = __builtin_amdgcn_alignbyte(a[k + 1], a[k], sh);
}

// What the one-window parse expects to see in the 24 bytes that END with the separator after the position, for a contig
// name of L bytes and positions of g digits whose first g - 4 digits are those of `top` (zero padded; none for g <= 4):
// [.. junk ..]['\n'][name, L][TAB][top digits, g - 4][last four digits][TAB].  lay[0..5] name bytes, the two TABs, the top digits and
// (when it fits: L + g <= 21) the '\n' before the line in place, lay[6..11] their byte masks, lay[12] the byte mask of the last
// min(g, 4) digits in the dword made of window bytes 19..22.  One lane per dword.
// (Only TAB-separated lines take the one-window parse — what samtools writes; a line with other whitespace there goes
// through the general parse or the exact parser.)
__device__ __noinline__ void scan_layout(uint32_t *lay, const uint32_t *hint_w, uint32_t L, uint32_t g, uint32_t top, uint32_t lane) {
    if (lane < 6) {
        uint32_t w = 0, m = 0;
        for (uint32_t k = 0; k < L; ++k) {
            const uint32_t b = 22u - g - L + k;                   // window byte of name byte k
            if ((b >> 2) == lane) {
                w |= ((hint_w[k >> 2] >> (8 * (k & 3))) & 0xFFu) << (8 * (b & 3));
                m |= 0xFFu << (8 * (b & 3));
            }
        }
        const uint32_t seps[2] = {22u - g, 23u};                  // the separators after the name and after the digits
        for (int q = 0; q < 2; ++q)
            if ((seps[q] >> 2) == lane) { w |= 9u << (8 * (seps[q] & 3)); m |= 0xFFu << (8 * (seps[q] & 3)); }
        if (L + g <= 21u) {                                       // the byte before the line: '\n' (the start is a proper one)
            const uint32_t pb = 21u - g - L;
            if ((pb >> 2) == lane) { w |= 10u << (8 * (pb & 3)); m |= 0xFFu << (8 * (pb & 3)); }
        }
        uint32_t t = top;                                         // the top digits, last one first: window bytes 18, 17, ... 23 - g
        for (uint32_t b = 18u; g > 4u && b >= 23u - g; --b) {
            if ((b >> 2) == lane) { w |= (0x30u + t % 10u) << (8 * (b & 3)); m |= 0xFFu << (8 * (b & 3)); }
            t /= 10u;
        }
        lay[lane] = w;
        lay[6 + lane] = m;
    }
    if (lane == 6) lay[12] = g >= 4u ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8u * (4u - g));
}

struct Hint {                // the wave's current contig (all members wave-uniform); the name itself is in LDS
    uint32_t len, cid, max_pos;
    uint64_t bit_off;
};

__device__ __noinline__ Hint load_hint(ContigTab ss, uint32_t cid, uint32_t *hint_w, uint32_t lane) {
    Hint h;
    h.cid = 0xFFFFFFFFu; h.len = 0; h.max_pos = 0; h.bit_off = 0;
    if (cid < ss.n_contigs) {
        const uint32_t off = ss.name_off[cid], len = ss.name_off[cid + 1] - off;
        h.cid = cid; h.len = len; h.max_pos = ss.max_pos[cid]; h.bit_off = ss.bit_off[cid];
        if (lane < SCAN_HINT_WORDS) {
            uint32_t w = 0;
            for (uint32_t j = 0; j < 4; ++j) { uint32_t i = lane * 4 + j; if (i < len) w |= (uint32_t)ss.names[off + i] << (8 * j); }
            hint_w[lane] = w;
        }
    } else if (lane < SCAN_HINT_WORDS) hint_w[lane] = 0;
    return h;
}

// 0x80 in every byte of w in 0x0A..0x0D ('\n' '\v' '\f' '\r'; ASCII input).  Three ops per dword; whether a flagged
// byte really is '\n' is checked when its line start is emitted.
__device__ __forceinline__ uint32_t term_flags(uint32_t w) { return (w + 0x76767676u) & ~(w + 0x72727272u) & 0x80808080u; }

#define SCAN_HINT_NONE 0xFFFFFFFFu           // no usable hint
#define SCAN_HINT_ABSENT 0xFFFFFFFEu         // the name is known NOT to be a contig of the site set

template <bool kExact, int kTime>
__global__ __launch_bounds__(1024) void k_scan_wave(ScanArgs a, SiteSetDev ss) {
    extern __shared__ uint4 scan_lds[];
    ScanShared &sh = *(ScanShared *)scan_lds;
    constexpr bool kDepth = kTime == 4;                      // also add up the depth column (collect_metrics by-product)
    if (kExact && a.ctl[1] == 0) return;                     // fallback pass: only when the slow-line queue overflowed
    if ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) >= a.samples[a.n_samples].wave0) return;   // spare wave
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveSlots &ws = sh.w[wave];
    // Phase B: a lane owns 64 CONTIGUOUS bytes of the tile (chunks 4 * lane .. + 3) and reads them in the order i ^ (quad & 3), so
    // that the sixteen lanes of every ds_read_b128 lane group hit sixteen different 16-byte slots; bit 16 * i + b of its
    // terminator word is byte b of the i-th chunk it READ, which is byte byte_of(16 * i + b) of the tile.
    const uint32_t k16 = ((lane >> 2) & 3u) << 4;
    auto byte_of = [&](uint32_t bit) -> uint32_t { return (lane << 6) | (bit ^ k16); };
    // the 16-bit groups of a terminator word into address order and back (the same exchange both ways): neighbours swap where
    // k16 & 16 (rotate both dwords by 16), the dwords swap where k16 & 32
    auto regroup = [&](uint64_t x) -> uint64_t {
        const uint32_t rot = k16 & 16u, lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t a_lo = __builtin_amdgcn_alignbit(lo, lo, rot), a_hi = __builtin_amdgcn_alignbit(hi, hi, rot);
        return (k16 & 32u) ? ((uint64_t)a_hi | ((uint64_t)a_lo << 32)) : ((uint64_t)a_lo | ((uint64_t)a_hi << 32));
    };
    bool crlf_mode = false;                                   // (wave-uniform, sticky) the file has "\r\n" line ends
    const uint32_t *bitmap = ss.bitmap, *rank = ss.rank;
    uint32_t hits = 0, lines_seen = 0, any_hi = 0;
    // Matched lines are published with atomicMax (the last duplicate of a position wins, whichever wave sees it).  An atomic issued
    // while a tile is parsed sits between the DMA requests of the next two tiles in the wave's memory queue, and the counted wait at
    // the top of the next tile cannot tell it from a request: the wave stood there until the atomic had been to memory and back —
    // 4-7 % of the launch, more or less by where site_line happened to lie (round 5: the "placement modes").  So the matches collect
    // in LDS and go out together right AFTER a top-of-tile wait: by the next one they have had a whole tile's time.
    uint32_t n_hit = 0;                                       // (wave-uniform) entries of ws.hit_site / ws.hit_off
    uint64_t hit_base = 0;                                    // (wave-uniform) what their offsets are counted from
    unsigned long long depth_acc = 0;
    unsigned long long t_a = 0, t_b = 0, t_c = 0, t_s = 0, t_i = 0, t_mark = kTime == 2 ? __builtin_readcyclecounter() : 0;
    constexpr bool kStamp = kTime == 1 || kTime == 2;
    const unsigned long long rt_start = kStamp ? __builtin_amdgcn_s_memrealtime() : 0;    // 100 MHz wall clock
    unsigned long long rt_prologue = 0, rt_first = 0;
#define WTICK(acc) do { if (kTime == 2) { unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - t_mark; t_mark = now_; } } while (0)
    if (threadIdx.x < 16) {
        const uint32_t nd = threadIdx.x > 10 ? 10 : threadIdx.x, first = 15u - nd;
        uint32_t m[3];
        for (int g = 0; g < 3; ++g) {
            uint32_t v = 0;
            for (uint32_t bb = 0; bb < 4; ++bb) { uint32_t idx = 4 + 4 * g + bb; if (idx >= first && idx <= 14) v |= 0xFFu << (8 * bb); }
            m[g] = v;
        }
        sh.digit_mask[threadIdx.x] = make_uint4(m[0], m[1], m[2], 0);
    }
    __syncthreads();                                          // the only barrier: digit_mask table visible

    const uint64_t gwave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    // my sample: the host dealt the waves of the launch to the samples in proportion to their sizes
    uint32_t si = 0;
    {
        uint32_t lo_i = 0, hi_i = a.n_samples;                // last sample whose first wave is <= gwave
        while (hi_i - lo_i > 1) { const uint32_t mid = (lo_i + hi_i) >> 1; if (a.samples[mid].wave0 <= gwave) lo_i = mid; else hi_i = mid; }
        si = __builtin_amdgcn_readfirstlane(lo_i);
    }
    const uint32_t s_wave0 = __builtin_amdgcn_readfirstlane(a.samples[si].wave0), s_waves = __builtin_amdgcn_readfirstlane(a.samples[si].n_waves);
    const ScanFile f = scan_file(a, si);
    // the tiles of the sample this launch covers: all of them, or the part of a file that has landed so far
    const uint64_t r_lo = __builtin_amdgcn_readfirstlane(a.samples[si].tile_lo);
    const uint64_t n_tiles = __builtin_amdgcn_readfirstlane(a.samples[si].tile_hi) - r_lo;
    auto interior = [&](uint64_t tt) { uint64_t x0 = tt * SCAN_TILE; return x0 >= f.lo + 16 && x0 + SCAN_TILE + SCAN_HALO <= f.hi; };
    auto flush_hits = [&]() {
        for (uint32_t i = lane; i < n_hit; i += 64)
            atomicMax((unsigned long long *)&f.site_line[ws.hit_site[i]], (unsigned long long)(hit_base + ws.hit_off[i]));
        n_hit = 0;
    };
    // request tile tt into slot `buf`; returns the number of DMA wave-instructions now in flight for it (0: staged synchronously)
    auto request = [&](uint64_t tt, int buf) -> uint32_t {
        if (interior(tt)) {
            // scalar base + per-lane 32-bit offset (lane * 16, the same for every tile) + immediate: no vector address
            // arithmetic per tile; the tile's base address lives in an SGPR pair
            const uint8_t *gs = f.base + tt * SCAN_TILE - 16;
            const uint32_t voff = lane * 16u;
            const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)&ws.slot[buf][0]);
#pragma unroll
            for (int r = 0; r < SCAN_DMA_PER_TILE; ++r) {
                // (the immediate offset — 12 bits — moves the global AND the LDS address; M0 carries the rest)
                const uint64_t ga = (uint64_t)(uintptr_t)gs + (r >= 4 ? 4096u : 0u);
                // (wave-uniform by construction; the readfirstlane pair makes that a fact for the "s" constraint)
                const uint64_t gr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ga >> 32)) << 32) |
                                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ga);      // (the builtin returns int)
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (r >= 4 ? 4096 : 0));
                if (r * 64 + lane < SCAN_WTILE_CHUNKS)                     // the last instruction has a partial exec mask
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(gr), "s"(m0v), "n"((r & 3) * 1024) : "memory");
            }
            return SCAN_DMA_PER_TILE;
        }
        // first / last tiles of the file: byte loads, bytes outside [lo,hi) read as '\n'
        const int64_t x0 = (int64_t)(tt * SCAN_TILE) - 16;
#pragma nounroll
        for (uint32_t e = lane; e < SCAN_WTILE_CHUNKS; e += 64) {
            uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int64_t idx = x0 + (int64_t)e * 16 + 4 * k + j;
                    uint32_t bb = (idx >= (int64_t)f.lo && idx < (int64_t)f.hi) ? (uint32_t)f.base[idx] : 10u;
                    d[k] |= bb << (8 * j);
                }
            ws.slot[buf][e] = make_uint4(d[0], d[1], d[2], d[3]);
        }
        return 0;
    };

    uint32_t win_base = 0xFFFFFF00u;                          // bitmap window [win_base, win_base + 64) dwords; starts empty
    uint32_t win_word = 0, win_rank = 0;
    uint16_t *lstart = ws.lstart;
    const uint32_t *hint_w = ws.hint_w;
    const uint32_t *hint_m = ws.hint_m;
    // the wave's contig hint (wave-uniform): length, probe parameters, and the first 16 name bytes + masks in registers
    uint32_t L = 0, h_max = 0, hint_bad = 1, hw[4] = {0, 0, 0, 0}, hm[4] = {0, 0, 0, 0};
    uint64_t h_off = 0;
    bool hint_long = false, hint_present = false;
    auto adopt = [&](Hint h) {                                // h came back from a call in VGPRs: make it uniform
        L = __builtin_amdgcn_readfirstlane(h.len);
        const uint32_t cid = __builtin_amdgcn_readfirstlane(h.cid);
        h_max = __builtin_amdgcn_readfirstlane(h.max_pos);
        h_off = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(h.bit_off >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)h.bit_off);
        hint_present = cid < SCAN_HINT_ABSENT;
        hint_bad = (cid != SCAN_HINT_NONE && L >= 1 && L <= 4 * SCAN_HINT_WORDS - 4) ? 0u : 1u;
        hint_long = L > 15;
        if (lane < SCAN_HINT_WORDS) {
            const uint32_t nb = L > 4u * lane ? L - 4u * lane : 0u;
            ws.hint_m[lane] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hw[k] = ws.hint_w[k];
            const uint32_t nb = L > 4u * k ? L - 4u * k : 0u;
            hm[k] = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
        }
    };
    // One-window parse: positions of a sorted pileup have the same number of digits g, and the same digits in front of the last four,
    // for long stretches, so with the name length L known the whole "name SEP digits SEP" prefix sits at a fixed place in the 24 bytes
    // that end with the second separator, and all of it but the last four digits is known in advance.  A line that differs — other top
    // digits (every 10 000 positions), another digit count, another contig — fails the masked compare (exactly: it passes iff the
    // bytes are the expected ones and the last four are digits); its round is done again in the general form, which sets g and the
    // top digits for the rounds to come.
    uint32_t g = 0, nw[6] = {0, 0, 0, 0, 0, 0}, nm[6] = {0, 0, 0, 0, 0, 0}, dmk4 = 0;
    uint32_t top = 0, pos_base = 0;                           // (wave-uniform) the positions' digits before the last four, and top * 10 000
    uint32_t streak = 0, cool = 0;                            // (wave-uniform) one-window rounds that failed in a row; general rounds still to run
    uint32_t mode = 0;                                        // (wave-uniform) bit 0: one-window rounds, bit 1: the name is checked in two pieces, bit 2: the
                                                              // byte before the line lies in the window (one register, read once per round)
    auto relayout = [&]() {
        // names that do not fit in front of the digits (L > 22 - g) are checked in two pieces: their tail in the window,
        // their first 16 bytes against the hint registers of the general parse (together: names up to 38 - g bytes)
        // (g = 10 and top digits from 429 496 on: top * 10 000 + 9 999 would not fit 32 bits — such positions are in no site set, and
        // the general parse knows)
        const bool fastc = hint_bad == 0 && g >= 1 && g <= 10 && L + g <= 38 && top <= 429495u;
        pos_base = top * 10000u;
        mode = (fastc ? 1u : 0u) | (L + g > 22 ? 2u : 0u) | (L + g <= 21 ? 4u : 0u);   // (scan_layout puts the '\n' before the line into the masks)
        if (!fastc) return;
        scan_layout(ws.lay, ws.hint_w, L, g, top, lane);
#pragma unroll
        for (int k = 0; k < 6; ++k) { nw[k] = ws.lay[k]; nm[k] = ws.lay[6 + k]; }
        dmk4 = ws.lay[12];
    };
    if (!kExact) adopt(load_hint(contig_tab(ss), 0, ws.hint_w, lane));

    // Every wave takes one contiguous run of its sample's tiles: a pileup is sorted, so the contig hint and the bitmap
    // window survive from one tile to the next.  Shares are static and equal; the launcher makes sure every CU holds
    // the same number of waves.  (Drawing work tickets from a device counter was measured and rejected: same-address
    // atomics cost ~12 ns each, device-wide, and stall the loads queued behind them.)
    // ... weighted by age: the SIMD issues oldest-first, so of the four waves it holds the oldest gets the most issue
    // slots; with equal shares it finishes ~35 % before the youngest, which then runs alone at half the SIMD's rate.
    // cum(x) = total weight of the launch's waves [0, x); a workgroup is 16 waves, wave w has age rank w / 4.
    // How much faster the older waves are depends on what a tile costs: where lines are long (30x and deeper: <= 47 lines per tile) the
    // measured best shares are a.share, where they are short (8x: 100 lines per tile and more) the flatter a.share_dense, in between
    // (15x: 73 lines) what lies between the two (tools/scan_sweep.py).  Every wave of a sample counts the terminators of the same
    // tile — the middle one of the sample's range in this launch — so all of them arrive at the same shares.
    uint32_t shr[4] = {a.share[0], a.share[1], a.share[2], a.share[3]};
    if (!kExact && n_tiles >= 3 && interior(r_lo + n_tiles / 2)) {
        const uint4 *gp = (const uint4 *)(f.base + (r_lo + n_tiles / 2) * SCAN_TILE) + 4 * lane;
        uint32_t n = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = gp[i];
            n += __popc(term_flags(v.x)) + __popc(term_flags(v.y)) + __popc(term_flags(v.z)) + __popc(term_flags(v.w));
        }
        for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o);
        const uint32_t lines = __builtin_amdgcn_readfirstlane(n);
        const uint32_t t256 = lines <= 47u ? 0u : lines >= 100u ? 256u : (lines - 47u) * 256u / 53u;
#pragma unroll
        for (int k = 0; k < 4; ++k) shr[k] = (a.share[k] * (256u - t256) + a.share_dense[k] * t256) >> 8;
    }
    auto cum = [&](uint64_t x) -> uint64_t {
        const uint32_t wpb = blockDim.x >> 6, r = (uint32_t)(x % wpb);
        uint64_t c = 0, blk = 0;
        for (uint32_t w = 0; w < wpb; ++w) {
            const uint32_t q = (w >> 2) & 3u, sw = q == 0 ? shr[0] : q == 1 ? shr[1] : q == 2 ? shr[2] : shr[3];
            blk += sw; c += w < r ? sw : 0;
        }
        return (x / wpb) * blk + c;
    };
    const uint64_t c_lo = cum(s_wave0), c_span = cum((uint64_t)s_wave0 + s_waves) - c_lo;
    const uint64_t t_first = r_lo + n_tiles * (cum(gwave) - c_lo) / c_span, t_end = r_lo + n_tiles * (cum(gwave + 1) - c_lo) / c_span;
    const uint64_t kNoTile = ~0ull;
    // slot b holds tile t_first + b + k * NBUF; bit b of dma_mask: the tile now in slot b was requested by DMA (an edge
    // tile is staged synchronously instead).  Wave-uniform scalars throughout.
    uint64_t t_req = t_first;
    uint32_t dma_mask = 0;
#pragma unroll
    for (int b = 0; b < SCAN_NBUF; ++b)
        if (t_req < t_end) dma_mask |= (__builtin_amdgcn_readfirstlane(request(t_req++, b)) ? 1u : 0u) << b;
    uint32_t cur = 0;
    bool first_tile = true;
    if (kStamp) rt_prologue = __builtin_amdgcn_s_memrealtime();
    for (uint64_t tt = t_first < t_end ? t_first : kNoTile; tt != kNoTile; tt = tt + 1 < t_end ? tt + 1 : kNoTile) {
        // the current tile's DMA has landed when only the requests of the later tiles are still outstanding
        const uint32_t later = (uint32_t)__builtin_popcount(dma_mask & ~(1u << cur));
        if (later == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SCAN_DMA_PER_TILE) : "memory");
        else if (later == 2 || SCAN_NBUF < 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * SCAN_DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * SCAN_DMA_PER_TILE) : "memory");
        __builtin_amdgcn_wave_barrier();
        // the matches so far go out here, a whole tile's time before the next counted wait (and long before their offsets outgrow 32 bits)
        if (n_hit >= SCAN_HIT_FLUSH || ((uint32_t)tt & 0xFFFFu) == 0u) flush_hits();    // (every 65 536 tiles at the latest: 256 MiB of offsets)
        WTICK(t_a);
        {
            const uint64_t t0 = tt * SCAN_TILE;
            const bool edge = ((dma_mask >> cur) & 1u) == 0;          // staged synchronously = not an interior tile
            const uint8_t *tile = (const uint8_t *)&ws.slot[cur][1];   // tile[-16 .. SCAN_TILE+SCAN_HALO)
            const uint4 *tile16 = &ws.slot[cur][1];
            uint32_t mismatch_at = 0xFFFFFFFFu;                      // line start of a lane whose name did not match the hint
            do {                                                // one pass; `break` leaves the tile early
                if (kTime == 3) { any_hi |= tile16[lane].x; break; }   // tuning: stream only, no parsing
                // ---- B: terminator flags of four 16-byte chunks per lane (chunk i*64+lane: conflict-free LDS reads) ----
                // bit 16*i + b of S: byte b of chunk i*64+lane is in 0x0A..0x0D; a line starts at the next byte.
                // Straight-line code.  That every flagged byte really is '\n' is checked on the list of line starts
                // (one byte per line) instead of on every byte here; if one is not ('\r', '\v', '\f'), the tile is
                // indexed again byte by byte with the universal-newline rules.
                uint64_t S;
                {
                    uint32_t bits[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint4 v = tile16[4 * lane + ((uint32_t)i ^ (k16 >> 4))];
                        any_hi |= v.x | v.y | v.z | v.w;
                        // the two SWAR adds on 8 bytes at a time (v_lshl_add_u64), the 3-input select per dword
                        const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                        const uint64_t la = lo + 0x7676767676767676ull, lb = lo + 0x7272727272727272ull;
                        const uint64_t ha = hi + 0x7676767676767676ull, hb = hi + 0x7272727272727272ull;
                        bits[i] = flags_to_bits16((uint32_t)la & ~(uint32_t)lb & 0x80808080u, (uint32_t)(la >> 32) & ~(uint32_t)(lb >> 32) & 0x80808080u,
                                                  (uint32_t)ha & ~(uint32_t)hb & 0x80808080u, (uint32_t)(ha >> 32) & ~(uint32_t)(hb >> 32) & 0x80808080u);
                    }
                    S = (uint64_t)(bits[0] | (bits[1] << 16)) | ((uint64_t)(bits[2] | (bits[3] << 16)) << 32);
                }
                WTICK(t_s);
                // The last byte of the sub-tile (lane 63, chunk 3, byte 15) starts a line in the NEXT sub-tile, which
                // sees it as its byte -1; byte 0 starts a line iff the byte before it ends a terminator.
                if (lane == 63) S &= ~(1ull << 15);                 // (lane 63 reads its last chunk first: k16 = 48)
                if (!kExact) {
                    // CR LF files: the '\r' of a pair flags the '\n' after it as a start.  Two flagged neighbours in one 16-byte
                    // chunk are looked at here (two byte reads per pair) and the '\r' loses its flag, so such files index and
                    // parse like LF files; a pair across two chunks keeps both flags, phase C sorts it out — and switches the
                    // wave to crlf_mode, in which every pair is looked at, the one that ends in the next lane too (a phantom
                    // start costs a whole round where a lane takes one line per round).
                    if (!crlf_mode) {
                        uint64_t pairs = S & (S >> 1) & 0x7FFF7FFF7FFF7FFFull;
                        if (__ballot(pairs != 0)) {
                            while (pairs) {
                                const uint32_t bpos = (uint32_t)__ffsll((long long)pairs) - 1;
                                pairs &= pairs - 1;
                                const uint32_t q = byte_of(bpos);
                                if (tile[q] == 13u && tile[q + 1] == 10u) S &= ~(1ull << bpos);
                            }
                        }
                    } else {                                        // every pair: the bits in address order, the next lane's first one
                        uint64_t U = regroup(S);
                        uint64_t pairs = U & ((U >> 1) | ((uint64_t)(__shfl_down((uint32_t)U, 1) & 1u) << 63));
                        while (pairs) {
                            const uint32_t bpos = (uint32_t)__ffsll((long long)pairs) - 1;
                            pairs &= pairs - 1;
                            const uint32_t q = (lane << 6) | bpos;
                            if (tile[q] == 13u && tile[q + 1] == 10u) U &= ~(1ull << bpos);
                        }
                        S = regroup(U);
                    }
                }
                const uint32_t pv0 = tile[-1], cv0 = tile[0];
                bool s0 = (lane == 0) && (pv0 == 10u || (pv0 == 13u && cv0 != 10u));
                uint32_t cnt = 0, n_lines = 0, base = 0;
                auto mask_edge = [&]() {                            // first / last tiles of a file: starts must lie inside it
#pragma nounroll
                    for (uint32_t b = 0; b < 64; ++b) {
                        const uint64_t st = t0 + (uint64_t)(byte_of(b) + 1u);
                        if (st < f.lo || st >= f.hi) S &= ~(1ull << b);
                    }
                    s0 = s0 && t0 >= f.lo && t0 < f.hi;
                };
                auto finish_index = [&]() {                         // (exact instantiation) per-lane counts, wave prefix sum
                    if (edge) mask_edge();
                    cnt = (uint32_t)__popcll(S) + (s0 ? 1u : 0u);
                    const uint32_t incl = wave_inclusive_sum(cnt);
                    n_lines = __builtin_amdgcn_readlane(incl, 63);
                    base = incl - cnt;
                };
                auto build_list = [&](uint32_t pass0) {
                    // list of the line starts [pass0, pass0 + CAP): two predicated slots, a loop only for lanes
                    // with three or more starts in their 64 bytes (lines shorter than ~21 bytes)
                    uint64_t s_bits = S;
                    uint32_t idx = base - pass0;                    // slots below 0 wrap to huge values and are skipped
                    if (s0 && idx < SCAN_LIST_CAP) lstart[idx] = 0;
                    idx += s0 ? 1u : 0u;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const uint32_t bpos = (uint32_t)__ffsll((long long)s_bits) - 1;         // 0xFFFFFFFF when empty
                        const bool have = s_bits != 0;
                        if (have && idx < SCAN_LIST_CAP) lstart[idx] = (uint16_t)(byte_of(bpos & 63u) + 1u);
                        idx += have ? 1u : 0u;
                        s_bits &= s_bits - 1;
                    }
                    if (__ballot(s_bits != 0)) {
                        while (s_bits) {
                            const uint32_t bpos = (uint32_t)__ffsll((long long)s_bits) - 1;
                            s_bits &= s_bits - 1;
                            if (idx < SCAN_LIST_CAP) lstart[idx] = (uint16_t)(byte_of(bpos) + 1u);
                            ++idx;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                };
                if (kExact) finish_index();
                // The fast index flags every byte in 0x0A..0x0D.  The fast parse checks each start itself (phase C: the byte
                // before it is '\n', else the lane decides from two bytes what it is looking at); the exact instantiation
                // checks the list here and, when a tile holds anything but "\n" and "\r\n", indexes it again byte by byte.
                bool redo = kExact && (n_lines > SCAN_LIST_CAP || __ballot((lane == 0) && (pv0 - 11u <= 2u)) != 0);
                if (kExact && !redo) {
                    build_list(0);
                    // A start is proper when it follows a '\n'.  In a CR LF file every '\r' flags a second start, the '\n'
                    // that follows it: such a phantom (it follows '\r' and IS '\n') is marked in the list (bit 15) and
                    // skipped, so CR LF files keep the fast index.  Anything else ('\r' alone, '\v', '\f') is odd.
                    bool odd = false;
                    uint32_t phantoms = 0;
                    for (uint32_t j = lane; j < n_lines; j += 64) {
                        const uint32_t st = lstart[j];
                        const uint32_t pv = tile[(int)st - 1], cv = tile[st];
                        const bool ph = pv == 13u && cv == 10u;
                        odd = odd || (pv != 10u && !ph);
                        if (ph) { lstart[j] = (uint16_t)(st | 0x8000u); ++phantoms; }
                        if (n_lines <= 64) break;
                    }
                    redo = __ballot(odd) != 0;
                    if (!redo && __ballot(phantoms != 0)) {
                        for (int o = 32; o; o >>= 1) phantoms += __shfl_xor(phantoms, o);
                        lines_seen -= (lane == 0) ? phantoms : 0;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                const bool listed = !redo;                          // the list of pass 0 is already in LDS
                if (redo) {
                    S = 0;                                          // rare: exact byte-wise index
#pragma nounroll
                    for (int i = 0; i < 4; ++i)
#pragma nounroll
                        for (int b = 0; b < 16; ++b) {
                            const int q = (int)byte_of((uint32_t)(16 * i + b));
                            const uint32_t cv = tile[q], nx = tile[q + 1];
                            if (cv == 10u || (cv == 13u && nx != 10u)) S |= 1ull << (16 * i + b);
                        }
                    if (lane == 63) S &= ~(1ull << 15);
                    s0 = (lane == 0) && (pv0 == 10u || (pv0 == 13u && cv0 != 10u));
                    finish_index();
                }
                if (kExact) lines_seen += (lane == 0) ? n_lines : 0;
                else {
                    if (edge) mask_edge();
                    lines_seen += (uint32_t)__popcll(S) + (s0 ? 1u : 0u);     // (every lane keeps its own count; added up at the end)
                    n_lines = 1;                                    // one trip through the pass loop below
                }
                WTICK(t_i);

                for (uint32_t pass0 = 0; pass0 < n_lines; pass0 += SCAN_LIST_CAP) {
                    if (kExact && (!listed || pass0 != 0)) build_list(pass0);
                    const uint32_t n_here = n_lines - pass0 < SCAN_LIST_CAP ? n_lines - pass0 : SCAN_LIST_CAP;
                    if (!kExact) {
                    // ---- C: one lane per line ------------------------------------------------------------------
                    {
                        // Straight-line code for "name SEP digits SEP" with as little scalar/exec traffic as possible:
                        // every test lands in one per-lane `bad` word; a line that does not fit (other contig, odd
                        // whitespace, > 10 digits, long name ...) is queued for the exact parser.
                        // No list of line starts: a lane takes the lines that start behind the terminators of ITS 64 bytes,
                        // one per round (30x: one round, now and then a second; 8x / 15x: two); the rounds end when no lane
                        // has a terminator left.  Lane 0 first takes the line that starts with the tile (s0).
                        uint64_t pend = S;
                        bool extra = s0;
                        uint64_t extra_mask = __ballot(extra);          // (a ballot of a plain compare is one v_cmp; of anything else the
                        // Lines of ANOTHER contig than the wave's hint are put aside (their terminator bits in `defer`); when the tile's rounds are
                        // through, that contig becomes the hint and they are taken in a second trip through the rounds, over the same slot.  (Up to
                        // round 5 they went to the exact parser's queue: 6 000 lines of every 200-contig sample.)
                        uint64_t defer = 0;
                        for (uint32_t second = 0;; second = 1u) {
                        for (;;) {                                      //  compiler first makes a 0 / 1 register: lane masks are kept in scalars)
                            bool active = extra || pend != 0;
                            uint64_t act_mask = __builtin_amdgcn_ballot_w64(pend != 0) | extra_mask;
                            extra_mask = 0;
                            if (!act_mask) break;
                            const uint32_t bpos = extra ? 0xFFFFFFFFu : (uint32_t)__ffsll((long long)pend) - 1u;
                            const uint32_t s = extra ? 0u : byte_of(bpos & 63u) + 1u;
                            if (!extra) pend &= pend - 1;
                            extra = false;
                            uint32_t bad, pos;
                            bool big;
                            uint32_t nd_seen = 0;                                        // digits of this line's position (0: unknown)
                            bool tabs = true;
                            const uint32_t mode_now = __builtin_amdgcn_readfirstlane(mode);
                            bool fast_round = (mode_now & 1u) != 0 && cool == 0;         // wave-uniform: a scalar branch
                            if (!fast_round && cool) --cool;
                            if (fast_round) {
                                uint32_t w[6];
                                lds_window24(tile, (int)s + (int)(L + g) - 22, w);       // ends with the second separator
                                // the '\n' before the line, the name, both TABs and all but the last four digits of the position in one
                                // masked compare per dword
                                uint32_t bad_name = ((w[0] ^ nw[0]) & nm[0]) | ((w[1] ^ nw[1]) & nm[1]) | ((w[2] ^ nw[2]) & nm[2]) |
                                                    ((w[3] ^ nw[3]) & nm[3]) | ((w[4] ^ nw[4]) & nm[4]) | ((w[5] ^ nw[5]) & nm[5]);
                                if (mode_now & 2u) {                                     // uniform: a long name's first 16 bytes
                                    uint32_t v0, v1, v2, v3;
                                    lds_window16(tile, (int)s, v0, v1, v2, v3);
                                    bad_name |= ((v0 ^ hw[0]) & hm[0]) | ((v1 ^ hw[1]) & hm[1]) | ((v2 ^ hw[2]) & hm[2]) | ((v3 ^ hw[3]) & hm[3]);
                                }
                                if (!(mode_now & 4u))                                    // uniform: the window starts with the name
                                    bad_name |= (uint32_t)tile[(int)s - 1] ^ 10u;
                                // the last four digits (window bytes 19..22) are the only ones that differ from line to line over long
                                // stretches of a sorted pileup: SWAR digit test, v_dot4_u32_u8 with weights 100, 10, 1 and one 24-bit multiply-add
                                const uint32_t xd = (__builtin_amdgcn_alignbyte(w[5], w[4], 3u) ^ 0x30303030u) & dmk4;
                                bad = bad_name | (((xd + 0x76767676u) | xd) & 0x80808080u);
                                pos = pos_base + __umul24(__builtin_amdgcn_udot4(xd, 0x00010A64u, 0u, false), 10u) + (xd >> 24);
                                big = false;
                                nd_seen = g;
                                // A round in which some line does not fit — other top digits (every 10 000 positions), another digit count,
                                // another contig, a start that is none, odd separators — is done again in the general form, which sorts
                                // out which it is and sets the layout for the rounds to come.
                                if (__builtin_amdgcn_ballot_w64(bad != 0) & act_mask) {
                                    fast_round = false;
                                    nd_seen = 0;
                                    if (++streak >= 4u) { streak = 0; cool = 64u; }      // (a file the layout keeps failing on: general rounds for a while)
                                } else streak = 0;
                            }
                            if (!fast_round) {
                            bad = hint_bad;                                              // uniform: no usable hint
                            {                                                            // is this a line start at all? (as above)
                                const uint32_t pv = tile[(int)s - 1];
                                if (__ballot(active && pv != 10u)) {
                                    const uint32_t cv = tile[s];
                                    const bool no_line = active && (pv == 13u ? cv == 10u : pv != 10u);
                                    const uint64_t nl = __ballot(no_line);
                                    active = active && !no_line;
                                    lines_seen -= (lane == 0) ? (uint32_t)__popcll(nl) : 0u;
                                    if (__ballot(no_line && pv == 13u)) crlf_mode = true;
                                }
                            }
                            // name: masked dword compare (masks are zero past the name)
                            uint32_t w0, w1, w2, w3;
                            lds_window16(tile, (int)s, w0, w1, w2, w3);
                            bad |= ((w0 ^ hw[0]) & hm[0]) | ((w1 ^ hw[1]) & hm[1]) | ((w2 ^ hw[2]) & hm[2]) | ((w3 ^ hw[3]) & hm[3]);
                            if (hint_long) {                                             // uniform: names of 16..44 bytes
                                uint32_t v0, v1, v2, v3;
                                lds_window16(tile, (int)s + 16, v0, v1, v2, v3);
                                bad |= ((v0 ^ hint_w[4]) & hint_m[4]) | ((v1 ^ hint_w[5]) & hint_m[5]) | ((v2 ^ hint_w[6]) & hint_m[6]) | ((v3 ^ hint_w[7]) & hint_m[7]);
                                lds_window16(tile, (int)s + 32, v0, v1, v2, v3);
                                bad |= ((v0 ^ hint_w[8]) & hint_m[8]) | ((v1 ^ hint_w[9]) & hint_m[9]) | ((v2 ^ hint_w[10]) & hint_m[10]) | ((v3 ^ hint_w[11]) & hint_m[11]);
                            }
                            const uint32_t c1 = tile[s + L];                             // the separator after the name
                            if (active && (bad | min(c1 ^ 9u, c1 ^ 32u)) != 0) mismatch_at = s;    // another contig?
                            uint4 q1;
                            lds_window16(tile, (int)(s + L + 1), q1.x, q1.y, q1.z, q1.w);       // digits + separator
                            // first byte <= 0x20 in the 16-byte window = number of digits (1..10 on the fast path)
                            const uint32_t ctl = flags_to_bits16(le20_flags(q1.x), le20_flags(q1.y), le20_flags(q1.z), le20_flags(q1.w));
                            const uint32_t nd_raw = (uint32_t)__ffs((int)ctl) - 1u;      // ctl == 0 -> 0xFFFFFFFF
                            bad |= (uint32_t)(nd_raw - 1u > 9u);
                            const uint32_t nd = nd_raw > 10u ? 10u : nd_raw;
                            uint4 q;
                            lds_window16(tile, (int)(s + L + 1 + nd) - 15, q.x, q.y, q.z, q.w);  // the digits end at byte 14 of this window
                            const uint4 dm = sh.digit_mask[nd];                          // keeps the nd digit bytes of q.y q.z q.w
                            const uint32_t c2 = q.w >> 24;
                            // separators of the fast path: TAB or space after the name; TAB, space or '\n' after the digits
                            bad |= min(c1 ^ 9u, c1 ^ 32u) | min(min(c2 ^ 9u, c2 ^ 32u), c2 ^ 10u);
                            const uint32_t x1 = (q.y ^ 0x30303030u) & dm.x, x2 = (q.z ^ 0x30303030u) & dm.y, x3 = (q.w ^ 0x30303030u) & dm.z;
                            bad |= (((x1 + 0x76767676u) | x1) | ((x2 + 0x76767676u) | x2) | ((x3 + 0x76767676u) | x3)) & 0x80808080u;
                            const uint64_t pos64 = ((uint64_t)(four_digits(x1) * 10000u + four_digits(x2))) * 1000ull + four_digits(x3 << 8);
                            big = pos64 > 0xFFFFFFFFull;
                            pos = (uint32_t)pos64;
                            nd_seen = bad == 0 ? nd : 0u;
                            tabs = ((c1 ^ 9u) | (c2 ^ 9u)) == 0;                          // the one-window parse wants TABs
                            }
                            if (kDepth) {
                                // 4th column (collect_metrics.py:325-340 sums int(tokens[3]) over the lines that have one):
                                // after the position's separator come one reference byte, a separator, 1..4 digits and a
                                // separator or the end of the line.  Anything else sends the line to the exact parser.
                                const uint32_t nd_here = nd_seen;                        // digits of the position (0: line is bad anyway)
                                uint32_t d0, d1, d2, d3;
                                lds_window16(tile, (int)(s + L + 1 + nd_here), d0, d1, d2, d3);   // [sep2][ref][sep][digits ...]
                                (void)d2; (void)d3;
                                const uint32_t c2d = d0 & 0xFFu, refb = (d0 >> 8) & 0xFFu, sepb = (d0 >> 16) & 0xFFu;
                                const uint32_t x = (d0 >> 24) | (d1 << 8);               // the four bytes after that separator
                                const uint32_t after4 = d1 >> 24;                        // ... and the fifth
                                const uint32_t ctl4 = __builtin_amdgcn_udot4(le20_flags(x), 0x08040201u, 0u, false) >> 7;
                                const uint32_t k = ctl4 ? (uint32_t)__ffs((int)ctl4) - 1u : 4u;  // digits before the first byte <= 0x20
                                const uint32_t xm = (x ^ 0x30303030u) & (k >= 4 ? 0xFFFFFFFFu : ((1u << (8 * k)) - 1u));
                                const bool digits_ok = ((((xm + 0x76767676u) | xm) & 0x80808080u) == 0) && k >= 1 && (k < 4 || after4 <= 0x20u);
                                const bool shape_ok = refb > 0x20u && (sepb == 9u || sepb == 32u) && digits_ok;
                                if (active && bad == 0 && nd_here != 0 && c2d != 10u) {  // the line goes on after the position (a line that is
                                                                                         // queued already has its depth added by the exact parser)
                                    if (shape_ok) depth_acc += four_digits(k >= 4 ? xm : xm << (8 * (4 - k)));
                                    else bad |= 1u;                                      // exact parser: any ref field, any depth
                                }
                            }
                            const uint64_t off1 = t0 + (uint64_t)s - f.lo + 1;
                            // rare: leave the line to k_scan_queue — with ONE atomic for all such lines of the round — unless it is a line of another
                            // contig on the first trip through the tile: that one is put aside for the trip after the contig change.
                            uint64_t qm = __builtin_amdgcn_ballot_w64(active && bad != 0);
                            if (qm) {
                                if (!second) {
                                    const bool aside = active && bad != 0 && mismatch_at == s && bpos != 0xFFFFFFFFu;
                                    if (aside) { defer |= 1ull << (bpos & 63u); bad = 0; active = false; }
                                    qm = __builtin_amdgcn_ballot_w64(active && bad != 0);
                                }
                                if (qm) {
                                    const uint32_t leader = (uint32_t)__ffsll((long long)qm) - 1u;
                                    uint32_t q0 = 0;
                                    if (lane == leader) q0 = atomicAdd(&a.ctl[0], (uint32_t)__popcll(qm));
                                    q0 = __builtin_amdgcn_readlane(q0, leader);
                                    if (active && bad != 0) {
                                        const uint32_t qi = q0 + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull));
                                        if (qi < a.q_cap) a.queue[qi] = f.sample | (off1 - 1); else a.ctl[1] = 1u;
                                    }
                                }
                            }
                            const bool probe = active && bad == 0 && hint_present && !big && pos <= h_max;
                            // Site bitmap probe without touching memory: the wave keeps a 64-dword window of the
                            // bitmap (and of its rank directory) in two VGPRs, one dword per lane; a pileup is
                            // position sorted, so a window (2048 positions) serves ~40 tiles before it is refilled
                            // with two coalesced 256-byte loads.  Lookup = ds_bpermute (cross-lane, no LDS memory).
                            const uint64_t bit = h_off + pos;
                            const uint32_t wi = (uint32_t)(bit >> 5);                   // < n_words, which fits 32 bits
                            uint32_t rel = wi - win_base;
                            uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel & 63u) << 2), (int)win_word);
                            uint32_t rk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel & 63u) << 2), (int)win_rank);
                            bool got = probe && rel < 64;
                            if (__ballot(probe && rel >= 64)) {                         // rare: some lane is outside the window
                                bool done = got || !probe;
                                for (;;) {
                                    const uint64_t miss = __ballot(!done);
                                    if (!miss) break;
                                    win_base = __builtin_amdgcn_readlane(wi, (uint32_t)__ffsll((long long)miss) - 1);
                                    const bool inb = (uint64_t)win_base + lane < ss.n_words;
                                    win_word = inb ? bitmap[(uint64_t)win_base + lane] : 0u;
                                    win_rank = inb ? rank[(uint64_t)win_base + lane] : 0u;
                                    // consume the two loads here: otherwise hipcc waits for them at the top of the loop with
                                    // vmcnt(0) on every round, which also waits for the next tile's LDS-DMA it knows nothing of
                                    asm volatile("" : "+v"(win_word), "+v"(win_rank));
                                    rel = wi - win_base;
                                    const uint32_t w_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel & 63u) << 2), (int)win_word);
                                    const uint32_t r_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel & 63u) << 2), (int)win_rank);
                                    if (!done && rel < 64) { word = w_; rk = r_; done = true; got = true; }
                                }
                            }
                            if (!got) word = 0;
                            const uint32_t shf = (uint32_t)(bit & 31);
                            {
                                const bool hit = ((word >> shf) & 1u) != 0;
                                hits += hit ? 1u : 0u;
                                const uint64_t hm = __ballot(hit);
                                if (hm) {                                                // (wave-uniform)
                                    if (n_hit + 64u > SCAN_HIT_CAP) flush_hits();       // (only where nearly every line is a site)
                                    if (n_hit == 0) hit_base = t0 - f.lo - 2 * SCAN_TILE;   // (modulo 2^64; every offset of this tile and the later ones lies above it)
                                    const uint32_t idx = n_hit + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                                    if (hit) {
                                        ws.hit_site[idx] = rk + __popc(word & ((1u << shf) - 1u));
                                        ws.hit_off[idx] = (uint32_t)(off1 - hit_base);
                                    }
                                    n_hit += (uint32_t)__popcll(hm);
                                }
                            }
                            if (!fast_round) {                                           // the general path calibrates the digit count
                                const uint64_t okm = __ballot(active && nd_seen != 0 && tabs && !big);   // lines separated otherwise never calibrate it
                                // the LAST such line of the round (lanes take their lines in file order): what comes next looks like it
                                const uint32_t src = 63u - (uint32_t)__clzll((long long)(okm | 1ull));
                                const uint32_t g_new = okm ? __builtin_amdgcn_readlane(nd_seen, src) : 0u;
                                const uint32_t top_new = okm && g_new > 4u ? __builtin_amdgcn_readlane(pos, src) / 10000u : 0u;
                                if (cool) { g = 0; mode = 0; }                           // (no new layout while the wave stays general: the first round after that sets it)
                                else if (g_new != g || top_new != top) { g = g_new; top = top_new; relayout(); }
                            }
                        }
                        // A line of another contig was seen: make that contig the wave's hint (rare: once per contig change), then take the
                        // lines that were put aside for it, once per tile.  (On that second trip a line that still does not match — a second
                        // contig change in the tile, a name the hint cannot hold, the hint's own name with something odd behind it — is queued.)
                        {
                            const uint64_t mm = __ballot(mismatch_at != 0xFFFFFFFFu);
                            if (mm) {
                                const uint32_t s1 = __builtin_amdgcn_readlane(mismatch_at, (uint32_t)__ffsll((long long)mm) - 1);
                                uint32_t len = 0;
                                while (len < 4 * SCAN_HINT_WORDS && __builtin_amdgcn_readfirstlane((uint32_t)tile[s1 + len]) > 0x20u) ++len;
                                // the same name as the hint's? then only the digit count (or a separator) differed: recalibrate, no look-up
                                bool same = hint_bad == 0 && len == L;
                                if (same) {
                                    bool diff = false;
                                    for (uint32_t i = lane; i < len; i += 64) diff = diff || tile[s1 + i] != ((ws.hint_w[i >> 2] >> (8 * (i & 3))) & 0xFFu);
                                    same = __ballot(diff) == 0;
                                }
                                if (same) { g = 0; mode = 0; }
                                else if (len >= 1 && len <= 4 * SCAN_HINT_WORDS - 4) {
                                    TileView tv{tile, f.base, t0, f.hi, (int64_t)(SCAN_TILE + SCAN_HALO)};
                                    const uint32_t cid = ss.n_contigs ? find_contig(contig_tab(ss), tv, (int64_t)s1, len) : 0xFFFFFFFFu;
                                    g = 0; mode = 0;                             // the next round calibrates against the new name
                                    if (cid != 0xFFFFFFFFu) adopt(load_hint(contig_tab(ss), cid, ws.hint_w, lane));
                                    else {                                       // not a contig of the site set: remember the name itself
                                        if (lane < SCAN_HINT_WORDS) {
                                            uint32_t w = 0;
                                            for (uint32_t j = 0; j < 4; ++j) { uint32_t i = lane * 4 + j; if (i < len) w |= (uint32_t)tile[s1 + i] << (8 * j); }
                                            ws.hint_w[lane] = w;
                                        }
                                        Hint h;
                                        h.len = len; h.cid = SCAN_HINT_ABSENT; h.max_pos = 0; h.bit_off = 0;
                                        adopt(h);
                                    }
                                }
                            }
                        }
                        if (second || !__ballot(defer != 0)) break;
                        mismatch_at = 0xFFFFFFFFu;
                        pend = defer;
                        defer = 0;
                        }
                    }
                    } else {
                        TileView tv{tile, f.base, t0, f.hi, (int64_t)(SCAN_TILE + SCAN_HALO)};
                        for (uint32_t j = lane; j < n_here; j += 64) {
                            if (lstart[j] & 0x8000u) continue;                          // the '\n' of a CR LF pair
                            const uint32_t s = lstart[j];
                            const uint64_t file_off = t0 + (uint64_t)s - f.lo;
                            SlowLine sl = parse_line_slow(tv, s, a.want_depth);
                            if (sl.err) { report_scan_error(f.status, file_off, sl.err); continue; }
                            depth_acc += sl.depth;
                            const uint32_t cid = ss.n_contigs ? find_contig(contig_tab(ss), tv, sl.f0, sl.f0len) : 0xFFFFFFFFu;
                            if (cid == 0xFFFFFFFFu || sl.pos > (uint64_t)ss.max_pos[cid]) continue;
                            const uint64_t bit = ss.bit_off[cid] + (uint32_t)sl.pos;
                            const uint32_t word = bitmap[bit >> 5];
                            const uint32_t shf = (uint32_t)(bit & 31);
                            if (!((word >> shf) & 1u)) continue;
                            const uint32_t site = rank[bit >> 5] + __popc(word & ((1u << shf) - 1u));
                            atomicMax((unsigned long long *)&f.site_line[site], (unsigned long long)(file_off + 1));
                            ++hits;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            } while (false);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of this slot has returned: it can be refilled
        __builtin_amdgcn_wave_barrier();
        WTICK(t_b);
        // the slot just parsed receives the tile NBUF ahead
        dma_mask &= ~(1u << cur);
        if (t_req < t_end) dma_mask |= (__builtin_amdgcn_readfirstlane(request(t_req++, (int)cur)) ? 1u : 0u) << cur;
        cur = cur + 1 == SCAN_NBUF ? 0 : cur + 1;
        WTICK(t_c);
        if (kStamp && first_tile) { rt_first = __builtin_amdgcn_s_memrealtime(); first_tile = false; }
    }
    if (kStamp && lane == 0 && a.dbg) {                        // per-wave record for tools/scan_waves.py
        unsigned long long *rec = a.dbg + 8 * gwave;
        rec[0] = rt_start; rec[1] = __builtin_amdgcn_s_memrealtime();
        rec[2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);   // XCC_ID, HW_ID
        rec[3] = t_a; rec[4] = t_b; rec[5] = t_c; rec[6] = kTime == 2 ? t_s : rt_prologue; rec[7] = kTime == 2 ? t_i : rt_first;
    }
#undef WTICK
    flush_hits();
    // a byte >= 0x80 anywhere in this wave's share of the file (checked once: the answers are void anyway)
    if (__ballot((any_hi & 0x80808080u) != 0) && lane == 0) report_scan_error(f.status, 0, SCAN_ERR_NON_ASCII);
    for (int o = 32; o; o >>= 1) { hits += __shfl_xor(hits, o); lines_seen += __shfl_xor(lines_seen, o); }
    if ((kExact && a.want_depth) || kDepth)
        for (int o = 32; o; o >>= 1) depth_acc += __shfl_xor(depth_acc, o);
    if (lane == 0) { a.totals[3 * gwave] = lines_seen; a.totals[3 * gwave + 1] = hits; a.totals[3 * gwave + 2] = depth_acc; }
}

// The exact parser over the queued lines (one lane per line, bytes read straight from global memory).
__global__ __launch_bounds__(256) void k_scan_queue(ScanArgs a, SiteSetDev ss) {
    const uint32_t n = a.ctl[0];
    if (n == 0 || a.ctl[1] != 0) return;                    // nothing queued, or overflow: the exact pass redoes the files
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t e = a.queue[i], off = e & ((1ull << 40) - 1);
        const ScanFile f = scan_file(a, (uint32_t)(e >> 40));
        TileView tv{nullptr, f.base + f.lo, 0, f.hi - f.lo, 0};   // lds_limit 0: every byte comes from global memory
        SlowLine sl = parse_line_slow(tv, (int64_t)off, a.want_depth);
        if (sl.err) { report_scan_error(f.status, off, sl.err); continue; }
        if (sl.depth) atomicAdd((unsigned long long *)&f.status[3], (unsigned long long)sl.depth);
        const uint32_t cid = ss.n_contigs ? find_contig(contig_tab(ss), tv, sl.f0, sl.f0len) : 0xFFFFFFFFu;
        if (cid == 0xFFFFFFFFu || sl.pos > (uint64_t)ss.max_pos[cid]) continue;
        const uint64_t bit = ss.bit_off[cid] + (uint32_t)sl.pos;
        const uint32_t word = ss.bitmap[bit >> 5];
        const uint32_t shf = (uint32_t)(bit & 31);
        if (!((word >> shf) & 1u)) continue;
        const uint32_t site = ss.rank[bit >> 5] + __popc(word & ((1u << shf) - 1u));
        atomicMax((unsigned long long *)&f.site_line[site], (unsigned long long)(off + 1));
        atomicAdd((unsigned long long *)&f.status[2], 1ull);
    }
}

// Before the scan, one launch: zero the line-offset rows of the batch, set every sample's status words, reset the queue
// control words and whatever small counters the caller wants cleared (the leftover counts of the call kernels).
__global__ __launch_bounds__(256) void k_scan_prepare(const SampleDev *samples, uint32_t n_samples, uint32_t *ctl, uint64_t *site_line,
                                                      uint64_t n_rows_words, uint32_t *zero32, uint32_t n_zero32) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows_words; i += (uint64_t)gridDim.x * blockDim.x) site_line[i] = 0;
    if (threadIdx.x == 0) {
        for (uint32_t b = blockIdx.x; b < n_samples; b += gridDim.x) {
            uint64_t *status = samples[b].status;
            status[0] = ~0ull; status[1] = status[2] = status[3] = 0;
        }
        if (blockIdx.x == 0) {
            ctl[0] = ctl[1] = 0;
            for (uint32_t k = 0; k < n_zero32; ++k) zero32[k] = 0;
        }
    }
}

// After a scan launch, one workgroup per sample: add up the per-wave totals.  exact == 0: the fast pass's, ADDED to the
// status words (a file may be scanned in several launches); exact == 1: the exact pass's — only when the queue
// overflowed, and they then REPLACE what the fast passes and the queue kernel counted.
__global__ __launch_bounds__(256) void k_scan_finish(const SampleDev *samples, const uint32_t *ctl, const uint64_t *totals, int exact) {
    __shared__ unsigned long long part[3][4];
    const SampleDev sd = samples[blockIdx.x];
    uint64_t *status = sd.status;
    const bool overflow = ctl[1] != 0;
    if (exact && !overflow) return;
    unsigned long long v[3] = {0, 0, 0};
    for (uint32_t w = threadIdx.x; w < sd.n_waves; w += blockDim.x)
        for (int k = 0; k < 3; ++k) v[k] += totals[3 * (size_t)(sd.wave0 + w) + k];
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned long long sum = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        // [1] lines: only the wave kernels count them; [2] matches, [3] depth: the queue kernel adds its own unless it is skipped
        if (exact) status[1 + threadIdx.x] = sum; else status[1 + threadIdx.x] += sum;
    }
}

// ------------------------------------------------------------------------------------------------
//            every line of a pileup (call_consensus --vcfAllPos, pileup.py:418-421)
// ------------------------------------------------------------------------------------------------
// With chrom_position_set None the reference builds a Record from EVERY line.  Three small kernels give the call
// kernel its work list: count the lines of every 16 KiB block, (prim scan of the counts,) emit 1 + the offset of every
// line in file order, and look every line's (chrom, pos) up in the site set for its flags (a line of an excluded
// position fails the Region filter, call_consensus.py:165-168).  A line ends with '\n', or with a '\r' that is not
// followed by '\n' (universal newlines, as the scan); a start at the end of the file is no line.
#define LINES_BLOCK 16384
#define LINES_THREADS 256
#define LINES_PER_THREAD (LINES_BLOCK / LINES_THREADS)

// 0x80 in every byte of w that equals the byte replicated in c4 (exact for any byte values: no carries between bytes)
__device__ __forceinline__ uint32_t eq_flags(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}

// Each thread owns 64 consecutive bytes (four 16-byte loads from the 16-byte aligned address at or below the buffer);
// blocks are laid out in those aligned coordinates.
template <bool kEmit>
__global__ __launch_bounds__(LINES_THREADS) void k_lines_index(const uint8_t *buf, uint64_t nbytes, uint32_t *block_counts, uint64_t *line_off, uint64_t capacity) {
    __shared__ uint32_t lds[17];
    const uint32_t shift = (uint32_t)((uintptr_t)buf & 15u);
    const uint4 *abase = (const uint4 *)(buf - shift);
    const uint64_t a0 = (uint64_t)blockIdx.x * LINES_BLOCK + (uint64_t)threadIdx.x * LINES_PER_THREAD;    // aligned coordinate of my byte 0
    const int64_t p0 = (int64_t)a0 - (int64_t)shift;                                                      // ... and its file offset
    // bit k of N / C: byte p0 + k is '\n' / '\r'
    uint64_t N = 0, C = 0;
    if (p0 < (int64_t)nbytes && p0 + LINES_PER_THREAD > 0) {
#pragma unroll
        for (int q = 0; q < LINES_PER_THREAD / 16; ++q) {
            const int64_t pq = p0 + 16 * q;
            if (pq >= (int64_t)nbytes || pq + 16 <= 0) continue;
            const uint4 v = abase[a0 / 16 + q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t nb = 0, cb = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                nb |= (__builtin_amdgcn_udot4(eq_flags(w[d], 0x0A0A0A0Au), 0x08040201u, 0u, false) >> 7) << (4 * d);
                cb |= (__builtin_amdgcn_udot4(eq_flags(w[d], 0x0D0D0D0Du), 0x08040201u, 0u, false) >> 7) << (4 * d);
            }
            N |= (uint64_t)nb << (16 * q);
            C |= (uint64_t)cb << (16 * q);
        }
        // bytes before the file / past its end are neither
        uint64_t valid = ~0ull;
        if (p0 < 0) valid &= ~0ull << (uint32_t)(-p0);
        if (p0 + LINES_PER_THREAD > (int64_t)nbytes) valid &= ~0ull >> (uint32_t)(p0 + LINES_PER_THREAD - (int64_t)nbytes);
        N &= valid;
        C &= valid;
    }
    // bit k of `starts`: a line starts at byte p0 + k — the byte before it is '\n', or a '\r' that it does not follow with '\n';
    // byte 0 of the file starts a line; a start at the end of the file is no line
    uint64_t starts = 0;
    if (p0 < (int64_t)nbytes && p0 + LINES_PER_THREAD > 0) {
        uint64_t pn = 0, pc = 0;
        if (p0 > 0) { const uint32_t b = buf[p0 - 1]; pn = b == 10u; pc = b == 13u; }
        starts = ((N << 1) | pn) | (((C << 1) | pc) & ~N);
        if (p0 <= 0) starts |= 1ull << (uint32_t)(-p0);
        if (p0 < 0) starts &= ~0ull << (uint32_t)(-p0);
        if (p0 + LINES_PER_THREAD > (int64_t)nbytes) starts &= ~0ull >> (uint32_t)(p0 + LINES_PER_THREAD - (int64_t)nbytes);
    }
    uint32_t total;
    const uint32_t mine = (uint32_t)__popcll(starts);
    const uint32_t ex = block_exclusive_sum(mine, lds, total);
    if (!kEmit) {
        if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
        return;
    }
    uint64_t idx = (uint64_t)block_counts[blockIdx.x] + ex;   // block_counts: exclusive scan of the counts (< 2^32 lines)
    while (starts) {
        const uint32_t k = (uint32_t)__ffsll((long long)starts) - 1;
        starts &= starts - 1;
        if (idx < capacity) line_off[idx] = (uint64_t)(p0 + (int64_t)k) + 1;
        ++idx;
    }
}

// flags[i] = site flags of line i's (chrom, pos) when it is in the site set, else 0; malformed chrom / pos columns are
// reported in status[0] as the scan does
__global__ __launch_bounds__(256) void k_lines_flags(const uint8_t *buf, uint64_t nbytes, const uint64_t *line_off, uint64_t n_lines,
                                                     SiteSetDev ss, uint8_t *flags, uint64_t *status) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t off = line_off[i] - 1;
        TileView tv{nullptr, buf, 0, nbytes, 0};             // lds_limit 0: every byte comes from global memory
        uint8_t f = 0;
        SlowLine sl = parse_line_slow(tv, (int64_t)off, 0);
        if (sl.err) report_scan_error(status, off, sl.err);
        else {
            const uint32_t cid = ss.n_contigs ? find_contig(contig_tab(ss), tv, sl.f0, sl.f0len) : 0xFFFFFFFFu;
            if (cid != 0xFFFFFFFFu && sl.pos <= (uint64_t)ss.max_pos[cid]) {
                const uint64_t bit = ss.bit_off[cid] + (uint32_t)sl.pos;
                const uint32_t word = ss.bitmap[bit >> 5], shf = (uint32_t)(bit & 31);
                if ((word >> shf) & 1u) f = ss.flags[ss.rank[bit >> 5] + __popc(word & ((1u << shf) - 1u))];
            }
        }
        flags[i] = f;
    }
}

static inline uint64_t lines_blocks(const uint8_t *d_buf, uint64_t nbytes) {     // blocks in the aligned coordinates of k_lines_index
    return nbytes ? (((uintptr_t)d_buf & 15u) + nbytes + LINES_BLOCK - 1) / LINES_BLOCK : 0;
}
size_t snpgpu_lines_workspace_words(uint64_t nbytes) { return prim_scan_workspace_words((nbytes + LINES_BLOCK - 1) / LINES_BLOCK + 2) + (nbytes + LINES_BLOCK - 1) / LINES_BLOCK + 2; }

// ws: snpgpu_lines_workspace_words(nbytes) words.  Leaves the number of lines in *d_total (a pointer into ws).
int snpgpu_enqueue_lines_count(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, uint32_t *ws, uint32_t **d_total) {
    const uint64_t nb = lines_blocks(d_buf, nbytes);
    if (nb > 0x7FFFFFFFull) return snpgpu_set_error(ctx, SNPGPU_E_UNSUPPORTED, "pileup too large for the line index");
    uint32_t *counts = ws, *scan_ws = ws + nb + 1;
    if (nb) k_lines_index<false><<<(unsigned)nb, LINES_THREADS, 0, ctx->stream>>>(d_buf, nbytes, counts, nullptr, 0);
    prim_exclusive_scan_u32(ctx->stream, counts, counts, nb, scan_ws, d_total);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// after snpgpu_enqueue_lines_count with the same ws: line offsets (+1) in file order and the flags of every line
int snpgpu_enqueue_lines_emit(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const uint8_t *d_buf, uint64_t nbytes, uint32_t *ws,
                              uint64_t *d_line_off, uint8_t *d_flags, uint64_t n_lines, uint64_t *d_status) {
    const uint64_t nb = lines_blocks(d_buf, nbytes);
    if (nb) k_lines_index<true><<<(unsigned)nb, LINES_THREADS, 0, ctx->stream>>>(d_buf, nbytes, ws, d_line_off, n_lines);
    if (n_lines) {
        const uint64_t blocks = (n_lines + 255) / 256, cap = (uint64_t)ctx->n_cu * 8;
        k_lines_flags<<<(unsigned)(blocks < cap ? blocks : cap), 256, 0, ctx->stream>>>(d_buf, nbytes, d_line_off, n_lines, ss->dev, d_flags, d_status);
    }
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// the offsets alone
int snpgpu_enqueue_lines_offsets(snpgpu_ctx *ctx, const uint8_t *d_buf, uint64_t nbytes, uint32_t *ws, uint64_t *d_line_off, uint64_t n_lines) {
    const uint64_t nb = lines_blocks(d_buf, nbytes);
    if (nb) k_lines_index<true><<<(unsigned)nb, LINES_THREADS, 0, ctx->stream>>>(d_buf, nbytes, ws, d_line_off, n_lines);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

size_t snpgpu_scan_totals_bytes(const snpgpu_ctx *ctx) { return 3 * 8 * (size_t)ctx->n_cu * 16 * 16 + 256; }   // (room for 16 launches' worth of waves: the grid may be a multiple of what is resident)
size_t snpgpu_scan_workspace_bytes(const snpgpu_ctx *ctx, uint32_t n_samples) {
    return ((size_t)(n_samples + 1) * sizeof(SampleDev) + 255) / 256 * 256 + snpgpu_scan_totals_bytes(ctx);
}

namespace {
struct ScanConfig {
    // oversub: the grid holds twice the workgroups that are resident at a time, so a CU that finishes its first one early takes
    // another (tools/scan_oversub.sh: 66.3 -> 67.9 % of HBM peak in a 96-sample launch at 30x; 4 and 8 give the same)
    int blocks_per_cu = 1, mode = 0, waves = SCAN_NBUF == 2 ? 16 : 12, oversub = 0;   // oversub 0: chosen by the launch's length
    int oversub_min = 1500;                                 // (tuning) tiles per resident wave from which a forced oversub applies
    int share[4] = {329, 282, 223, 169};                    // measured: 1 / (finish time with equal shares), oldest first (30x)
    int share_dense[4] = {320, 280, 225, 175};              // ... at 8x (tools/scan_sweep.py; {290, 266, 238, 206} while the atomics of the matches still stood in the waves' way: round 5)
    bool ready = false;
};
ScanConfig &scan_config() {
    static ScanConfig c;
#ifdef SNPGPU_TUNING
    if (getenv("SNPGPU_SCAN_RELOAD")) c = ScanConfig();     // tools/scan_sweep.py: many settings in one process
#endif
    if (!c.ready) {
#ifdef SNPGPU_TUNING                                        // development builds only (tools/): never in the product library
        const char *b = getenv("SNPGPU_SCAN_BLOCKS_PER_CU"), *m = getenv("SNPGPU_SCAN_MODE"), *w = getenv("SNPGPU_SCAN_WAVES");
        if (const char *sh = getenv("SNPGPU_SCAN_SHARE")) {
            int v[4];
            if (sscanf(sh, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0)
                for (int g = 0; g < 4; ++g) c.share[g] = c.share_dense[g] = v[g];
        }
        if (const char *sh = getenv("SNPGPU_SCAN_SHARE_DENSE")) {
            int v[4];
            if (sscanf(sh, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0)
                for (int g = 0; g < 4; ++g) c.share_dense[g] = v[g];
        }
        c.blocks_per_cu = b && atoi(b) > 0 && atoi(b) <= 16 ? atoi(b) : 1;
        if (const char *o = getenv("SNPGPU_SCAN_OVERSUB")) if (atoi(o) >= 0 && atoi(o) <= 16) c.oversub = atoi(o);
        if (const char *o = getenv("SNPGPU_SCAN_OVERSUB_MIN")) if (atoi(o) >= 0) c.oversub_min = atoi(o);
        c.mode = m ? atoi(m) : 0;
        if (w && atoi(w) >= 1 && atoi(w) <= 16) c.waves = atoi(w);
#endif
        c.ready = true;
    }
    return c;
}
// The scan kernels ask for more than 64 KiB of dynamic LDS; the permission is a per-device function attribute, so every
// context sets it for its own device (a process may drive several GPUs: call_consensus_batch).
void scan_allow_lds(snpgpu_ctx *ctx) {
    if (ctx->scan_lds_attr) return;
#ifdef SNPGPU_TUNING
    for (auto f : {(const void *)k_scan_wave<false, 1>, (const void *)k_scan_wave<false, 2>, (const void *)k_scan_wave<false, 3>})
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    for (auto f : {(const void *)k_scan_wave<false, 0>, (const void *)k_scan_wave<false, 4>, (const void *)k_scan_wave<true, 0>})
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    ctx->scan_lds_attr = true;
}
// Exactly blocks_per_cu workgroups fit on a CU (the LDS request is padded to make sure) and a full grid is
// n_cu * blocks_per_cu, so every CU runs the same number of waves and equal shares finish together.
size_t scan_lds_bytes(const ScanConfig &c) {
    const size_t lds_need = sizeof(ScanShared) - sizeof(WaveSlots) + (size_t)c.waves * sizeof(WaveSlots);
    const size_t lds_pad = (size_t)(160 * 1024) / (c.blocks_per_cu + 1) + 1024;
    return lds_need > lds_pad ? lds_need : lds_pad;
}
ScanArgs scan_args(const snpgpu_siteset *ss, const ScanConfig &c, const SampleDev *d_table, uint32_t n, uint64_t *d_totals,
                   uint64_t *d_site_line, int want_depth) {
    ScanArgs sa;
    sa.samples = d_table;
    sa.n_samples = n;
    sa.n_sites = ss->n_sites;
    sa.site_line = d_site_line;
    sa.want_depth = want_depth;
    for (int g = 0; g < 4; ++g) {
        const bool weighted = (c.waves == 16 || c.waves == 12) && c.blocks_per_cu == 1;
        sa.share[g] = weighted ? (uint32_t)c.share[g] : 1u;
        sa.share_dense[g] = weighted ? (uint32_t)c.share_dense[g] : 1u;
    }
    sa.queue = ss->slow_queue;
    sa.q_cap = SNPGPU_SLOW_QUEUE_CAP - 65536;               // the tail holds the tuning modes' per-wave records
    sa.ctl = ss->slow_ctl;
    sa.totals = d_totals;
    sa.dbg = nullptr;
    return sa;
}
}  // namespace

// Deals the waves of one launch to the samples of a host table in proportion to the tiles each has in its range (at
// least one wave each, at least min_tiles_per_wave tiles per wave where the range allows it).  Returns the wave count.
uint32_t snpgpu_scan_deal(const snpgpu_ctx *ctx, SampleDev *h, uint32_t n, uint32_t min_tiles_per_wave) {
    const ScanConfig &c = scan_config();
    if (!min_tiles_per_wave) min_tiles_per_wave = 1;
    std::vector<uint64_t> tiles(n);
    uint64_t total_tiles = 0;
    for (uint32_t i = 0; i < n; ++i) {
        tiles[i] = h[i].tile_hi - h[i].tile_lo;
        total_tiles += tiles[i];
    }
    // the grid is a multiple of what is resident only for launches long enough that a wave's start-up does not show (a second
    // workgroup per CU costs ~1.5 % at 800 tiles per wave, gains 2 % at 2 500: tools/scan_oversub.sh)
    // (round 4, without the line list: eight times the resident waves 72.5 % of HBM peak at the headline, four 72.2, two 71.2, one 69.5;
    // 125 samples at 8x — 1 536 tiles per resident wave — four 65.3, two 64.7, one 63.6; 48 samples at 30x — 1 237 — four 69.4, one 68.0;
    // 16 at 100x — 1 117 — four 74.0, one 72.9; from 800 down one workgroup per CU is as good or better: tools/scan_sweep.py)
    const uint64_t resident = (uint64_t)ctx->n_cu * c.blocks_per_cu * c.waves, per_wave = total_tiles / resident;
    const uint64_t max_waves = resident * (c.oversub ? (per_wave >= (uint64_t)c.oversub_min ? (uint64_t)c.oversub : 1) : per_wave >= 2500 ? 8 : per_wave >= 1000 ? 4 : 1);
#ifdef SNPGPU_TUNING
    const uint64_t max_waves_ = (c.mode == 8 || c.mode == 9) ? resident : max_waves;   // the per-wave records of these modes have room for the resident waves only
#define max_waves max_waves_
#endif
    uint64_t budget = total_tiles / min_tiles_per_wave, used = 0;
    if (budget > max_waves) budget = max_waves;
    if (budget < n) budget = n;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t w = total_tiles ? budget * tiles[i] / total_tiles : 0;
        h[i].n_waves = (uint32_t)(w ? w : 1);
        used += h[i].n_waves;
    }
    while (used < budget) {                                 // leftovers go where a wave's share is largest
        uint32_t best = 0;
        double worst = -1;
        for (uint32_t i = 0; i < n; ++i) { double r = (double)tiles[i] / h[i].n_waves; if (r > worst) { worst = r; best = i; } }
        if (worst <= 1.0) break;
        ++h[best].n_waves;
        ++used;
    }
    while (used > max_waves) {                              // only when many samples were rounded up to one wave
        uint32_t best = 0;
        for (uint32_t i = 0; i < n; ++i) if (h[i].n_waves > h[best].n_waves) best = i;
        --h[best].n_waves;
        --used;
    }
    uint32_t w0 = 0;
    for (uint32_t i = 0; i < n; ++i) { h[i].wave0 = w0; w0 += h[i].n_waves; }
#ifdef SNPGPU_TUNING
#undef max_waves
#endif
    return w0;
}

int snpgpu_scan_begin(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n, uint64_t *d_site_line,
                      uint32_t *d_zero32, uint32_t n_zero32) {
    const uint64_t row_words = (uint64_t)n * ss->n_sites;
    const uint64_t want_blocks = (row_words + 2047) / 2048 + 1, cap_blocks = (uint64_t)ctx->n_cu * 4;
    k_scan_prepare<<<(unsigned)(want_blocks < cap_blocks ? want_blocks : cap_blocks), 256, 0, ctx->stream>>>(
        d_table, n, ss->slow_ctl, d_site_line, row_words, d_zero32, n_zero32);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_scan_range(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n, uint32_t n_waves,
                      uint64_t *d_totals, uint64_t *d_site_line, int want_depth) {
    const ScanConfig &c = scan_config();
    scan_allow_lds(ctx);
    hipStream_t st = ctx->stream;
    if (n > (uint64_t)ctx->n_cu * c.blocks_per_cu * c.waves) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "too many samples for one scan launch");
    ScanArgs sa = scan_args(ss, c, d_table, n, d_totals, d_site_line, want_depth);
    const size_t lds = scan_lds_bytes(c);
    const unsigned grid = (unsigned)((n_waves + c.waves - 1) / c.waves), threads = (unsigned)c.waves * 64;
    hipEvent_t ta = snpgpu_time_begin(ctx);
    if (want_depth) {
        k_scan_wave<false, 4><<<grid, threads, lds, st>>>(sa, ss->dev);  // the same parse + the 4th column
#ifdef SNPGPU_TUNING
    } else if (c.mode == 7) {                               // tuning: LDS-DMA streaming rate without any parsing
        k_scan_wave<false, 3><<<grid, threads, lds, st>>>(sa, ss->dev);
    } else if (c.mode == 8 || c.mode == 9) {                // tuning: per-wave time stamps (9) + phase cycle counts (8)
        sa.dbg = (unsigned long long *)(ss->slow_queue + SNPGPU_SLOW_QUEUE_CAP - 65536);
        if (c.mode == 8) k_scan_wave<false, 2><<<grid, threads, lds, st>>>(sa, ss->dev);
        else k_scan_wave<false, 1><<<grid, threads, lds, st>>>(sa, ss->dev);
        if (const char *path = getenv("SNPGPU_SCAN_DUMP")) {
            const size_t nrec = (size_t)n_waves * 8;
            unsigned long long *recs = (unsigned long long *)malloc(nrec * 8);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(recs, sa.dbg, nrec * 8, hipMemcpyDeviceToHost);
            if (FILE *fp = fopen(path, "wb")) { fwrite(recs, 8, nrec, fp); fclose(fp); }
            free(recs);
        }
        sa.dbg = nullptr;
#endif
    } else {
        k_scan_wave<false, 0><<<grid, threads, lds, st>>>(sa, ss->dev);
    }
    snpgpu_time_end(ctx, SNPGPU_K_SCAN, ta);
    k_scan_finish<<<n, 256, 0, st>>>(d_table, ss->slow_ctl, d_totals, 0);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_scan_end(snpgpu_ctx *ctx, const snpgpu_siteset *ss, const SampleDev *d_table, uint32_t n, uint32_t n_waves,
                    uint64_t *d_totals, uint64_t *d_site_line, int want_depth) {
    const ScanConfig &c = scan_config();
    scan_allow_lds(ctx);
    hipStream_t st = ctx->stream;
    ScanArgs sa = scan_args(ss, c, d_table, n, d_totals, d_site_line, want_depth);
    const unsigned grid = (unsigned)((n_waves + c.waves - 1) / c.waves), threads = (unsigned)c.waves * 64;
    k_scan_queue<<<ctx->n_cu, 256, 0, st>>>(sa, ss->dev);
    k_scan_wave<true, 0><<<grid, threads, scan_lds_bytes(c), st>>>(sa, ss->dev);   // returns at once unless the queue overflowed
    k_scan_finish<<<n, 256, 0, st>>>(d_table, ss->slow_ctl, d_totals, 1);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

// Scans a batch of resident pileups with one launch.  h_samples[i].buf/nbytes/status are filled by the caller; tile
// ranges and wave shares are set here.  d_site_line is [n][n_sites] (zeroed here, together with the n_zero32 words at
// d_zero32).  `workspace` holds snpgpu_scan_workspace_bytes(); the device table stays at its start.
int snpgpu_enqueue_scan(snpgpu_ctx *ctx, const snpgpu_siteset *ss, std::vector<SampleDev> &h_samples, void *workspace,
                        uint64_t *d_site_line, int want_depth, uint32_t *d_zero32, uint32_t n_zero32) {
    const uint32_t n = (uint32_t)h_samples.size();
    if (!n) return SNPGPU_OK;
    for (uint32_t i = 0; i < n; ++i) {
        h_samples[i].tile_lo = 0;
        h_samples[i].tile_hi = (uint32_t)snpgpu_scan_tiles(h_samples[i].buf, h_samples[i].nbytes);
    }
    const uint32_t n_waves = snpgpu_scan_deal(ctx, h_samples.data(), n, 1);
    SampleDev sentinel{};
    sentinel.wave0 = n_waves;
    h_samples.push_back(sentinel);
    SampleDev *d_samples = (SampleDev *)workspace;
    HIP_TRY(ctx, hipMemcpyAsync(d_samples, h_samples.data(), (size_t)(n + 1) * sizeof(SampleDev), hipMemcpyHostToDevice, ctx->stream));
    h_samples.pop_back();
    uint64_t *d_totals = (uint64_t *)((char *)workspace + ((size_t)(n + 1) * sizeof(SampleDev) + 255) / 256 * 256);
    int rc = snpgpu_scan_begin(ctx, ss, d_samples, n, d_site_line, d_zero32, n_zero32);
    if (rc == SNPGPU_OK) rc = snpgpu_scan_range(ctx, ss, d_samples, n, n_waves, d_totals, d_site_line, want_depth);
    if (rc == SNPGPU_OK) rc = snpgpu_scan_end(ctx, ss, d_samples, n, n_waves, d_totals, d_site_line, want_depth);
    return rc;
}
