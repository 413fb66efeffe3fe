// K4-pack + K5: samples x sites matrix packed 4 bits/site, and the all-pairs SNP distance.
//
// Replaces utils.calculate_sequence_distance (snppipeline/utils.py:1135-1165) called for every pair by
// distance.calculate_snp_distances (snppipeline/distance.py:93-98):  a site counts for a pair iff, after
// upper-casing, both bytes are in {A,C,G,T} and they differ.
//
// Packed layout (HBM): packed[row][word] = uint4 { valid, code_hi, code_lo, lower } for 32 consecutive sites,
// bit i = site 32*word + i;  A=0 C=1 G=2 T=3.  Per pair and word:
//      mismatches += popcount( ((xh ^ yh) | (xl ^ yl)) & xv & yv )
// which gfx950 does in 4 VALU ops per 32 site-compares (v_xor, 2 x v_bitop3, v_bcnt).  A row is padded with zero words
// to a multiple of DIST_KW words.  This is integer VALU + LDS work (no MFMA): a 128x128 block of pairs per workgroup,
// 8x8 pairs per lane; slabs of DIST_KW words of both operands stream into a double-buffered LDS area with LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no LDS store instructions) one slab ahead of the arithmetic, one
// barrier per slab.  Slot s of row r holds word s ^ ((r >> 2) & 3) of the slab, which makes the 16 rows a wave reads at
// once fall into 16 different bank groups.
#include "internal.h"

#define DIST_TILE 128
#define DIST_KW 4                // words (of 32 sites) per slab
#define DIST_THREADS 256

static inline uint32_t padded_words(uint32_t n_sites) { return ((n_sites + 31) / 32 + DIST_KW - 1) / DIST_KW * DIST_KW; }
extern "C" size_t snpgpu_packed_row_bytes(uint32_t n_sites) { return (size_t)padded_words(n_sites) * 16; }

// One wavefront packs 64 consecutive sites of one row: coalesced byte loads, four ballots.
__global__ __launch_bounds__(256) void k_pack_matrix(const uint8_t *sym, uint32_t n_rows, uint32_t n_sites, size_t stride,
                                                     uint4 *packed, uint32_t words) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t groups_per_row = (words + 1) / 2;
    const uint64_t total = (uint64_t)n_rows * groups_per_row;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < total; g += (uint64_t)gridDim.x * 4) {
        uint32_t row = (uint32_t)(g / groups_per_row);
        uint32_t grp = (uint32_t)(g % groups_per_row);
        uint32_t site = grp * 64 + lane;
        uint32_t c = site < n_sites ? sym[(size_t)row * stride + site] : 0u;
        uint32_t u = to_upper(c);
        bool v = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
        bool hi = (u == 'G') | (u == 'T');
        bool lo = (u == 'C') | (u == 'T');
        uint64_t V = __ballot(v), H = __ballot(v && hi), Lo = __ballot(v && lo), Lc = __ballot(c != u);
        if (lane < 2) {
            uint32_t w = grp * 2 + lane;
            if (w < words) {
                uint32_t sh = lane * 32;
                packed[(size_t)row * words + w] = make_uint4((uint32_t)(V >> sh), (uint32_t)(H >> sh), (uint32_t)(Lo >> sh), (uint32_t)(Lc >> sh));
            }
        }
    }
}

struct DistArgs {
    const uint4 *packed;
    uint32_t n, words, n_tiles;     // n_tiles = ceil(n / DIST_TILE)
    uint32_t tile_rank, tile_nranks;
    uint64_t total_tiles;           // n_tiles * (n_tiles + 1) / 2
    uint32_t k_parts, k_chunk;      // split-K: the words are cut in k_parts ranges of k_chunk words (small matrices)
    int32_t *out;
};

__device__ __forceinline__ void tile_coords(uint64_t t, uint32_t nt, uint32_t &bi, uint32_t &bj) {
    // row-major enumeration of the upper triangle: row b starts at b*nt - b*(b-1)/2
    double fn = (double)nt + 0.5;
    int64_t b = (int64_t)(fn - sqrt(fn * fn - 2.0 * (double)t));
    if (b < 0) b = 0;
    if (b >= nt) b = nt - 1;
    auto start = [&](int64_t r) { return (uint64_t)r * nt - (uint64_t)r * (uint64_t)(r - 1) / 2; };
    while (b > 0 && start(b) > t) --b;
    while (b + 1 < (int64_t)nt && start(b + 1) <= t) ++b;
    bi = (uint32_t)b;
    bj = (uint32_t)(b + (t - start(b)));
}

// kSplit: a few tiles only (small sample counts): every tile is shared by k_parts workgroups, each sums its range of
// words and adds it to the (zeroed) output with integer atomics — the result does not depend on the order.
template <bool kSplit>
__global__ __launch_bounds__(DIST_THREADS, 3) void k_distance(DistArgs a) {
    // [buffer][operand][row][slot]; a wave's 64 DMA lanes fill 64 consecutive 16-byte slots
    __shared__ uint4 slab[2][2][DIST_TILE][DIST_KW];
    const uint32_t tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint64_t gg = blockIdx.x;; gg += gridDim.x) {
        const uint64_t g = kSplit ? gg / a.k_parts : gg;
        const uint32_t part = kSplit ? (uint32_t)(gg % a.k_parts) : 0;
        uint64_t t = (uint64_t)a.tile_rank + g * a.tile_nranks;
        if (t >= a.total_tiles) break;
        const uint32_t k_begin = kSplit ? part * a.k_chunk : 0;                      // multiples of DIST_KW
        const uint32_t k_end = kSplit ? (k_begin + a.k_chunk < a.words ? k_begin + a.k_chunk : a.words) : a.words;
        uint32_t bi, bj;
        tile_coords(t, a.n_tiles, bi, bj);
        const uint32_t r0 = bi * DIST_TILE, c0 = bj * DIST_TILE;
        int32_t acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0;

        // DMA: chunk p = e * 256 + tid of an operand's slab is slot (p & 3) of row (p >> 2).  Rows past the matrix read
        // the last row instead (their pairs are never stored).
        uint32_t row_of[2][DIST_TILE * DIST_KW / DIST_THREADS], kk_of[DIST_TILE * DIST_KW / DIST_THREADS];
#pragma unroll
        for (int e = 0; e < DIST_TILE * DIST_KW / DIST_THREADS; ++e) {
            const uint32_t p = e * DIST_THREADS + tid, rr = p >> 2;
            kk_of[e] = (p & 3u) ^ ((rr >> 2) & 3u);
            row_of[0][e] = r0 + rr < a.n ? r0 + rr : a.n - 1;
            row_of[1][e] = c0 + rr < a.n ? c0 + rr : a.n - 1;
        }
        auto request = [&](uint32_t k0, int buf) {
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int e = 0; e < DIST_TILE * DIST_KW / DIST_THREADS; ++e) {
                    const uint4 *gp = a.packed + ((size_t)row_of[op][e] * a.words + (k0 + kk_of[e]));
                    const uint32_t m0v = __builtin_amdgcn_readfirstlane(
                        (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)&slab[buf][op][0][0]) + (e * DIST_THREADS + wave * 64) * 16);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(m0v) : "memory");
                }
        };
        __syncthreads();                                            // the previous tile's last slab has been read
        request(k_begin, 0);
        int buf = 0;
        for (uint32_t k0 = k_begin; k0 < k_end; k0 += DIST_KW, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my part of this slab has landed ...
            __syncthreads();                                        // ... everybody's has, and slab k0 - KW has been read
            if (k0 + DIST_KW < k_end) request(k0 + DIST_KW, buf ^ 1);
#pragma unroll
            for (int kk = 0; kk < DIST_KW; ++kk) {
                uint4 x[8], y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const uint32_t r = ty + 16 * i; x[i] = slab[buf][0][r][kk ^ ((r >> 2) & 3u)]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) { const uint32_t c = tx + 16 * j; y[j] = slab[buf][1][c][kk ^ ((c >> 2) & 3u)]; }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // ((xh ^ yh) | (xl ^ yl)) & xv & yv as v_xor + two 3-input bit ops (truth tables 0xBE, 0x80)
                        const uint32_t u = __builtin_amdgcn_bitop3_b32(x[i].y, y[j].y, x[i].z ^ y[j].z, 0xBE);
                        acc[i][j] += __popc(__builtin_amdgcn_bitop3_b32(u, x[i].x, y[j].x, 0x80));
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t r = r0 + ty + 16 * i;
            if (r >= a.n) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t c = c0 + tx + 16 * j;
                if (c >= a.n) continue;
                if (kSplit) {
                    if (acc[i][j]) {
                        atomicAdd(&a.out[(size_t)r * a.n + c], acc[i][j]);
                        if (bi != bj) atomicAdd(&a.out[(size_t)c * a.n + r], acc[i][j]);
                    }
                } else {
                    a.out[(size_t)r * a.n + c] = acc[i][j];
                    if (bi != bj) a.out[(size_t)c * a.n + r] = acc[i][j];
                }
            }
        }
    }
}

// zero the tiles (and mirror images) a rank owns, before a split-K accumulation
__global__ __launch_bounds__(DIST_THREADS) void k_distance_zero(DistArgs a) {
    for (uint64_t g = blockIdx.x;; g += gridDim.x) {
        uint64_t t = (uint64_t)a.tile_rank + g * a.tile_nranks;
        if (t >= a.total_tiles) break;
        uint32_t bi, bj;
        tile_coords(t, a.n_tiles, bi, bj);
        for (uint32_t e = threadIdx.x; e < DIST_TILE * DIST_TILE; e += DIST_THREADS) {
            uint32_t r = bi * DIST_TILE + e / DIST_TILE, c = bj * DIST_TILE + e % DIST_TILE;
            if (r < a.n && c < a.n) { a.out[(size_t)r * a.n + c] = 0; a.out[(size_t)c * a.n + r] = 0; }
        }
    }
}

extern "C" {

int snpgpu_pack_matrix_dev(snpgpu_ctx *ctx, const uint8_t *d_symbols, uint32_t n_rows, uint32_t n_sites,
                           size_t row_stride, void *d_packed) {
    if (!ctx) return SNPGPU_E_ARG;
    if (n_rows == 0 || n_sites == 0) return SNPGPU_OK;
    if (!d_symbols || !d_packed || row_stride < n_sites) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "bad pack arguments");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    uint32_t words = padded_words(n_sites);                 // the padding words come out all zero (no valid site)
    uint64_t groups = (uint64_t)n_rows * ((words + 1) / 2);
    uint64_t blocks = (groups + 3) / 4, cap = (uint64_t)ctx->n_cu * 32;
    k_pack_matrix<<<(unsigned)(blocks < cap ? blocks : cap), 256, 0, ctx->stream>>>(d_symbols, n_rows, n_sites, row_stride, (uint4 *)d_packed, words);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_distance_packed_dev(snpgpu_ctx *ctx, const void *d_packed, uint32_t n_rows, uint32_t n_sites,
                               uint32_t tile_rank, uint32_t tile_nranks, int32_t *d_out) {
    if (!ctx) return SNPGPU_E_ARG;
    if (tile_nranks == 0 || tile_rank >= tile_nranks) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "bad tile rank %u of %u", tile_rank, tile_nranks);
    if (n_rows == 0) return SNPGPU_OK;
    if (!d_out || (n_sites && !d_packed)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null distance argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    DistArgs a;
    a.packed = (const uint4 *)d_packed;
    a.n = n_rows;
    a.words = padded_words(n_sites);
    a.n_tiles = (n_rows + DIST_TILE - 1) / DIST_TILE;
    a.tile_rank = tile_rank;
    a.tile_nranks = tile_nranks;
    a.total_tiles = (uint64_t)a.n_tiles * (a.n_tiles + 1) / 2;
    a.out = d_out;
    uint64_t mine = a.total_tiles > tile_rank ? (a.total_tiles - tile_rank + tile_nranks - 1) / tile_nranks : 0;
    if (mine == 0) return SNPGPU_OK;
    if (a.words == 0) {                                         // no sites at all: every distance of the rank's tiles is 0
        k_distance_zero<<<(unsigned)(mine < (uint64_t)ctx->n_cu * 64 ? mine : (uint64_t)ctx->n_cu * 64), DIST_THREADS, 0, ctx->stream>>>(a);
        HIP_TRY(ctx, hipGetLastError());
        return SNPGPU_OK;
    }
    uint64_t cap = (uint64_t)ctx->n_cu * 64;
    a.k_parts = 1;
    a.k_chunk = a.words;
    hipEvent_t ta = snpgpu_time_begin(ctx);
    if (mine < (uint64_t)ctx->n_cu && a.words >= 4 * DIST_KW) {        // too few tiles to fill the chip: split the word range
        uint64_t parts = ((uint64_t)ctx->n_cu * 2 + mine - 1) / mine;
        uint64_t max_parts = a.words / (2 * DIST_KW);
        if (parts > max_parts) parts = max_parts;
        a.k_chunk = (uint32_t)(((a.words + parts - 1) / parts + DIST_KW - 1) / DIST_KW * DIST_KW);
        a.k_parts = (a.words + a.k_chunk - 1) / a.k_chunk;
        k_distance_zero<<<(unsigned)mine, DIST_THREADS, 0, ctx->stream>>>(a);
        k_distance<true><<<(unsigned)(mine * a.k_parts), DIST_THREADS, 0, ctx->stream>>>(a);
    } else {
        k_distance<false><<<(unsigned)(mine < cap ? mine : cap), DIST_THREADS, 0, ctx->stream>>>(a);
    }
    snpgpu_time_end(ctx, SNPGPU_K_DISTANCE, ta);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_distance(snpgpu_ctx *ctx, const uint8_t *symbols, uint32_t n_rows, uint32_t n_sites, int32_t *out) {
    if (!ctx) return SNPGPU_E_ARG;
    if (n_rows == 0) return SNPGPU_OK;
    if (!out || (n_sites && !symbols)) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "null distance argument");
    HIP_TRY(ctx, snpgpu_enter(ctx));
    size_t sym_bytes = (size_t)n_rows * n_sites;
    size_t o_pack = (sym_bytes + 255) / 256 * 256;
    size_t pack_bytes = snpgpu_packed_row_bytes(n_sites) * n_rows;
    size_t o_out = o_pack + (pack_bytes + 255) / 256 * 256;
    size_t out_bytes = (size_t)n_rows * n_rows * 4;
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, o_out + out_bytes + 256);
    if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    char *b = (char *)d;
    hipStream_t st = ctx->stream;
    int rc = SNPGPU_OK;
    hipError_t he = hipSuccess;
    if (sym_bytes) he = hipMemcpyAsync(b, symbols, sym_bytes, hipMemcpyHostToDevice, st);
    if (he == hipSuccess && n_sites) rc = snpgpu_pack_matrix_dev(ctx, (const uint8_t *)b, n_rows, n_sites, n_sites, b + o_pack);
    if (he == hipSuccess && rc == SNPGPU_OK) rc = snpgpu_distance_packed_dev(ctx, b + o_pack, n_rows, n_sites, 0, 1, (int32_t *)(b + o_out));
    if (he == hipSuccess && rc == SNPGPU_OK) he = hipMemcpyAsync(out, b + o_out, out_bytes, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (he != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "distance failed: %s", hipGetErrorString(he));
    return rc;
}

}  // extern "C"
