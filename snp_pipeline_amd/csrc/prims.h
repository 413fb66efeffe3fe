// Hand-written device-wide primitives shared by the small kernels of libsnpgpu.so (gfx950, 64-lane waves):
//   - wave / workgroup prefix sums on DPP row shifts (no LDS traffic inside a wave)
//   - a three-launch exclusive prefix sum over uint32 arrays of any length (reduce, spine, apply); nothing in it
//     synchronises with the host, the grand total is left in device memory
// Header-only: every translation unit that includes it gets its own copy of the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// Exclusive prefix sum of one value per thread over a workgroup of up to 1024 threads; `total` is the workgroup's sum.
// `lds` holds 17 words; the call contains two barriers and leaves `lds` reusable after the next barrier of the caller.
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t *lds, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    const uint32_t incl = wave_inclusive_sum(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const uint32_t w = lane < n_waves ? lds[lane] : 0u;
        const uint32_t wi = wave_inclusive_sum(w);
        if (lane < n_waves) lds[lane] = wi - w;              // exclusive offset of every wave
        if (lane == 63) lds[16] = wi;                        // lane 63 holds the sum of all waves
    }
    __syncthreads();
    total = lds[16];
    return lds[wave] + incl - v;
}

#define PRIM_SCAN_THREADS 1024
#define PRIM_SCAN_ITEMS 8                                    // per thread: a workgroup covers 8192 elements
#define PRIM_SCAN_BLOCK (PRIM_SCAN_THREADS * PRIM_SCAN_ITEMS)

static inline uint32_t prim_scan_blocks(uint64_t n) { return (uint32_t)((n + PRIM_SCAN_BLOCK - 1) / PRIM_SCAN_BLOCK); }
// words of device workspace an exclusive scan of n elements needs (block sums + total)
static inline size_t prim_scan_workspace_words(uint64_t n) { return (size_t)prim_scan_blocks(n) + 2; }

__global__ __launch_bounds__(PRIM_SCAN_THREADS) static void k_prim_scan_reduce(const uint32_t *in, uint64_t n, uint32_t *block_sums) {
    __shared__ uint32_t lds[17];
    const uint64_t base = (uint64_t)blockIdx.x * PRIM_SCAN_BLOCK + (uint64_t)threadIdx.x * PRIM_SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < PRIM_SCAN_ITEMS; ++k) s += base + k < n ? in[base + k] : 0u;
    uint32_t total;
    (void)block_exclusive_sum(s, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// One workgroup: exclusive scan of the block sums in place; total[0] = grand total.
__global__ __launch_bounds__(PRIM_SCAN_THREADS) static void k_prim_scan_spine(uint32_t *block_sums, uint32_t n_blocks, uint32_t *total) {
    __shared__ uint32_t lds[17];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += PRIM_SCAN_THREADS) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        uint32_t t;
        const uint32_t ex = block_exclusive_sum(v, lds, t);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += t;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

__global__ __launch_bounds__(PRIM_SCAN_THREADS) static void k_prim_scan_apply(const uint32_t *in, uint32_t *out, uint64_t n, const uint32_t *block_sums) {
    __shared__ uint32_t lds[17];
    const uint64_t base = (uint64_t)blockIdx.x * PRIM_SCAN_BLOCK + (uint64_t)threadIdx.x * PRIM_SCAN_ITEMS;
    uint32_t v[PRIM_SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < PRIM_SCAN_ITEMS; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
    uint32_t total;
    uint32_t run = block_sums[blockIdx.x] + block_exclusive_sum(s, lds, total);
#pragma unroll
    for (int k = 0; k < PRIM_SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// out[i] = in[0] + ... + in[i-1] (in == out allowed); ws: prim_scan_workspace_words(n) words; the grand total is left in
// ws[prim_scan_blocks(n)] (pointer returned through d_total when not null).  Three launches, no host synchronisation.
static inline void prim_exclusive_scan_u32(hipStream_t st, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *ws, uint32_t **d_total) {
    const uint32_t nb = prim_scan_blocks(n);
    uint32_t *total = ws + nb;
    if (d_total) *d_total = total;
    if (n == 0) { (void)hipMemsetAsync(total, 0, 4, st); return; }
    k_prim_scan_reduce<<<nb, PRIM_SCAN_THREADS, 0, st>>>(in, n, ws);
    k_prim_scan_spine<<<1, PRIM_SCAN_THREADS, 0, st>>>(ws, nb, total);
    k_prim_scan_apply<<<nb, PRIM_SCAN_THREADS, 0, st>>>(in, out, n, ws);
}
