// Deterministic synthetic pileups generated on the device (bench + parity-test inputs).
// Not part of the reference; it only has to produce text that `samtools mpileup` could have produced:
//   <contig>\t<pos>\t<REF>\t<depth>\t<bases>\t<quals>\n   one line per covered position, ascending.
// Counter-based RNG keyed by (seed, sample, position, read, draw), so the length pass and the write pass see
// the same line and any host can regenerate the same bytes.

#include <math.h>
#include <string.h>

#include "internal.h"
#include "prims.h"

namespace {

__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct Rng {
    uint64_t key, ctr;
    __device__ Rng(uint64_t seed, uint32_t sample, uint32_t pos) : key(mix64(seed ^ mix64(((uint64_t)sample << 32) | pos))), ctr(0) {}
    __device__ uint32_t u32() { return (uint32_t)(mix64(key + (ctr++) * 0xD1342543DE82EF95ull) >> 32); }
    __device__ float uni() { return (u32() >> 8) * (1.0f / 16777216.0f); }
};

struct SynthDev {
    uint64_t seed;
    uint32_t sample, genome_len, n_clades;
    float p_same, p_other;
    char contig[32];
    uint32_t contig_len;
    float depth_cdf[256];
};

struct CountSink {
    uint32_t n = 0;
    __device__ void put(uint32_t) { ++n; }
};
struct WriteSink {
    uint8_t *p;
    __device__ void put(uint32_t c) { *p++ = (uint8_t)c; }
};

__device__ inline uint32_t other_base(uint32_t ref, uint32_t r) {
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    uint32_t k = r % 3, j = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        if ((uint32_t)acgt[i] == ref) continue;
        if (j++ == k) return acgt[i];
    }
    return 'N';
}

template <typename Sink>
__device__ void put_uint(Sink &s, uint32_t v) {
    char tmp[10];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.put(tmp[--n]);
}

template <typename Sink>
__device__ void gen_line(const SynthDev &P, uint32_t pos, uint32_t ref, uint32_t alt, Sink &s) {
    Rng rng(P.seed, P.sample, pos);
    float u = rng.uni();
    uint32_t lo = 0, hi = 255;                          // depth = first k with u <= cdf[k]
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (u <= P.depth_cdf[mid]) hi = mid; else lo = mid + 1; }
    uint32_t depth = lo > 250 ? 250 : lo;
    bool carrier = false;
    if (alt) {
        uint32_t site_clade = (uint32_t)(mix64(P.seed ^ (0xC1ADEull << 20) ^ pos) % P.n_clades);
        carrier = rng.uni() < ((P.sample % P.n_clades) == site_clade ? P.p_same : P.p_other);
    }
    if (depth == 0) return;
    for (uint32_t i = 0; i < P.contig_len; ++i) s.put(P.contig[i]);
    s.put('\t'); put_uint(s, pos); s.put('\t'); s.put(ref); s.put('\t'); put_uint(s, depth); s.put('\t');
    Rng qr(P.seed ^ 0x5157ull, P.sample, pos);
    for (uint32_t r = 0; r < depth; ++r) {
        uint32_t bits = rng.u32();
        bool fwd = bits & 1;
        float ub = rng.uni();
        uint32_t b;
        if (carrier && ub < 0.97f) b = fwd ? alt : to_lower(alt);
        else if (!carrier && ub < 0.005f) { uint32_t o = other_base(to_upper(ref), bits >> 8); b = fwd ? o : to_lower(o); }
        else if (ub > 0.9995f) b = '*';
        else b = fwd ? '.' : ',';
        if (((bits >> 1) & 0xFFFF) < 437) { s.put('^'); s.put(33 + ((bits >> 17) % 43)); }       // p ~ 1/150
        s.put(b);
        uint32_t e = rng.u32();
        if ((e & 0xFFFFF) < 1049) {                                                                  // p ~ 1e-3
            uint32_t len = 1 + ((e >> 20) % 3);
            s.put((e >> 24) & 1 ? '+' : '-');
            s.put('0' + len);
            for (uint32_t k = 0; k < len; ++k) {
                uint32_t nb = "ACGT"[(e >> (26 + 2 * k)) & 3];
                s.put(fwd ? nb : to_lower(nb));
            }
        }
        if (((e >> 4) & 0xFFFF) < 437 && b != '*') s.put('$');
    }
    s.put('\t');
    for (uint32_t r = 0; r < depth; ++r) {
        // ~N(35,5): Irwin-Hall of 4 uniforms (mean 2, sd 0.577)
        float z = (qr.uni() + qr.uni() + qr.uni() + qr.uni() - 2.0f) * 1.7320508f;
        int q = (int)rintf(35.0f + 5.0f * z);
        q = q < 2 ? 2 : (q > 41 ? 41 : q);
        s.put(33 + q);
    }
    s.put('\n');
}

__global__ void k_synth_ref(uint64_t seed, uint32_t genome_len, uint8_t *ref) {
    uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos > genome_len) return;
    ref[pos] = pos == 0 ? 'N' : "ACGT"[mix64(seed ^ (0x4EFull << 40) ^ pos) & 3];
}

__global__ void k_synth_len(SynthDev P, const uint8_t *ref, const uint8_t *alt, uint64_t *len) {
    uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (pos > P.genome_len) return;
    CountSink s;
    gen_line(P, pos, ref[pos], alt ? alt[pos] : 0u, s);
    len[pos - 1] = s.n;
}

__global__ void k_synth_write(SynthDev P, const uint8_t *ref, const uint8_t *alt, const uint64_t *off, uint8_t *out) {
    uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (pos > P.genome_len) return;
    WriteSink s{out + off[pos - 1]};
    gen_line(P, pos, ref[pos], alt ? alt[pos] : 0u, s);
}

}  // namespace

struct U64Add { __device__ uint64_t operator()(const uint64_t &a, const uint64_t &b) const { return a + b; } };

extern "C" {

int snpgpu_synth_reference_dev(snpgpu_ctx *ctx, uint64_t seed, uint32_t genome_len, uint8_t *d_ref) {
    if (!ctx || !d_ref) return SNPGPU_E_ARG;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    k_synth_ref<<<(genome_len + 1 + 255) / 256, 256, 0, ctx->stream>>>(seed, genome_len, d_ref);
    HIP_TRY(ctx, hipGetLastError());
    return SNPGPU_OK;
}

int snpgpu_synth_pileup_dev(snpgpu_ctx *ctx, const snpgpu_synth_params *p, const uint8_t *d_ref,
                            const uint8_t *d_site_alt, uint8_t *d_out, size_t capacity, size_t *out_nbytes) {
    if (!ctx || !p || !d_ref || !out_nbytes) return SNPGPU_E_ARG;
    *out_nbytes = 0;
    if (p->genome_len == 0) return SNPGPU_OK;
    HIP_TRY(ctx, snpgpu_enter(ctx));
    SynthDev P;
    P.seed = p->seed; P.sample = p->sample; P.genome_len = p->genome_len;
    P.n_clades = p->n_clades ? p->n_clades : 1;
    P.p_same = p->carrier_p_same_clade; P.p_other = p->carrier_p_other_clade;
    memset(P.contig, 0, sizeof P.contig);
    strncpy(P.contig, p->contig, sizeof P.contig - 1);
    P.contig_len = (uint32_t)strlen(P.contig);
    if (P.contig_len == 0) return snpgpu_set_error(ctx, SNPGPU_E_ARG, "empty contig name");
    // Poisson(mean) CDF
    double mean = p->mean_depth > 0 ? p->mean_depth : 30.0, term = exp(-mean), acc = 0;
    for (int k = 0; k < 256; ++k) {
        acc += term;
        P.depth_cdf[k] = (float)(acc > 1.0 ? 1.0 : acc);
        term *= mean / (k + 1);
    }
    P.depth_cdf[255] = 2.0f;
    // line lengths at d_len[1 ..], d_len[0] = 0: an inclusive prefix sum then leaves every line's offset in d_len[pos - 1]
    // and the total in d_len[genome_len]
    const uint64_t n_scan = (uint64_t)P.genome_len + 1;
    const size_t len_bytes = (8ull * n_scan + 255) / 256 * 256;
    void *scratch = nullptr;
    int rc = snpgpu_scratch(ctx, len_bytes + 8ull * prim_gscan_blocks(n_scan) + 256, &scratch);
    if (rc) return rc;
    uint64_t *d_len = (uint64_t *)scratch;
    uint64_t *d_aggr = (uint64_t *)((char *)scratch + len_bytes);
    hipStream_t st = ctx->stream;
    unsigned blocks = (P.genome_len + 255) / 256;
    HIP_TRY(ctx, hipMemsetAsync(d_len, 0, 8, st));
    k_synth_len<<<blocks, 256, 0, st>>>(P, d_ref, d_site_alt, d_len + 1);
    prim_inclusive_scan<uint64_t, U64Add>(st, d_len, d_len, n_scan, d_aggr, U64Add());
    uint64_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, d_len + P.genome_len, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return snpgpu_set_error(ctx, SNPGPU_E_HIP, "synth scan failed: %s", hipGetErrorString(e));
    *out_nbytes = (size_t)total;
    if (!d_out) return SNPGPU_OK;                      // size query
    if (total > capacity) return snpgpu_set_error(ctx, SNPGPU_E_NOMEM, "synthetic pileup needs %llu bytes, capacity %zu", (unsigned long long)total, capacity);
    k_synth_write<<<blocks, 256, 0, st>>>(P, d_ref, d_site_alt, d_len, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return SNPGPU_OK;
}

}  // extern "C"
