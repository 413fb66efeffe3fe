// FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md, HBM: "other access widths ... are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Four read patterns over an 8 GiB buffer (32 x the Infinity Cache),
// each kernel reads a KNOWN set of bytes exactly once; run under `rocprofv3 --pmc FETCH_SIZE` and compare per kernel:
//   k_stream        16 B per lane, coalesced, every byte once                         (the guide's calibrated case: counter = bytes / 2)
//   k_window128     K2's staging: per lane the 128 bytes from a 16-byte aligned random address (8 x 16 B); windows are disjoint,
//                   one per 4 KiB page slot, so the 128-byte cache lines touched are countable: 1 line when the address is 128-aligned,
//                   else 2
//   k_scatter16     one 16-byte load per lane at a random 16-byte aligned address, one per 4 KiB slot (1 cache line each)
//   k_line128       one aligned 128-byte line per lane (8 x 16 B), one per 4 KiB slot
// and one write pattern, k_write128: 128 bytes per lane to consecutive aligned slots (what a compact [sample][site] line store
// would do).  Prints the byte counts each kernel touched; build: hipcc --offload-arch=gfx950 -O2 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_stream(const uint4 *buf, uint64_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
// slot s: the 4 KiB page s; the window starts at page + 16 * (mix(s) % 240)  (so that the 128 bytes stay inside the page)
__global__ void k_window128(const uint8_t *buf, uint64_t n_slots, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 *p = (const uint4 *)(buf + s * 4096 + 16 * (mix((uint32_t)s) % 240u));
#pragma unroll
        for (int k = 0; k < 8; ++k) { uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_scatter16(const uint8_t *buf, uint64_t n_slots, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = *(const uint4 *)(buf + s * 4096 + 16 * (mix((uint32_t)s) % 256u));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_line128(const uint8_t *buf, uint64_t n_slots, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 *p = (const uint4 *)(buf + s * 4096 + 128 * (mix((uint32_t)s) % 32u));
#pragma unroll
        for (int k = 0; k < 8; ++k) { uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_write128(uint4 *out, uint64_t n_slots) {
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = make_uint4((uint32_t)s, 1, 2, 3);
#pragma unroll
        for (int k = 0; k < 8; ++k) out[s * 8 + k] = v;
    }
}

int main(int argc, char **argv) {
    const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8;
    const uint64_t bytes = gib << 30, n_slots = bytes / 4096;
    uint8_t *buf = nullptr;
    uint4 *out = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, n_slots * 128));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    uint64_t two = 0;                                           // windows that lie in two 128-byte cache lines
    for (uint64_t s = 0; s < n_slots; ++s) {
        uint32_t x = (uint32_t)s; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        if ((x % 240u) % 8u) ++two;
    }
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = 256 * 8, block = 256;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
#define RUN(name, launch, touched, lines)                                                                                              \
        CHECK(hipEventRecord(a)); launch; CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b)); \
        if (rep) printf("%-12s bytes_asked %llu  bytes_of_128B_lines_touched %llu  bytes_of_64B_halves_touched %llu  ms %.3f  GB/s(asked) %.0f\n", name,   \
                        (unsigned long long)(touched), (unsigned long long)(lines), (unsigned long long)(halves), ms, (touched) / (ms * 1e6));
        uint64_t halves;
        halves = bytes;
        RUN("k_stream", (k_stream<<<grid, block>>>((const uint4 *)buf, bytes / 16, sink)), bytes, bytes)
        halves = 0;                                             // (counted on the device side would be exact; an estimate: 128 B at 16-B alignment covers 2 or 3 halves)
        for (uint64_t s = 0; s < n_slots && rep; ++s) { uint32_t x = (uint32_t)s; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; halves += ((x % 240u) % 4u) ? 3 * 64 : 2 * 64; }
        RUN("k_window128", (k_window128<<<grid, block>>>(buf, n_slots, sink)), n_slots * 128, (n_slots + two) * 128)
        halves = n_slots * 64;
        RUN("k_scatter16", (k_scatter16<<<grid, block>>>(buf, n_slots, sink)), n_slots * 16, n_slots * 128)
        halves = n_slots * 128;
        RUN("k_line128", (k_line128<<<grid, block>>>(buf, n_slots, sink)), n_slots * 128, n_slots * 128)
        halves = n_slots * 128;
        RUN("k_write128", (k_write128<<<grid, block>>>(out, n_slots)), n_slots * 128, n_slots * 128)
    }
    return 0;
}
