// fetch_calibration.hip — what does rocprofv3's FETCH_SIZE count for THIS engine's gather?  
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE for wide coalesced streaming reads only (16 B per lane: the counter reports
// exactly half the bytes) and says other access widths are uncalibrated.  The residual kernel does not stream: a lane reads
// the 64-byte slot of its point's cell and 40 of the 64 bytes of six neighbour slots.  This program issues reads of KNOWN
// byte counts against a table far beyond every cache (>= 1 GiB; every slot touched at most once per kernel), one kernel per
// access pattern, each launched once so that a `rocprofv3 --pmc FETCH_SIZE` (or TCC_EA0_RDREQ_sum ...) pass lists one row per
// pattern; tools/fetch_calibration.py divides the known bytes by the counter.
//   k_cal_stream16   : 16 B per lane, coalesced, n_slots x 64 B in all                       (control: factor 2.0 expected)
//   k_cal_slot64     : one distinct random slot per lane, all 64 bytes (4 x 16 B)            (the centre-cell read)
//   k_cal_slot40     : one distinct random slot per lane, 40 of its 64 bytes (16 + 16 + 8)   (a neighbour read)
//   k_cal_pattern    : the kernel's own pattern per lane: one 64-B slot + six 40-of-64-B slots, seven distinct random slots
//   k_cal_pattern_coh: the same with LiDAR-like coherence: the 64 lanes of a wave share 8 random "cells" (8 lanes per cell),
//                      so a wave touches 56 distinct lines instead of 448
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calibration tools/ubench/fetch_calibration.hip
// Run  : tools/ubench/fetch_calibration            (prints the known byte counts as JSON on stdout)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((aligned(64))) Slot { int4 a; float4 b; int4 c, d; };
static_assert(sizeof(Slot) == 64, "64-byte slots, as VoxSlot");

__device__ __forceinline__ float use40(const Slot* s) {
    const int4 a = *(const int4*)&s->a;
    const float4 b = s->b;
    const float2 c = *(const float2*)&s->c;
    return (float)a.x + b.x + c.x + c.y + (float)a.w + b.w;
}
__device__ __forceinline__ float use64(const Slot* s) {
    const int4 a = s->a; const float4 b = s->b; const int4 c = s->c, d = s->d;
    return (float)a.x + b.x + (float)c.x + (float)d.x + (float)a.w + b.w + (float)c.w + (float)d.w;
}

__global__ __launch_bounds__(256) void k_cal_stream16(const int4* __restrict__ p, size_t n16, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const int4 v = p[i]; acc += (float)(v.x ^ v.w); }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_slot64(const Slot* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) acc = use64(t + idx[i]);
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_slot40(const Slot* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) acc = use40(t + idx[i]);
    if (acc == 123.456f) out[0] = acc;
}
// lane i: slots idx[7 i] (64 B) and idx[7 i + 1 .. 7 i + 6] (40 B each)
__global__ __launch_bounds__(256) void k_cal_pattern(const Slot* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        uint32_t s[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) s[k] = idx[7 * i + k];
        acc = use64(t + s[0]);
#pragma unroll
        for (int k = 1; k < 7; ++k) acc += use40(t + s[k]);
    }
    if (acc == 123.456f) out[0] = acc;
}
// the same, eight lanes per "cell": lane i uses the index group of lane-group i / 8
__global__ __launch_bounds__(256) void k_cal_pattern_coh(const Slot* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        const size_t g = i / 8;
        uint32_t s[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) s[k] = idx[7 * g + k];
        acc = use64(t + s[0]);
#pragma unroll
        for (int k = 1; k < 7; ++k) acc += use40(t + s[k]);
    }
    if (acc == 123.456f) out[0] = acc;
}
// timing mode: M distinct random 64-byte slots per lane, all 4 M loads issued before any is used
template <int M>
__global__ __launch_bounds__(256) void k_cal_mlp(const Slot* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        uint32_t s[M];
#pragma unroll
        for (int k = 0; k < M; ++k) s[k] = idx[(size_t)k * n + i];
#pragma unroll
        for (int k = 0; k < M; ++k) acc += use64(t + s[k]);
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_cal_fill(int4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_int4((int)i, 1, 2, 3);
}

int main(int argc, char** argv) {
    // default: 32 Mi slots x 64 B = 2 GiB, far beyond L2 (32 MB) and the 256 MB Infinity Cache; 4 Mi lanes per gather kernel
    // (7 x 4 Mi = 28 Mi distinct slots < n_slots).  `time <log2 slots>`: another table size (lanes = slots / 8)
    const int lg = argc > 2 ? atoi(argv[2]) : 25;
    if (lg < 16 || lg > 27) { fprintf(stderr, "log2 slots must be 16..27\n"); return 1; }
    const size_t n_slots = (size_t)1 << lg;
    const size_t n_lanes = n_slots >> 3;
    Slot* table = nullptr; uint32_t* idx = nullptr; float* out = nullptr;
    CHECK(hipMalloc(&table, n_slots * sizeof(Slot)));
    CHECK(hipMalloc(&idx, 7 * n_lanes * 4));
    CHECK(hipMalloc(&out, 64));
    k_cal_fill<<<4096, 256>>>((int4*)table, n_slots * 4);
    CHECK(hipDeviceSynchronize());
    // 7 x n_lanes DISTINCT random slots (a random sample without replacement: every line is touched at most once per kernel)
    std::vector<uint32_t> perm(n_slots);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(20241220);
    for (size_t i = 0; i < 7 * n_lanes; ++i) { std::uniform_int_distribution<size_t> d(i, n_slots - 1); std::swap(perm[i], perm[d(rng)]); }
    CHECK(hipMemcpy(idx, perm.data(), 7 * n_lanes * 4, hipMemcpyHostToDevice));
    const unsigned nb = (unsigned)(n_lanes / 256);
    if (argc > 1 && argv[1][0] == 't') {
        // `fetch_calibration time`: the random-64-byte-line gather rate of this GPU by loads in flight per lane (HIP events, no
        // profiler; the 0.27 - 1.9 GB a kernel touches are far beyond the 256 MB Infinity Cache, so repeats stay uncached)
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, int m, auto&& launch) {
            launch(); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) launch();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / 5.0, bytes = (double)n_lanes * m * 64.0;
            printf("%-34s %8.1f us  %6.2f TB/s of 64-byte lines (%.0f MB per launch)\n", name, us, bytes / us / 1e6, bytes / 1e6);
        };
        printf("table %zu MB\n", n_slots * 64 >> 20);
        timeit("stream 16 B/lane (whole table)", 8, [&] { k_cal_stream16<<<8192, 256>>>((const int4*)table, n_slots * 4, out); });
        timeit("random slot, 1 in flight per lane", 1, [&] { k_cal_mlp<1><<<nb, 256>>>(table, idx, n_lanes, out); });
        timeit("random slot, 2 in flight per lane", 2, [&] { k_cal_mlp<2><<<nb, 256>>>(table, idx, n_lanes, out); });
        timeit("random slot, 4 in flight per lane", 4, [&] { k_cal_mlp<4><<<nb, 256>>>(table, idx, n_lanes, out); });
        timeit("random slot, 7 in flight per lane", 7, [&] { k_cal_mlp<7><<<nb, 256>>>(table, idx, n_lanes, out); });
        timeit("engine pattern (64 + 6 x 40 B)", 7, [&] { k_cal_pattern<<<nb, 256>>>(table, idx, n_lanes, out); });
        return 0;
    }
    // a 2 GiB stream between the kernels evicts whatever the previous one left in the caches
    auto flush = [&]() { k_cal_fill<<<4096, 256>>>((int4*)table, n_slots * 4); };
    flush(); k_cal_stream16<<<8192, 256>>>((const int4*)table, n_slots * 4, out);
    flush(); k_cal_slot64<<<nb, 256>>>(table, idx, n_lanes, out);
    flush(); k_cal_slot40<<<nb, 256>>>(table, idx, n_lanes, out);
    flush(); k_cal_pattern<<<nb, 256>>>(table, idx, n_lanes, out);
    flush(); k_cal_pattern_coh<<<nb, 256>>>(table, idx, n_lanes, out);
    CHECK(hipDeviceSynchronize());
    // known bytes: "requested" = what the lanes load; "lines" = distinct 64-byte lines touched; index reads listed apart
    printf("{\"k_cal_stream16\": {\"requested\": %zu, \"lines\": %zu, \"index\": 0},\n", n_slots * 64, n_slots * 64);
    printf(" \"k_cal_slot64\": {\"requested\": %zu, \"lines\": %zu, \"index\": %zu},\n", n_lanes * 64, n_lanes * 64, n_lanes * 4);
    printf(" \"k_cal_slot40\": {\"requested\": %zu, \"lines\": %zu, \"index\": %zu},\n", n_lanes * 40, n_lanes * 64, n_lanes * 4);
    printf(" \"k_cal_pattern\": {\"requested\": %zu, \"lines\": %zu, \"index\": %zu},\n", n_lanes * (64 + 6 * 40), n_lanes * 7 * 64, n_lanes * 28);
    printf(" \"k_cal_pattern_coh\": {\"requested\": %zu, \"lines\": %zu, \"index\": %zu}}\n", n_lanes * (64 + 6 * 40), n_lanes / 8 * 7 * 64, n_lanes / 8 * 28);
    CHECK(hipFree(table)); CHECK(hipFree(idx)); CHECK(hipFree(out));
    return 0;
}
