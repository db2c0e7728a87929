// ndt_kernels.h — device helpers shared by the residual kernels (ndt_residual.hip) and the PCL-compatibility
// kernels (ndt_pcl.hip): table reads and the fixed-order lane -> wave -> block reduction of 32 accumulators.
#pragma once
#include "ndt_internal.h"

#define SLOT_NONE 0xFFFFFFFFu

// neighbour offsets: centre, -x, +x, -y, +y, -z, +z  (same order as the oracle)
struct __attribute__((aligned(8))) Key2 { int64_t x, y; };   // two adjacent entries of the dense key array

__device__ inline bool cell_ok(int32_t v) { return (uint32_t)(v + NDT_KEY_BIAS) < (uint32_t)(2 * NDT_KEY_BIAS); }

// the 40 payload bytes of a slot: {valid, i22, off.x, off.y} {off.z, i00, i01, i02} {i11, i12}
struct Payload { int4 a; float4 b; float2 c; };
__device__ __forceinline__ Payload load_payload(const VoxSlot* __restrict__ slots, uint32_t slot) {
    const int4* sp = (const int4*)(slots + (slot != SLOT_NONE ? slot : 0u));   // slot 0 stands in for a miss: no predicated load
    Payload p;
    p.a = sp[0];
    p.b = *(const float4*)(sp + 1);
    p.c = *(const float2*)(sp + 2);
    return p;
}


// Butterfly reduce-scatter of 32 per-lane doubles over the 64 lanes of a wave: after it, lanes 2i and 2i+1 both
// hold the wave total of acc[i].  31 exchanges in a fixed tree.  The two widest levels use the gfx950 row-swap
// instructions (v_permlane32_swap / v_permlane16_swap: no select, no LDS crossbar); the rest are lane shuffles.
__device__ __forceinline__ double swap_add32(double a, double b) {   // lanes <32: a + a(lane+32); lanes >=32: b(lane-32) + b
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {   // even 16-lane rows keep a, odd rows keep b
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// value of lane (lane ^ BIT), BIT = 8, 4, 2 or 1, through DPP moves inside a row of 16 lanes (no LDS crossbar round trip, which
// ds_bpermute — what __shfl_xor compiles to — costs: four dependent ones on the critical tail of every residual launch).
// BIT = 8 / 4: lanes whose bit is set take the value BIT lanes below them (row_shr), the others BIT lanes above (row_shl);
// the bank mask selects which four-lane banks of each row a move writes.  BIT = 2 / 1: one quad permutation.
template <int BIT>
__device__ __forceinline__ int xor_lane_dpp(int v) {
    if constexpr (BIT == 8) {
        int t = __builtin_amdgcn_update_dpp(v, v, 0x118 /* row_shr:8 */, 0xf, 0xc, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x108 /* row_shl:8 */, 0xf, 0x3, false);
    } else if constexpr (BIT == 4) {
        int t = __builtin_amdgcn_update_dpp(v, v, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    } else if constexpr (BIT == 2) {
        return __builtin_amdgcn_update_dpp(v, v, 0x4e /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
    } else {
        static_assert(BIT == 1, "xor_lane_dpp: BIT must be 8, 4, 2 or 1");
        return __builtin_amdgcn_update_dpp(v, v, 0xb1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
    }
}
template <int BIT>
__device__ __forceinline__ double xor_lane(double x) {
#ifdef NDT_NO_DPP_BUTTERFLY
    return __shfl_xor(x, BIT);
#else
    return __hiloint2double(xor_lane_dpp<BIT>(__double2hiint(x)), xor_lane_dpp<BIT>(__double2loint(x)));
#endif
}
template <int HALF, int BIT>
__device__ __forceinline__ void rs_step(double* acc, int lane) {
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int j = 0; j < HALF; ++j) {
        const double send = up ? acc[j] : acc[j + HALF];
        const double keep = up ? acc[j + HALF] : acc[j];
        acc[j] = keep + xor_lane<BIT>(send);
    }
}
__device__ __forceinline__ double wave_reduce_scatter32(double* acc) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = swap_add32(acc[j], acc[j + 16]);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = swap_add16(acc[j], acc[j + 8]);
    rs_step<4, 8>(acc, lane);
    rs_step<2, 4>(acc, lane);
    rs_step<1, 2>(acc, lane);
    return acc[0] + xor_lane<1>(acc[0]);
}

// rows: partials[blockIdx.x][32]
template <int THREADS>
__device__ __forceinline__ void block_store_row(double* acc, double* __restrict__ row_out) {
    __shared__ double sm[THREADS / 64][ACC_N];
    const double v = wave_reduce_scatter32(acc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (!(lane & 1)) sm[w][lane >> 1] = v;
    __syncthreads();
    if (threadIdx.x < ACC_N) {
        double v = sm[0][threadIdx.x];
#pragma unroll
        for (int j = 1; j < THREADS / 64; ++j) v += sm[j][threadIdx.x];
        row_out[threadIdx.x] = v;
    }
}

