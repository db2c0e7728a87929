// ndt_device.h — device-side data layout and small math shared by the engine's kernels.
//
// HBM layout:
//   source cloud    : three float32 arrays x[], y[], z[] (structure of arrays, coalesced 4-byte lanes)
//   voxel hash      : open addressing, linear probing, power-of-two capacity; a dense int64 key array (what a
//                     probe reads) and one 64-byte VoxSlot per slot: the 40 payload bytes the residual
//                     kernel needs plus the slots of the six face neighbours, so that only the centre
//                     cell of a point has to be hashed.  Cells that touch a voxel but hold no points are
//                     kept as "ghost" entries (valid = 0, count = 0) to carry those links.
//   voxel statistics: per slot, float64 running mean offset (3) and covariance (6) + count, used only
//                     by map build / incremental merge, never by the residual kernel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NDT_KEY_BIAS (1 << 20)
#define NDT_KEY_EMPTY INT64_MIN

#define NDT_LINK_NONE 0xFFFFFFFFu
struct __attribute__((aligned(64))) VoxSlot {
    int32_t valid;     // 1: n >= min_pts and the regularised inverse is finite (0 for ghost cells)
    float i22;         // info(2,2)
    float off[3];      // mean - key*res, float32
    float i00, i01, i02, i11, i12;   // info upper triangle: 00 01 02 11 12 (22 above)   -> bytes 0..39 = the 40-byte payload
    uint32_t link[6];  // slot of the face neighbour -x +x -y +y -z +z, NDT_LINK_NONE if that cell is not in the table
};
static_assert(sizeof(VoxSlot) == 64, "VoxSlot must be one 64-byte segment");

// accumulator row: 21 upper-triangle H (row-major), 6 b, chi2, n_eff, hits, points
#define ACC_N 32
#define ACC_CHI2 27
#define ACC_NEFF 28
#define ACC_HITS 29
#define ACC_PTS 30

struct GnState {          // one per alignment problem, lives in HBM for the whole Gauss-Newton loop
    double R[9];          // row-major rotation
    double t[3];
    double dx[6];
    double acc[ACC_N];    // reduced accumulators of the last evaluation
    double dx_norm2;      // |dx|^2 of the last step
    int32_t iter;         // Gauss-Newton steps taken
    int32_t done;         // 1: loop finished (converged, cap reached or failed) -> later launches are no-ops
    int32_t converged;
    int32_t status;       // NDT_OK or a negative ndt error
};

struct KeyParams {
    float inv_leaf_f32;
    double inv_res_d;
    double res_d;
    int32_t key_rule;     // 0 FLOOR_F32, 1 ROUND_F64
};

struct EvalParams {
    KeyParams kp;
    double outlier_th;
    int32_t nearby;       // 1 or 7
};

struct SolveParams {
    double eps;
    int32_t max_iterations;
    int32_t min_effective_pts;
};

__host__ __device__ inline int64_t ndt_pack(int32_t ix, int32_t iy, int32_t iz) {
    return ((int64_t)(ix + NDT_KEY_BIAS) << 42) | ((int64_t)(iy + NDT_KEY_BIAS) << 21) |
           (int64_t)(iz + NDT_KEY_BIAS);
}
__host__ __device__ inline void ndt_unpack(int64_t k, int32_t& ix, int32_t& iy, int32_t& iz) {
    ix = (int32_t)((k >> 42) & 0x1FFFFF) - NDT_KEY_BIAS;
    iy = (int32_t)((k >> 21) & 0x1FFFFF) - NDT_KEY_BIAS;
    iz = (int32_t)(k & 0x1FFFFF) - NDT_KEY_BIAS;
}
__host__ __device__ inline uint64_t ndt_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// Table hash: per-axis multiplicative spatial hash (a face neighbour's hash is one add and two xors away from
// the centre's) followed by one 32-bit avalanche round.  Capacity is a power of two <= 2^31.
#define NDT_HP0 73856093u
#define NDT_HP1 19349669u
#define NDT_HP2 83492791u
__host__ __device__ inline uint32_t ndt_hash_fin(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13;
    return h;
}
__host__ __device__ inline uint32_t ndt_hash_cell(int32_t ix, int32_t iy, int32_t iz) {
    return ndt_hash_fin(((uint32_t)ix * NDT_HP0) ^ ((uint32_t)iy * NDT_HP1) ^ ((uint32_t)iz * NDT_HP2));
}
__host__ __device__ inline uint32_t ndt_hash_key(int64_t k) {
    int32_t x, y, z;
    ndt_unpack(k, x, y, z);
    return ndt_hash_cell(x, y, z);
}
__device__ inline bool ndt_in_range(double v) { return v >= -(double)NDT_KEY_BIAS && v < (double)NDT_KEY_BIAS; }

// Arithmetic contract with the oracle: plain float64 multiplies and adds in this order, nothing fused by
// the compiler (the library is built with -ffp-contract=off); the explicit fma() calls in the residual kernels' gate are
// part of the contract and written the same way in the oracle.
__device__ inline void ndt_xform(const double* R, const double* t, float xf, float yf, float zf, double q[3]) {
    double x = xf, y = yf, z = zf;
#pragma unroll
    for (int j = 0; j < 3; ++j) q[j] = ((R[3 * j] * x + R[3 * j + 1] * y) + R[3 * j + 2] * z) + t[j];
}
// cell of a float32 map point: PCL grid rule floor(p_f32 * inv_leaf_f32) in float32, or round-half-away in f64
__device__ inline bool ndt_target_cell(const KeyParams& kp, float x, float y, float z, int32_t c[3]) {
    float p[3] = {x, y, z};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double v = kp.key_rule == 0 ? (double)floorf(p[j] * kp.inv_leaf_f32) : round((double)p[j] * kp.inv_res_d);
        if (!ndt_in_range(v)) return false;   // also catches NaN / inf
        c[j] = (int32_t)v;
    }
    return true;
}
// cell of a float64 transformed source point: same rule evaluated in float64
__device__ inline bool ndt_source_cell(const KeyParams& kp, const double q[3], int32_t c[3]) {
    double v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = q[j] * kp.inv_res_d;
    // the rule is uniform over the launch: a real branch, not both roundings and a select (the empty asm keeps the
    // compiler from speculating the rarely used half-away rounding into the hot loop)
    if (kp.key_rule == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = floor(v[j]);
    } else {
        asm volatile("");
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = round(v[j]);
    }
    const bool ok = ndt_in_range(v[0]) && ndt_in_range(v[1]) && ndt_in_range(v[2]);      // false for NaN / inf too
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] = ok ? (int32_t)v[j] : 0;
    return ok;
}
