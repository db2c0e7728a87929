// ndt_residual.hip — the per-iteration residual / accumulation pass and the Gauss-Newton update
// (replaces the body of ndt_.align(), src/slam/src/process.cc:94, in the
// Gauss-Newton / NEARBY1-NEARBY7 form called at src/slam/reference/lio_preinteg.cc:116-121).
//
// k_residual: one source point per lane.  q' = R q + t in float64, cell = rule(q'/res), K hash probes
// (centre + 6 face neighbours), Mahalanobis residual against the 40-byte voxel payload, chi^2 gate.  The
// per-point sums S = sum Sigma^-1 and g = sum Sigma^-1 e are folded once per point through the 3x6
// Jacobian J = [-R hat(q) | I] into 21 + 6 accumulators per lane; lanes -> wave (shuffle tree) -> block
// (LDS) -> one 32-double row per block in HBM.  k_reduce_solve adds the rows in a fixed order and takes
// the Gauss-Newton step on the device, so an iteration never returns to the host.
// Bound: HBM/L2 gather (12 + 8K + 40h algorithmic bytes per point); no MFMA.
#include "ndt_internal.h"

static const __device__ int NB7[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};

#define RES_THREADS 256
#define NACC 30   // 21 H + 6 b + chi2 + n_eff + hits

template <int K>
__global__ __launch_bounds__(RES_THREADS) void k_residual(const float* __restrict__ xs, const float* __restrict__ ys,
                                                          const float* __restrict__ zs, int n, const VoxSlot* __restrict__ slots,
                                                          uint64_t mask, const GnState* __restrict__ st, EvalParams ep,
                                                          double* __restrict__ partials, int force, int64_t* keys_out, uint8_t* mask_out) {
    if (!force && st->done) return;
    double R[9], t[3];
#pragma unroll
    for (int j = 0; j < 9; ++j) R[j] = st->R[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] = st->t[j];

    double acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.0;

    for (int i = blockIdx.x * RES_THREADS + threadIdx.x; i < n; i += gridDim.x * RES_THREADS) {
        const float xf = xs[i], yf = ys[i], zf = zs[i];
        double q[3];
        ndt_xform(R, t, xf, yf, zf, q);
        int32_t c[3];
        if (!ndt_source_cell(ep.kp, q, c)) {   // non-finite or out of the key range: the point is skipped
            if (keys_out) { keys_out[i] = NDT_KEY_EMPTY; mask_out[i] = 0; }
            continue;
        }
        double S[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        unsigned hitmask = 0;
#pragma unroll
        for (int pr = 0; pr < K; ++pr) {
            const int32_t n0 = c[0] + NB7[pr][0], n1 = c[1] + NB7[pr][1], n2 = c[2] + NB7[pr][2];
            if (!ndt_in_range((double)n0) || !ndt_in_range((double)n1) || !ndt_in_range((double)n2)) continue;
            const int64_t key = ndt_pack(n0, n1, n2);
            uint64_t slot = ndt_mix64((uint64_t)key) & mask;
            int4 head;
            bool found = false;
            for (;;) {
                head = *(const int4*)&slots[slot];
                const int64_t k = ((int64_t)(uint32_t)head.y << 32) | (uint32_t)head.x;
                if (k == key) { found = true; break; }
                if (k == NDT_KEY_EMPTY) break;
                slot = (slot + 1) & mask;
            }
            if (!found || !head.z) continue;
            const float4 p1 = *((const float4*)&slots[slot] + 1);   // off.xyz, i00
            const float4 p2 = *((const float4*)&slots[slot] + 2);   // i01 i02 i11 i12
            const double i00 = p1.w, i01 = p2.x, i02 = p2.y, i11 = p2.z, i12 = p2.w, i22 = __int_as_float(head.w);
            acc[29] += 1.0;
            const double e0 = (q[0] - (double)n0 * ep.kp.res_d) - (double)p1.x;
            const double e1 = (q[1] - (double)n1 * ep.kp.res_d) - (double)p1.y;
            const double e2 = (q[2] - (double)n2 * ep.kp.res_d) - (double)p1.z;
            const double g0 = (i00 * e0 + i01 * e1) + i02 * e2;
            const double g1 = (i01 * e0 + i11 * e1) + i12 * e2;
            const double g2 = (i02 * e0 + i12 * e1) + i22 * e2;
            const double chi2 = (e0 * g0 + e1 * g1) + e2 * g2;
            if (isnan(chi2) || chi2 > ep.outlier_th) continue;
            hitmask |= 1u << pr;
            S[0] += i00; S[1] += i01; S[2] += i02; S[3] += i11; S[4] += i12; S[5] += i22;
            g[0] += g0; g[1] += g1; g[2] += g2;
            acc[27] += chi2;
            acc[28] += 1.0;
        }
        if (keys_out) { keys_out[i] = ndt_pack(c[0], c[1], c[2]); mask_out[i] = (uint8_t)hitmask; }
        if (!hitmask) continue;
        // A = -R hat(q):  H = [A^T S A, A^T S; S A, S],  b = [-A^T g; -g]
        const double x = xf, y = yf, z = zf;
        double A[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            A[r][0] = -(R[3 * r + 1] * z - R[3 * r + 2] * y);
            A[r][1] = -(R[3 * r + 2] * x - R[3 * r + 0] * z);
            A[r][2] = -(R[3 * r + 0] * y - R[3 * r + 1] * x);
        }
        const double Sm[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
        double SA[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) SA[r][cc] = (Sm[r][0] * A[0][cc] + Sm[r][1] * A[1][cc]) + Sm[r][2] * A[2][cc];
        // upper triangle of the 6x6, row-major: row r holds columns r..5
        // rows 0..2 (rotation): [A^T S A](r, r..2), [A^T S](r, 0..2)
        int tix = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int cc = r; cc < 3; ++cc) acc[tix++] += (A[0][r] * SA[0][cc] + A[1][r] * SA[1][cc]) + A[2][r] * SA[2][cc];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) acc[tix++] += SA[cc][r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = r; cc < 3; ++cc) acc[tix++] += Sm[r][cc];
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[21 + r] += -((A[0][r] * g[0] + A[1][r] * g[1]) + A[2][r] * g[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[24 + r] += -g[r];
    }

    // lanes -> wave: fixed shuffle tree
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        double v = acc[j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o);
        acc[j] = v;
    }
    __shared__ double sm[RES_THREADS / 64][ACC_N];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) sm[w][j] = acc[j];
        sm[w][30] = 0.0; sm[w][31] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x < ACC_N) {
        double v = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
        partials[(size_t)blockIdx.x * ACC_N + threadIdx.x] = v;
    }
}

int residual_grid(size_t n) {
    size_t g = (n + RES_THREADS - 1) / RES_THREADS;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

void launch_residual(const float* x, const float* y, const float* z, int n, const VoxSlot* slots, uint64_t mask,
                     const GnState* st, const EvalParams& ep, double* partials, int grid, int force,
                     int64_t* keys_out, uint8_t* mask_out, hipStream_t s) {
    if (ep.nearby == 7)
        k_residual<7><<<grid, RES_THREADS, 0, s>>>(x, y, z, n, slots, mask, st, ep, partials, force, keys_out, mask_out);
    else
        k_residual<1><<<grid, RES_THREADS, 0, s>>>(x, y, z, n, slots, mask, st, ep, partials, force, keys_out, mask_out);
}

// ------------------------------------------------------------------ fixed-order final sum + Gauss-Newton step

__device__ inline bool chol_solve6(const double* H, const double* b, double* x) {
    double L[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = H[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i][i] = sqrt(s);
            } else L[i][j] = s / L[j][j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
    return true;
}

// R <- R Exp(w), t <- t + v  (Exp: Rodrigues as src/slam/reference/math_utils.h:256-269, identity below 1e-7 rad)
__device__ void gn_step(GnState* st, const SolveParams& sp) {
    const double* acc = st->acc;
    if ((long long)acc[ACC_NEFF] < (long long)sp.min_effective_pts) { st->status = NDT_ERR_TOO_FEW_POINTS; st->done = 1; return; }
    double H[36], b[6], dx[6];
    int t = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c, ++t) { H[r * 6 + c] = acc[t]; H[c * 6 + r] = acc[t]; }
    for (int r = 0; r < 6; ++r) b[r] = acc[21 + r];
    if (!chol_solve6(H, b, dx)) { st->status = NDT_ERR_SINGULAR; st->done = 1; return; }
    double nrm = 0.0;
    for (int j = 0; j < 6; ++j) { st->dx[j] = dx[j]; nrm += dx[j] * dx[j]; }
    nrm = sqrt(nrm);
    double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double th = sqrt((dx[0] * dx[0] + dx[1] * dx[1]) + dx[2] * dx[2]);
    if (th > 1e-7) {
        const double a0 = dx[0] / th, a1 = dx[1] / th, a2 = dx[2] / th;
        const double Km[9] = {0, -a2, a1, a2, 0, -a0, -a1, a0, 0};
        const double s = sin(th), c1 = 1.0 - cos(th);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double kk = (Km[3 * i] * Km[j] + Km[3 * i + 1] * Km[3 + j]) + Km[3 * i + 2] * Km[6 + j];
                E[3 * i + j] += s * Km[3 * i + j] + c1 * kk;
            }
    }
    double Rn[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[3 * i + j] = (st->R[3 * i] * E[j] + st->R[3 * i + 1] * E[3 + j]) + st->R[3 * i + 2] * E[6 + j];
    for (int j = 0; j < 9; ++j) st->R[j] = Rn[j];
    for (int j = 0; j < 3; ++j) st->t[j] += dx[3 + j];
    st->dx_norm = nrm;
    st->iter += 1;
    if (nrm < sp.eps) { st->converged = 1; st->done = 1; }
    if (st->iter >= sp.max_iterations) st->done = 1;
}

#define RED_THREADS 1024
// thread (row r = tid / 32, column c = tid % 32) adds partial rows r, r+32, ... in order; rows are then
// combined by a fixed binary tree in LDS.
__global__ __launch_bounds__(RED_THREADS) void k_reduce_solve(const double* __restrict__ partials, int n_partials, GnState* st,
                                                              SolveParams sp, int do_solve, int force) {
    if (!force && st->done) return;
    __shared__ double sm[32][ACC_N + 1];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    double v = 0.0;
    for (int i = r; i < n_partials; i += 32) v += partials[(size_t)i * ACC_N + c];
    sm[r][c] = v;
    __syncthreads();
    for (int o = 16; o >= 1; o >>= 1) {
        if (r < o) sm[r][c] += sm[r + o][c];
        __syncthreads();
    }
    if (threadIdx.x < ACC_N) st->acc[threadIdx.x] = sm[0][threadIdx.x];
    __syncthreads();
    if (do_solve && threadIdx.x == 0) {
        __threadfence_block();
        gn_step(st, sp);
    }
}
// multi-GPU form: the local sum goes to a row that RCCL all-reduces, and the step reads the reduced row
__global__ __launch_bounds__(RED_THREADS) void k_reduce_rows(const double* __restrict__ partials, int n_partials, const GnState* st,
                                                             double* row_out, int force) {
    __shared__ double sm[32][ACC_N + 1];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const bool live = force || !st->done;     // a finished state contributes zeros (its partials are stale)
    double v = 0.0;
    if (live) for (int i = r; i < n_partials; i += 32) v += partials[(size_t)i * ACC_N + c];
    sm[r][c] = v;
    __syncthreads();
    for (int o = 16; o >= 1; o >>= 1) {
        if (r < o) sm[r][c] += sm[r + o][c];
        __syncthreads();
    }
    if (threadIdx.x < ACC_N) row_out[threadIdx.x] = sm[0][threadIdx.x];
}
__global__ void k_solve_only(GnState* st, const double* reduced, SolveParams sp, int do_solve, int force) {
    if (!force && st->done) return;
    if (threadIdx.x < ACC_N) st->acc[threadIdx.x] = reduced[threadIdx.x];
    __syncthreads();
    if (do_solve && threadIdx.x == 0) gn_step(st, sp);
}

void launch_reduce_solve(const double* partials, int n_partials, GnState* st, const SolveParams& sp, int do_solve, int force, hipStream_t s) {
    k_reduce_solve<<<1, RED_THREADS, 0, s>>>(partials, n_partials, st, sp, do_solve, force);
}
void launch_reduce_rows(const double* partials, int n_partials, const GnState* st, double* row_out, int force, hipStream_t s) {
    k_reduce_rows<<<1, RED_THREADS, 0, s>>>(partials, n_partials, st, row_out, force);
}
void launch_solve_only(GnState* st, const double* reduced, const SolveParams& sp, int do_solve, int force, hipStream_t s) {
    k_solve_only<<<1, 64, 0, s>>>(st, reduced, sp, do_solve, force);
}
