// ndt_pcl.hip — PCL compatibility mode: the objective, derivatives and optimiser that
// pcl::NormalDistributionsTransform runs behind ndt_.align() (src/slam/src/process.cc:94, parameters
// process.cc:373-376), as written down in SURVEY.md Appendix A.3-A.6 (PCL's own sources are not in this image:
// parity unpinned).
//
// k_pcl_pass: one source point per lane.  The point is transformed in float32 (PCL's Scalar = float), its cell
// follows the grid rule, the 27 cells around it are looked up in the voxel hash and a voxel is a neighbour when it
// is valid and its float32 centroid lies strictly within `resolution` of the point — the result set of PCL's radius
// search on the centroid kd-tree.  PCL then updates a 6-vector and a 6x6 matrix per neighbour; here the neighbours
// of a point are first summed into
//     G = sum w C x',   S = sum w C,   M = sum w (C x')(C x')^T,      w = d1 d2 exp(-d2/2 x'^T C x')
// and folded once per point through the point Jacobian J = [I | Ja] and the second derivatives of the transform:
//     gradient = J^T G,      Hessian = J^T (S - d2 M) J + [0 0; 0 G . H_E]      (Magnusson 2009 eq. 6.12, 6.13)
// which is the same sum.  Lanes -> wave -> block reduction as in the residual kernels; the block rows are added in
// a fixed order by k_pcl_reduce.  Bound: gather (27 probes x 32 B of keys + 40 B per voxel found); no MFMA.
//
// The Newton / More-Thuente driver runs on the host: every trial step needs the reduced score and gradient before
// the next one can be chosen, so each evaluation is one launch pair and one 256-byte read-back.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "ndt_internal.h"
#include "ndt_kernels.h"

#define PCL_THREADS 256
#define PCL_ACC_G 21
#define PCL_ACC_SCORE 27
#define PCL_ACC_NB 28
#define PCL_ACC_VOX 29      // valid voxels met in the 27 cells (for the algorithmic byte count)

struct PclPass {
    float M[12];          // row-major 3x4, float32 like PCL's final_transformation_
    double aj[8][3];      // angular_jacobian_ rows a..h
    double ah[15][3];     // angular_hessian_ rows a2..f3
    double d1, d2;
    KeyParams kp;
    float r2f;            // (float)(resolution^2)
};

// the three cells of a row once their table entries are known: radius test, Magnusson weight, neighbour sums
__device__ __forceinline__ void pcl_row_sums(const PclPass& pp, const float* qf, const int32_t* c, int32_t n1, int32_t n2, const uint32_t* fs,
                                             const Payload* pay, double* G, double* S, double* Mm, double& score, unsigned& nb, unsigned& nvox) {
    const double res = pp.kp.res_d;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int32_t n0 = c[0] + pr - 1;
        const bool vox = fs[pr] != SLOT_NONE && pay[pr].a.x != 0;
        if (!__any(vox)) continue;                       // no lane of the wave has a voxel in this cell
        nvox += vox ? 1u : 0u;
        const float o0 = __int_as_float(pay[pr].a.z), o1 = __int_as_float(pay[pr].a.w), o2 = pay[pr].b.x;
        // radius search on the centroid cloud: float32 centroid, float32 squared distance, strictly inside
        const float c0 = (float)((double)n0 * res + (double)o0), c1 = (float)((double)n1 * res + (double)o1), c2 = (float)((double)n2 * res + (double)o2);
        const float f0 = qf[0] - c0, f1 = qf[1] - c1, f2 = qf[2] - c2;
        const float dist2 = (f0 * f0 + f1 * f1) + f2 * f2;
        const bool nbr = vox && dist2 < pp.r2f;
        if (!__any(nbr)) continue;                       // ... or none within the radius: skip the float64 part (~60 instructions)
        nb += nbr ? 1u : 0u;
        const double i22 = __int_as_float(pay[pr].a.y), i00 = pay[pr].b.y, i01 = pay[pr].b.z, i02 = pay[pr].b.w, i11 = pay[pr].c.x, i12 = pay[pr].c.y;
        const double x0 = ((double)qf[0] - (double)n0 * res) - (double)o0;
        const double x1 = ((double)qf[1] - (double)n1 * res) - (double)o1;
        const double x2 = ((double)qf[2] - (double)n2 * res) - (double)o2;
        const double g0 = (i00 * x0 + i01 * x1) + i02 * x2;
        const double g1 = (i01 * x0 + i11 * x1) + i12 * x2;
        const double g2 = (i02 * x0 + i12 * x1) + i22 * x2;
        const double qv = (x0 * g0 + x1 * g1) + x2 * g2;
        const double ex = exp(-pp.d2 * qv / 2.0);
        const double e = pp.d2 * ex;
        const bool ok = nbr && !(e > 1.0 || e < 0.0 || e != e);         // updateDerivatives' validity test
        const double w = ok ? pp.d1 * e : 0.0;
        score += ok ? -pp.d1 * ex : 0.0;
        // selects, not multiplies by zero: the stand-in payload of a miss may give inf / NaN above
        const double a0 = ok ? g0 : 0.0, a1 = ok ? g1 : 0.0, a2 = ok ? g2 : 0.0;
        G[0] = fma(w, a0, G[0]); G[1] = fma(w, a1, G[1]); G[2] = fma(w, a2, G[2]);
        S[0] = fma(w, ok ? i00 : 0.0, S[0]); S[1] = fma(w, ok ? i01 : 0.0, S[1]); S[2] = fma(w, ok ? i02 : 0.0, S[2]);
        S[3] = fma(w, ok ? i11 : 0.0, S[3]); S[4] = fma(w, ok ? i12 : 0.0, S[4]); S[5] = fma(w, ok ? i22 : 0.0, S[5]);
        const double w0 = w * a0, w1 = w * a1, w2 = w * a2;
        Mm[0] = fma(w0, a0, Mm[0]); Mm[1] = fma(w0, a1, Mm[1]); Mm[2] = fma(w0, a2, Mm[2]);
        Mm[3] = fma(w1, a1, Mm[3]); Mm[4] = fma(w1, a2, Mm[4]); Mm[5] = fma(w2, a2, Mm[5]);
    }
}

// three cells (dx in -1..1) of the row (dy, dz): keys, resolve, payloads, neighbour sums.  Three probes at a time keep
// the kernel at two waves per SIMD; nine at a time need 386 registers.
__device__ __forceinline__ void pcl_row(const PclPass& pp, const float* qf, const int32_t* c, int dy, int dz, const int64_t* __restrict__ keys,
                                          const VoxSlot* __restrict__ slots, uint32_t mask, bool point_ok, double* G, double* S, double* Mm,
                                          double& score, unsigned& nb, unsigned& nvox) {
    int64_t want[3];
    uint32_t s0[3];
    Key2 ka[3], kb[3];
    bool live[3];
    const int32_t n1 = c[1] + dy, n2 = c[2] + dz;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int32_t n0 = c[0] + pr - 1;
        live[pr] = point_ok && cell_ok(n0) && cell_ok(n1) && cell_ok(n2);
        want[pr] = ndt_pack(n0, n1, n2);
        s0[pr] = live[pr] ? ndt_hash_cell(n0, n1, n2) & mask : 0u;
        const Key2* kp = (const Key2*)(keys + s0[pr]);      // keys[cap .. cap+3] mirror keys[0 .. 3]: no wrap
        ka[pr] = kp[0];
        kb[pr] = kp[1];
    }
    uint32_t fs[3], cur[3];
    unsigned pending = 0;
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) {
        const int64_t w = want[pr];
        const bool m0 = ka[pr].x == w, m1 = ka[pr].y == w, m2 = kb[pr].x == w, m3 = kb[pr].y == w;
        const bool e0 = ka[pr].x == NDT_KEY_EMPTY, e1 = ka[pr].y == NDT_KEY_EMPTY, e2 = kb[pr].x == NDT_KEY_EMPTY, e3 = kb[pr].y == NDT_KEY_EMPTY;
        const uint32_t off = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u;
        const bool found = m0 || (!e0 && (m1 || (!e1 && (m2 || (!e2 && m3)))));
        const bool more = !(m0 || e0 || m1 || e1 || m2 || e2 || m3 || e3);
        fs[pr] = (live[pr] && found) ? (s0[pr] + off) & mask : SLOT_NONE;
        pending |= (live[pr] && more) ? 1u << pr : 0u;
        cur[pr] = (s0[pr] + 4) & mask;
    }
    while (__any(pending != 0)) {      // chains longer than four entries (rare): all probes advance together
        int64_t kx[3];
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) kx[pr] = keys[cur[pr]];
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const bool pend = (pending >> pr) & 1u;
            const bool hit = pend && kx[pr] == want[pr], stop = pend && (hit || kx[pr] == NDT_KEY_EMPTY);
            fs[pr] = hit ? cur[pr] : fs[pr];
            pending &= stop ? ~(1u << pr) : ~0u;
            cur[pr] = (pend && !stop) ? (cur[pr] + 1) & mask : cur[pr];
        }
    }
    Payload pay[3];
#pragma unroll
    for (int pr = 0; pr < 3; ++pr) pay[pr] = load_payload(slots, fs[pr]);
    pcl_row_sums(pp, qf, c, n1, n2, fs, pay, G, S, Mm, score, nb, nvox);
}

// The same row through the neighbour links: only the middle cell (dx = 0) is hashed; if a voxel sits at dx = +-1 the
// middle cell is in the table too (as a voxel or as a ghost cell) and its entry holds their slots.
// A row is three dependent round trips (keys -> entry -> neighbour payloads); the key read of the NEXT row is issued
// before the current row is finished, so nine rows cost 19 trips instead of 27.
struct RowProbe { int64_t want; uint32_t s0; int32_t n1, n2; bool live; Key2 ka, kb; };
__device__ __forceinline__ RowProbe pcl_row_issue(const int32_t* c, int row, const int64_t* __restrict__ keys, uint32_t mask, bool point_ok) {
    RowProbe r;
    r.n1 = c[1] + row % 3 - 1; r.n2 = c[2] + row / 3 - 1;
    r.live = point_ok && cell_ok(c[0]) && cell_ok(r.n1) && cell_ok(r.n2);
    r.want = ndt_pack(c[0], r.n1, r.n2);
    r.s0 = r.live ? ndt_hash_cell(c[0], r.n1, r.n2) & mask : 0u;
    const Key2* kp = (const Key2*)(keys + r.s0);
    r.ka = kp[0]; r.kb = kp[1];
    return r;
}
__device__ __forceinline__ void pcl_row_finish(const PclPass& pp, const float* qf, const int32_t* c, const RowProbe& r, const int64_t* __restrict__ keys,
                                               const VoxSlot* __restrict__ slots, uint32_t mask, double* G, double* S, double* Mm,
                                               double& score, unsigned& nb, unsigned& nvox) {
    const int64_t want = r.want;
    const bool m0 = r.ka.x == want, m1 = r.ka.y == want, m2 = r.kb.x == want, m3 = r.kb.y == want;
    const bool e0 = r.ka.x == NDT_KEY_EMPTY, e1 = r.ka.y == NDT_KEY_EMPTY, e2 = r.kb.x == NDT_KEY_EMPTY, e3 = r.kb.y == NDT_KEY_EMPTY;
    const uint32_t off = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u;
    const bool found = r.live && (m0 || (!e0 && (m1 || (!e1 && (m2 || (!e2 && m3))))));
    uint32_t fsm = found ? (r.s0 + off) & mask : SLOT_NONE;
    bool pending = r.live && !(m0 || e0 || m1 || e1 || m2 || e2 || m3 || e3);
    uint32_t cur = (r.s0 + 4) & mask;
    while (__any(pending)) {           // chains longer than four entries (rare)
        const int64_t k = keys[cur];
        const bool hit = pending && k == want, stop = pending && (hit || k == NDT_KEY_EMPTY);
        fsm = hit ? cur : fsm;
        pending = pending && !stop;
        cur = pending ? (cur + 1) & mask : cur;
    }
    const int4* sp = (const int4*)(slots + (fsm != SLOT_NONE ? fsm : 0u));      // payload + links of the middle cell: one 64-byte line
    const int4 a = sp[0];
    const float4 b = *(const float4*)(sp + 1);
    const int4 c2 = sp[2];
    const bool have = fsm != SLOT_NONE;
    uint32_t fs[3] = {have ? (uint32_t)c2.z : SLOT_NONE, fsm, have ? (uint32_t)c2.w : SLOT_NONE};   // link[0] = -x, link[1] = +x
    Payload pay[3];
    pay[1].a = a; pay[1].b = b; pay[1].c = make_float2(__int_as_float(c2.x), __int_as_float(c2.y));
    pay[0] = load_payload(slots, fs[0]);
    pay[2] = load_payload(slots, fs[2]);
    pcl_row_sums(pp, qf, c, r.n1, r.n2, fs, pay, G, S, Mm, score, nb, nvox);
}

template <bool HESS, bool GRAD, bool LINKED>
__global__ __launch_bounds__(PCL_THREADS) void k_pcl_pass(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                                                          int64_t n, const int64_t* __restrict__ keys, const VoxSlot* __restrict__ slots,
                                                          uint32_t mask, PclPass pp, double* __restrict__ partials) {
    double acc[ACC_N];
#pragma unroll
    for (int j = 0; j < ACC_N; ++j) acc[j] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * PCL_THREADS;
    const int64_t first = (int64_t)blockIdx.x * PCL_THREADS;
    for (int64_t base = first; base < n; base += stride) {      // block-uniform trip count: the wave-level votes below see all lanes
        const int64_t i = base + threadIdx.x;
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < n) { x = xs[i]; y = ys[i]; z = zs[i]; }
        float qf[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) qf[r] = ((pp.M[4 * r] * x + pp.M[4 * r + 1] * y) + pp.M[4 * r + 2] * z) + pp.M[4 * r + 3];
        int32_t c[3] = {0, 0, 0};
        bool ok = i < n && isfinite(qf[0]) && isfinite(qf[1]) && isfinite(qf[2]);
        ok = ok && ndt_target_cell(pp.kp, qf[0], qf[1], qf[2], c);
        if (!ok) { c[0] = c[1] = c[2] = 0; qf[0] = qf[1] = qf[2] = 0.f; x = y = z = 0.f; }   // nothing non-finite may reach the sums
        double G[3] = {0, 0, 0}, S[6] = {0, 0, 0, 0, 0, 0}, Mm[6] = {0, 0, 0, 0, 0, 0}, score = 0.0;
        unsigned nb = 0, nvox = 0;                       // integer counters: no float64 chains through the 27 cells
#pragma unroll 1
        for (int row = 0; row < 9 && !LINKED; ++row) pcl_row(pp, qf, c, row % 3 - 1, row / 3 - 1, keys, slots, mask, ok, G, S, Mm, score, nb, nvox);
        if (LINKED) {
            RowProbe next = pcl_row_issue(c, 0, keys, mask, ok);
#pragma unroll 1
            for (int row = 0; row < 9; ++row) {
                const RowProbe cur = next;
                if (row < 8) next = pcl_row_issue(c, row + 1, keys, mask, ok);
                pcl_row_finish(pp, qf, c, cur, keys, slots, mask, G, S, Mm, score, nb, nvox);
            }
        }
        acc[PCL_ACC_NB] += (double)nb;
        acc[PCL_ACC_VOX] += (double)nvox;
        if (!__any(nb != 0u)) continue;
        if (GRAD) acc[PCL_ACC_SCORE] += score;
        // point Jacobian (computePointDerivatives): columns 3..5 of J_E from the ORIGINAL point
        const double px = x, py = y, pz = z;
        double xa[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xa[k] = (pp.aj[k][0] * px + pp.aj[k][1] * py) + pp.aj[k][2] * pz;
        const double Ja[3][3] = {{0.0, xa[2], xa[5]}, {xa[0], xa[3], xa[6]}, {xa[1], xa[4], xa[7]}};
        if (GRAD) {
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[PCL_ACC_G + r] += G[r];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[PCL_ACC_G + 3 + j] += fma(Ja[2][j], G[2], fma(Ja[1][j], G[1], Ja[0][j] * G[0]));
        }
        if (HESS) {
            double B[3][3];
            B[0][0] = fma(-pp.d2, Mm[0], S[0]); B[0][1] = fma(-pp.d2, Mm[1], S[1]); B[0][2] = fma(-pp.d2, Mm[2], S[2]);
            B[1][1] = fma(-pp.d2, Mm[3], S[3]); B[1][2] = fma(-pp.d2, Mm[4], S[4]); B[2][2] = fma(-pp.d2, Mm[5], S[5]);
            B[1][0] = B[0][1]; B[2][0] = B[0][2]; B[2][1] = B[1][2];
            double BJ[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) BJ[r][j] = fma(B[r][2], Ja[2][j], fma(B[r][1], Ja[1][j], B[r][0] * Ja[0][j]));
            double xh[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) xh[k] = (pp.ah[k][0] * px + pp.ah[k][1] * py) + pp.ah[k][2] * pz;
            // G . H_E(i,j): a, b, c have a zero first component
            const double hv[6] = {fma(G[2], xh[1], G[1] * xh[0]), fma(G[2], xh[3], G[1] * xh[2]), fma(G[2], xh[5], G[1] * xh[4]),
                                  fma(G[2], xh[8], fma(G[1], xh[7], G[0] * xh[6])), fma(G[2], xh[11], fma(G[1], xh[10], G[0] * xh[9])),
                                  fma(G[2], xh[14], fma(G[1], xh[13], G[0] * xh[12]))};
            // upper triangle, row-major, order [t, angles]
            acc[0] += B[0][0]; acc[1] += B[0][1]; acc[2] += B[0][2]; acc[3] += BJ[0][0]; acc[4] += BJ[0][1]; acc[5] += BJ[0][2];
            acc[6] += B[1][1]; acc[7] += B[1][2]; acc[8] += BJ[1][0]; acc[9] += BJ[1][1]; acc[10] += BJ[1][2];
            acc[11] += B[2][2]; acc[12] += BJ[2][0]; acc[13] += BJ[2][1]; acc[14] += BJ[2][2];
            int t = 15, hi = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = a; b < 3; ++b, ++t, ++hi)
                    acc[t] += fma(Ja[2][a], BJ[2][b], fma(Ja[1][a], BJ[1][b], Ja[0][a] * BJ[0][b])) + hv[hi];
        }
    }
    block_store_row<PCL_THREADS>(acc, partials + (size_t)blockIdx.x * ACC_N);
}

// fixed-order sum of the block rows: thread (g, j) adds rows g, g+32, ... of column j; then a tree over g
__global__ __launch_bounds__(1024) void k_pcl_reduce(const double* __restrict__ partials, int n_rows, double* __restrict__ out) {
    __shared__ double sm[32][ACC_N + 1];
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
    double s = 0.0;
    for (int r = g; r < n_rows; r += 32) s += partials[(size_t)r * ACC_N + j];
    sm[g][j] = s;
    __syncthreads();
    for (int h = 16; h >= 1; h >>= 1) {
        if (g < h) sm[g][j] += sm[g + h][j];
        __syncthreads();
    }
    if (g == 0) out[j] = sm[0][j];
}

// ------------------------------------------------------------------ host side

static void launch_pass(const PclEngine& e, const PclPass& pp, int grid, bool with_gradient, bool with_hessian) {
    const int64_t n = (int64_t)e.n;
#define PCL_LAUNCH(H, G, L) k_pcl_pass<H, G, L><<<grid, PCL_THREADS, 0, e.stream>>>(e.sx, e.sy, e.sz, n, e.keys, e.slots, e.mask, pp, e.partials)
    if (e.use_links) {
        if (with_hessian && with_gradient) PCL_LAUNCH(true, true, true); else if (with_gradient) PCL_LAUNCH(false, true, true); else PCL_LAUNCH(true, false, true);
    } else {
        if (with_hessian && with_gradient) PCL_LAUNCH(true, true, false); else if (with_gradient) PCL_LAUNCH(false, true, false); else PCL_LAUNCH(true, false, false);
    }
#undef PCL_LAUNCH
}

namespace {

void gauss_constants(double res, double outlier_ratio, double* d1, double* d2) {      // eq. 6.8, computeTransformation()
    const double c1 = 10.0 * (1.0 - outlier_ratio);
    const double c2 = outlier_ratio / std::pow(res, 3);
    const double d3 = -std::log(c2);
    *d1 = -std::log(c1 + c2) - d3;
    *d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / *d1);
}

void rt_from_vector(const double p[6], double R[3][3], double t[3]) {      // convertTransform: R = Rx Ry Rz
    const double cx = std::cos(p[3]), sx = std::sin(p[3]), cy = std::cos(p[4]), sy = std::sin(p[4]), cz = std::cos(p[5]), sz = std::sin(p[5]);
    R[0][0] = cy * cz;                  R[0][1] = -(cy * sz);               R[0][2] = sy;
    R[1][0] = cx * sz + (sx * sy) * cz; R[1][1] = cx * cz - (sx * sy) * sz; R[1][2] = -(sx * cy);
    R[2][0] = sx * sz - (cx * sy) * cz; R[2][1] = sx * cz + (cx * sy) * sz; R[2][2] = cx * cy;
    t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
}

void cos_sin(double a, double* c, double* s) {      // computeAngleDerivatives' small-angle shortcut
    if (std::fabs(a) < 10e-5) { *c = 1.0; *s = 0.0; } else { *c = std::cos(a); *s = std::sin(a); }
}
void angle_jacobian(const double p[6], double aj[8][3]) {
    double cx, cy, cz, sx, sy, sz;
    cos_sin(p[3], &cx, &sx); cos_sin(p[4], &cy, &sy); cos_sin(p[5], &cz, &sz);
    const double v[8][3] = {{(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},
                            {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},
                            {(-sy * cz), sy * sz, cy},
                            {sx * cy * cz, (-sx * cy * sz), sx * sy},
                            {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},
                            {(-cy * sz), (-cy * cz), 0},
                            {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},
                            {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};
    std::memcpy(aj, v, sizeof v);
}
void angle_hessian(const double p[6], int d1_as_pcl, double ah[15][3]) {
    double cx, cy, cz, sx, sy, sz;
    cos_sin(p[3], &cx, &sx); cos_sin(p[4], &cy, &sy); cos_sin(p[5], &cz, &sz);
    const double v[15][3] = {{(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},
                             {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},
                             {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},
                             {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},
                             {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},
                             {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},
                             {(-cy * cz), (cy * sz), d1_as_pcl ? (sy) : (-sy)},      // d1: PCL has +sy, eq. 6.21 has -sy
                             {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},
                             {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},
                             {(sy * sz), (sy * cz), 0},
                             {(-sx * cy * sz), (-sx * cy * cz), 0},
                             {(cx * cy * sz), (cx * cy * cz), 0},
                             {(-cy * cz), (cy * sz), 0},
                             {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},
                             {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};
    std::memcpy(ah, v, sizeof v);
}

// cyclic Jacobi eigen-decomposition of a symmetric 6x6 (columns of V = eigenvectors)
void sym6_eig(const double Ain[36], double lam[6], double V[36]) {
    double A[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { A[i][j] = 0.5 * (Ain[i * 6 + j] + Ain[j * 6 + i]); V[i * 6 + j] = (i == j); }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < 6; ++i) { dia += A[i][i] * A[i][i]; for (int j = i + 1; j < 6; ++j) off += A[i][j] * A[i][j]; }
        if (off <= 1e-32 * dia || off == 0.0) break;
        for (int pp = 0; pp < 5; ++pp)
            for (int qq = pp + 1; qq < 6; ++qq) {
                if (A[pp][qq] == 0.0) continue;
                const double th = (A[qq][qq] - A[pp][pp]) / (2.0 * A[pp][qq]);
                const double tt = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 6; ++k) { const double a = A[k][pp], b = A[k][qq]; A[k][pp] = c * a - s * b; A[k][qq] = s * a + c * b; }
                for (int k = 0; k < 6; ++k) { const double a = A[pp][k], b = A[qq][k]; A[pp][k] = c * a - s * b; A[qq][k] = s * a + c * b; }
                for (int k = 0; k < 6; ++k) { const double a = V[k * 6 + pp], b = V[k * 6 + qq]; V[k * 6 + pp] = c * a - s * b; V[k * 6 + qq] = s * a + c * b; }
            }
    }
    for (int i = 0; i < 6; ++i) lam[i] = A[i][i];
}
// JacobiSVD(H).solve(b) for a symmetric H: pseudo-inverse over the eigenvalues above Eigen's default rank threshold
void svd_solve(const double H[36], const double b[6], double x[6]) {
    double lam[6], V[36], mx = 0.0;
    sym6_eig(H, lam, V);
    for (int i = 0; i < 6; ++i) mx = std::fmax(mx, std::fabs(lam[i]));
    const double thr = 6.0 * DBL_EPSILON * mx;
    std::memset(x, 0, 6 * sizeof(double));
    for (int i = 0; i < 6; ++i) {
        if (!(std::fabs(lam[i]) > thr)) continue;
        double d = 0.0;
        for (int k = 0; k < 6; ++k) d += V[k * 6 + i] * b[k];
        d /= lam[i];
        for (int k = 0; k < 6; ++k) x[k] += d * V[k * 6 + i];
    }
}

// More-Thuente pieces (ndt.hpp: auxilaryFunction_PsiMT, auxilaryFunction_dPsiMT, updateIntervalMT, trialValueSelectionMT)
double mt_psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
double mt_dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }
bool mt_update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t, double f_t, double g_t) {
    if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
    if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
    return true;
}
double mt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
    if (a_t == a_l && a_t == a_u) return a_t;
    int cond;
    if (a_t == a_l) cond = 4;
    else if (f_t > f_l) cond = 1;
    else if (g_t * g_l < 0) cond = 2;
    else if (std::fabs(g_t) <= std::fabs(g_l)) cond = 3;
    else cond = 4;
    if (cond == 1) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        const double w = std::sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        if (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) return a_c;
        return 0.5 * (a_q + a_c);
    }
    if (cond == 2) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        const double w = std::sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        if (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) return a_c;
        return a_s;
    }
    if (cond == 3) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        const double w = std::sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double a_n = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
        if (a_t > a_l) return std::fmin(a_t + 0.66 * (a_u - a_t), a_n);
        return std::fmax(a_t + 0.66 * (a_u - a_t), a_n);
    }
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    const double w = std::sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

struct Driver {
    const PclEngine* e;
    const ndt_pcl_params* prm;
    double d1, d2;
    double aj[8][3], ah[15][3];      // PCL's angular_jacobian_ / angular_hessian_ members
    int evals = 0;
    int64_t n_nb = 0;
    std::string* err;

    // one pass over the source on the device; fills what the flags ask for
    int pass(const double p[6], bool with_gradient, bool with_hessian, double* score, double g[6], double H[36]) {
        PclPass pp;
        double R[3][3], t[3];
        rt_from_vector(p, R, t);
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pp.M[4 * r + c] = (float)R[r][c]; pp.M[4 * r + 3] = (float)t[r]; }
        std::memcpy(pp.aj, aj, sizeof aj); std::memcpy(pp.ah, ah, sizeof ah);
        pp.d1 = d1; pp.d2 = d2; pp.kp = e->kp;
        pp.r2f = (float)(e->kp.res_d * e->kp.res_d);
        const int grid = pcl_grid(e->n, e->n_cu);
        launch_pass(*e, pp, grid, with_gradient, with_hessian);
        k_pcl_reduce<<<1, 1024, 0, e->stream>>>(e->partials, grid, e->d_row);
        hipError_t rc = hipGetLastError();
        if (rc == hipSuccess) rc = hipMemcpyAsync(e->h_row, e->d_row, ACC_N * sizeof(double), hipMemcpyDeviceToHost, e->stream);
        if (rc == hipSuccess) rc = hipStreamSynchronize(e->stream);
        if (rc != hipSuccess) { *err = std::string("PCL-mode pass: ") + hipGetErrorString(rc); return NDT_ERR_HIP; }
        ++evals;
        const double* a = e->h_row;
        if (with_gradient) {
            if (score) *score = a[PCL_ACC_SCORE];
            for (int k = 0; k < 6; ++k) g[k] = a[PCL_ACC_G + k];
            n_nb = (int64_t)a[PCL_ACC_NB];
        }
        if (with_hessian) {
            int t2 = 0;
            for (int r = 0; r < 6; ++r)
                for (int c = r; c < 6; ++c, ++t2) { H[r * 6 + c] = a[t2]; H[c * 6 + r] = a[t2]; }
        }
        return NDT_OK;
    }
    // computeDerivatives(): refreshes the angular Jacobian, and the angular Hessian only when the Hessian is asked for
    int derivs(const double p[6], bool with_hessian, double* score, double g[6], double H[36]) {
        angle_jacobian(p, aj);
        if (with_hessian) angle_hessian(p, prm->hessian_d1_as_pcl, ah);
        return pass(p, true, with_hessian, score, g, H);
    }
};

}  // namespace

int pcl_grid(size_t n, int n_cu) {
    // one point per lane up to one round of resident blocks (two 256-thread blocks per CU at 256 VGPRs), then the lanes loop
    size_t g = (n + PCL_THREADS - 1) / PCL_THREADS;
    const size_t cap = (size_t)2 * (size_t)(n_cu > 0 ? n_cu : 256);
    return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

void pcl_matrix_from_vector(const double p[6], double T[16]) {
    double R[3][3], t[3];
    rt_from_vector(p, R, t);
    std::memset(T, 0, 16 * sizeof(double));
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[c * 4 + r] = R[r][c]; T[12 + r] = t[r]; }
    T[15] = 1.0;
}
// Eigen's MatrixBase::eulerAngles(0,1,2): first angle in [0, pi]
void pcl_vector_from_matrix(const double T[16], double p[6]) {
    double m[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m[r][c] = T[c * 4 + r];
    const double PI = 3.14159265358979323846;
    double r0 = std::atan2(m[1][2], m[2][2]);
    const double c2 = std::sqrt(m[0][0] * m[0][0] + m[0][1] * m[0][1]);
    double r1;
    if (r0 > 0.0) { r0 -= PI; r1 = std::atan2(-m[0][2], -c2); }
    else r1 = std::atan2(-m[0][2], c2);
    const double s1 = std::sin(r0), c1 = std::cos(r0);
    const double r2 = std::atan2(s1 * m[2][0] - c1 * m[1][0], c1 * m[1][1] - s1 * m[2][1]);
    p[0] = T[12]; p[1] = T[13]; p[2] = T[14];
    p[3] = -r0; p[4] = -r1; p[5] = -r2;
}

int pcl_eval(const PclEngine& e, const ndt_pcl_params& prm, const double p[6], double* score, double g[6], double* H, int64_t* n_nb, std::string& err) {
    Driver d;
    d.e = &e; d.prm = &prm; d.err = &err;
    gauss_constants(e.kp.res_d, prm.outlier_ratio, &d.d1, &d.d2);
    double sc = 0.0, gg[6], HH[36];
    int rc = d.derivs(p, H != nullptr, &sc, gg, HH);
    if (rc != NDT_OK) return rc;
    if (score) *score = sc;
    if (g) std::memcpy(g, gg, sizeof gg);
    if (H) std::memcpy(H, HH, sizeof HH);
    if (n_nb) *n_nb = d.n_nb;
    return NDT_OK;
}

// computeTransformation + computeStepLengthMT
int pcl_align(const PclEngine& e, const ndt_pcl_params& prm, int max_iterations, double trans_eps, const double guess[16], ndt_pcl_result* out,
              std::string& err) {
    std::memset(out, 0, sizeof *out);
    Driver d;
    d.e = &e; d.prm = &prm; d.err = &err;
    gauss_constants(e.kp.res_d, prm.outlier_ratio, &d.d1, &d.d2);
    double p[6], g[6], H[36], score = 0.0;
    pcl_vector_from_matrix(guess, p);
    int rc = d.derivs(p, true, &score, g, H);
    if (rc != NDT_OK) return out->status = rc;
    bool converged = false;
    int iters = 0;
    while (!converged) {
        double ng[6], delta[6];
        for (int k = 0; k < 6; ++k) ng[k] = -g[k];
        svd_solve(H, ng, delta);
        double dn = 0.0;
        for (int k = 0; k < 6; ++k) dn += delta[k] * delta[k];
        dn = std::sqrt(dn);
        if (dn == 0 || dn != dn) { converged = dn == 0; break; }
        for (int k = 0; k < 6; ++k) delta[k] /= dn;
        // ---- computeStepLengthMT(p, delta, dn, step_size, eps / 2, score, g, H)
        const double step_max = prm.step_size, step_min = trans_eps / 2;
        double a_t;
        {
            const double phi_0 = -score;
            double d_phi_0 = 0.0;
            for (int k = 0; k < 6; ++k) d_phi_0 += g[k] * delta[k];
            d_phi_0 = -d_phi_0;
            bool zero_step = false;
            if (d_phi_0 >= 0) {
                if (d_phi_0 == 0) zero_step = true;
                else { d_phi_0 *= -1; for (int k = 0; k < 6; ++k) delta[k] *= -1; }
            }
            if (zero_step) a_t = 0;
            else {
                const int max_step_iterations = 10;
                int step_iterations = 0;
                const double mu = 1.e-4, nu = 0.9;
                double a_l = 0, a_u = 0;
                double f_l = mt_psi(a_l, phi_0, phi_0, d_phi_0, mu), g_l = mt_dpsi(d_phi_0, d_phi_0, mu);
                double f_u = mt_psi(a_u, phi_0, phi_0, d_phi_0, mu), g_u = mt_dpsi(d_phi_0, d_phi_0, mu);
                bool interval_converged = (step_max - step_min) < 0, open_interval = true;
                a_t = dn;
                a_t = std::fmin(a_t, step_max);
                a_t = std::fmax(a_t, step_min);
                double x_t[6];
                for (int k = 0; k < 6; ++k) x_t[k] = p[k] + delta[k] * a_t;
                if ((rc = d.derivs(x_t, true, &score, g, H)) != NDT_OK) return out->status = rc;
                double phi_t = -score, d_phi_t = 0.0;
                for (int k = 0; k < 6; ++k) d_phi_t += g[k] * delta[k];
                d_phi_t = -d_phi_t;
                double psi_t = mt_psi(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = mt_dpsi(d_phi_t, d_phi_0, mu);
                while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
                    if (open_interval) a_t = mt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
                    else a_t = mt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
                    a_t = std::fmin(a_t, step_max);
                    a_t = std::fmax(a_t, step_min);
                    for (int k = 0; k < 6; ++k) x_t[k] = p[k] + delta[k] * a_t;
                    if ((rc = d.derivs(x_t, false, &score, g, H)) != NDT_OK) return out->status = rc;
                    phi_t = -score; d_phi_t = 0.0;
                    for (int k = 0; k < 6; ++k) d_phi_t += g[k] * delta[k];
                    d_phi_t = -d_phi_t;
                    psi_t = mt_psi(a_t, phi_t, phi_0, d_phi_0, mu);
                    d_psi_t = mt_dpsi(d_phi_t, d_phi_0, mu);
                    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
                        open_interval = false;
                        f_l += phi_0 - mu * d_phi_0 * a_l; g_l += mu * d_phi_0;
                        f_u += phi_0 - mu * d_phi_0 * a_u; g_u += mu * d_phi_0;
                    }
                    if (open_interval) interval_converged = mt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
                    else interval_converged = mt_update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
                    step_iterations++;
                }
                // computeHessian: angular_jacobian_ of x_t, angular_hessian_ still that of the first trial (as PCL leaves it)
                if (step_iterations && (rc = d.pass(x_t, false, true, nullptr, nullptr, H)) != NDT_OK) return out->status = rc;
            }
        }
        for (int k = 0; k < 6; ++k) delta[k] *= a_t;
        double Rd[3][3], td[3];
        rt_from_vector(delta, Rd, td);
        for (int k = 0; k < 6; ++k) p[k] += delta[k];
        const double cos_angle = 0.5 * ((Rd[0][0] + Rd[1][1] + Rd[2][2]) - 1.0);
        const double tsq = (td[0] * td[0] + td[1] * td[1]) + td[2] * td[2];
        ++iters;
        const double te = trans_eps, re = prm.rotation_eps;
        if (iters >= max_iterations || ((te > 0 && tsq <= te) && (re > 0 && cos_angle >= re)) || ((te <= 0) && (re > 0 && cos_angle >= re)) ||
            ((te > 0 && tsq <= te) && (re <= 0)))
            converged = true;
    }
    out->iterations = iters; out->converged = converged ? 1 : 0; out->evaluations = d.evals;
    out->score = score; out->trans_probability = e.n ? score / (double)e.n : 0.0; out->n_neighbours = d.n_nb;
    std::memcpy(out->p, p, sizeof p); std::memcpy(out->gradient, g, sizeof g); std::memcpy(out->hessian, H, sizeof H);
    pcl_matrix_from_vector(p, out->T);
    out->status = NDT_OK;
    return NDT_OK;
}

// Measurement: mean duration of the derivative pass alone (HIP events on the engine's stream around `reps`
// back-to-back launches), and what it met per source point.
int pcl_time_pass(const PclEngine& e, const ndt_pcl_params& prm, const double p[6], bool with_hessian, int reps, double* us_per_launch,
                  double* voxels_per_point, double* neighbours_per_point, std::string& err) {
    Driver d;
    d.e = &e; d.prm = &prm; d.err = &err;
    gauss_constants(e.kp.res_d, prm.outlier_ratio, &d.d1, &d.d2);
    double sc, g[6], H[36];
    int rc = d.derivs(p, with_hessian, &sc, g, H);       // warm-up + the per-point counts
    if (rc != NDT_OK) return rc;
    if (voxels_per_point) *voxels_per_point = e.n ? e.h_row[PCL_ACC_VOX] / (double)e.n : 0.0;
    if (neighbours_per_point) *neighbours_per_point = e.n ? e.h_row[PCL_ACC_NB] / (double)e.n : 0.0;
    PclPass pp;
    double R[3][3], t[3];
    rt_from_vector(p, R, t);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) pp.M[4 * r + c] = (float)R[r][c]; pp.M[4 * r + 3] = (float)t[r]; }
    std::memcpy(pp.aj, d.aj, sizeof d.aj); std::memcpy(pp.ah, d.ah, sizeof d.ah);
    pp.d1 = d.d1; pp.d2 = d.d2; pp.kp = e.kp; pp.r2f = (float)(e.kp.res_d * e.kp.res_d);
    hipEvent_t a, b;
    hipError_t hr = hipEventCreate(&a);
    if (hr == hipSuccess) hr = hipEventCreate(&b);
    if (hr != hipSuccess) { err = hipGetErrorString(hr); return NDT_ERR_HIP; }
    const int grid = pcl_grid(e.n, e.n_cu);
    (void)hipEventRecord(a, e.stream);
    for (int i = 0; i < reps; ++i) {
        launch_pass(e, pp, grid, true, with_hessian);
    }
    (void)hipEventRecord(b, e.stream);
    hr = hipEventSynchronize(b);
    float ms = 0.f;
    if (hr == hipSuccess) hr = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (hr != hipSuccess) { err = hipGetErrorString(hr); return NDT_ERR_HIP; }
    if (us_per_launch) *us_per_launch = 1000.0 * (double)ms / (double)reps;
    return NDT_OK;
}
