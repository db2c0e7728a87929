// ndt_residual.hip — the per-iteration residual / accumulation pass and the Gauss-Newton update
// (replaces the body of ndt_.align(), src/slam/src/process.cc:94, in the
// Gauss-Newton / NEARBY1-NEARBY7 form called at src/slam/reference/lio_preinteg.cc:116-121).
//
// k_residual: one source point per lane.  q' = R q + t in float64, cell = rule(q'/res), K hash probes
// (centre + 6 face neighbours), Mahalanobis residual against the 40-byte voxel payload, chi^2 gate.  The
// per-point sums S = sum Sigma^-1 and g = sum Sigma^-1 e are folded once per point through the 3x6
// Jacobian J = [-R hat(q) | I] into 21 + 6 accumulators per lane.
//   memory schedule : all K key reads (two 16-byte reads = slots s .. s+3 of the dense key array) are issued before
//                     any is used; then all payload reads of the slots that matched; chains longer than four slots
//                     are finished for all K probes together in one wave-uniform loop.  With neighbour links (the
//                     default) only the point's own cell is hashed: its 64-byte entry carries the six neighbours' slots.
//   reduction       : lanes -> wave by a butterfly reduce-scatter (32 double exchanges instead of 30 x 6),
//                     wave -> block through LDS, one 32-double row per block in HBM; every order is fixed.
// k_reduce_solve adds the rows in a fixed order and takes the Gauss-Newton step on the device, so an iteration
// never returns to the host.
// Bound: HBM/L2 gather (12 + 8K + 40h algorithmic bytes per point); no MFMA.
#include "ndt_internal.h"

#include <cstddef>
#include "ndt_kernels.h"

#define RES_THREADS 256    // batched launches: several blocks per CU overlap their tails
#define RES1_THREADS 512   // single-problem launches: one block per CU, half as many rows to reduce

struct Pose { double R[9]; double t[3]; };

// Mahalanobis residual of the point against the K voxels in pay[], chi^2 gate, select-free accumulation of
// S = sum Sigma^-1, g = sum Sigma^-1 e, chi2, n_eff, hits.  Same operation order as the oracle.
template <int K>
__device__ __forceinline__ unsigned gate_and_sum(const Payload* pay, const uint32_t* fs, const double* q, const int32_t* c,
                                                 const EvalParams& ep, double* S, double* g, double* acc) {
    unsigned hitmask = 0, voxmask = 0;
#pragma unroll
    for (int pr = 0; pr < K; ++pr) {
        const int dx = pr == 1 ? -1 : pr == 2 ? 1 : 0, dy = pr == 3 ? -1 : pr == 4 ? 1 : 0, dz = pr == 5 ? -1 : pr == 6 ? 1 : 0;
        const bool vox = fs[pr] != SLOT_NONE && pay[pr].a.x != 0;      // a valid voxel was found
        const double i22 = __int_as_float(pay[pr].a.y), i00 = pay[pr].b.y, i01 = pay[pr].b.z, i02 = pay[pr].b.w, i11 = pay[pr].c.x,
                     i12 = pay[pr].c.y;
        const double e0 = (q[0] - (double)(c[0] + dx) * ep.kp.res_d) - (double)__int_as_float(pay[pr].a.z);
        const double e1 = (q[1] - (double)(c[1] + dy) * ep.kp.res_d) - (double)__int_as_float(pay[pr].a.w);
        const double e2 = (q[2] - (double)(c[2] + dz) * ep.kp.res_d) - (double)pay[pr].b.x;
        // fused multiply-adds in this fixed order on both sides (oracle: C fma()): same bits, 8 instructions fewer per voxel
        const double g0 = fma(i02, e2, fma(i01, e1, i00 * e0));
        const double g1 = fma(i12, e2, fma(i11, e1, i01 * e0));
        const double g2 = fma(i22, e2, fma(i12, e1, i02 * e0));
        const double chi2 = fma(e2, g2, fma(e1, g1, e0 * g0));
        const bool ok = vox && chi2 <= ep.outlier_th;               // false for NaN
        hitmask |= ok ? 1u << pr : 0u;
        voxmask |= vox ? 1u << pr : 0u;
        // w is exactly 0 or 1 and every operand is finite (payloads of valid voxels are finite, slot 0 stands in for
        // misses), so fma(w, v, s) equals the plain s + v when the residual is accepted and s otherwise
        const double w = ok ? 1.0 : 0.0;
        S[0] = fma(w, i00, S[0]); S[1] = fma(w, i01, S[1]); S[2] = fma(w, i02, S[2]);
        S[3] = fma(w, i11, S[3]); S[4] = fma(w, i12, S[4]); S[5] = fma(w, i22, S[5]);
        g[0] = fma(w, g0, g[0]); g[1] = fma(w, g1, g[1]); g[2] = fma(w, g2, g[2]);
        acc[ACC_CHI2] = fma(w, chi2, acc[ACC_CHI2]);
    }
    acc[ACC_HITS] += (double)__popc(voxmask);                        // the counters: small integers, exact either way
    acc[ACC_NEFF] += (double)__popc(hitmask);
    return hitmask;
}

// Folds one point's S = sum Sigma^-1 and g = sum Sigma^-1 e into the 21 + 6 accumulators, in the WORLD frame.
// With u = R x (the point's lever arm, q - t) and K = hat(u):  J = [-R hat(x) | I] = [-K R | I], so
//   H = [R^T M R, R^T N; N^T R, S]  with  M = K^T S K = -(K S) K,  N = K S,      b = [-R^T (u x g); -g].
// The lanes accumulate the pose-independent sums M (6), N (9), S (6), u x g (3), -g (3) in the slots of the H / b entries
// they turn into; raw_to_Hb applies R once per reduced row (in the reduce / solve kernels).  57 float64 instructions per
// point instead of the 90 that forming A = -R hat(x), S A and A^T S A took.
// (explicit fma: this part is compared at the 1e-5 tolerance, not bit for bit)
__device__ __forceinline__ void fold_point(const Pose& P, const double* q, unsigned hitmask, const double* S, const double* g, double* acc) {
    if (!__any(hitmask != 0)) return;
    const double u0 = q[0] - P.t[0], u1 = q[1] - P.t[1], u2 = q[2] - P.t[2];
    const double Sm[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    double n[3][3];                                   // n = K S
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        n[0][c] = fma(u1, Sm[2][c], -(u2 * Sm[1][c]));
        n[1][c] = fma(u2, Sm[0][c], -(u0 * Sm[2][c]));
        n[2][c] = fma(u0, Sm[1][c], -(u1 * Sm[0][c]));
    }
    // M = -(n K), upper triangle;  (n K)_i0 = n_i1 u2 - n_i2 u1,  (n K)_i1 = n_i2 u0 - n_i0 u2,  (n K)_i2 = n_i0 u1 - n_i1 u0
    acc[0] = fma(n[0][2], u1, fma(-n[0][1], u2, acc[0]));
    acc[1] = fma(n[0][0], u2, fma(-n[0][2], u0, acc[1]));
    acc[2] = fma(n[0][1], u0, fma(-n[0][0], u1, acc[2]));
    acc[6] = fma(n[1][0], u2, fma(-n[1][2], u0, acc[6]));
    acc[7] = fma(n[1][1], u0, fma(-n[1][0], u1, acc[7]));
    acc[11] = fma(n[2][1], u0, fma(-n[2][0], u1, acc[11]));
    // N
    acc[3] += n[0][0]; acc[4] += n[0][1]; acc[5] += n[0][2];
    acc[8] += n[1][0]; acc[9] += n[1][1]; acc[10] += n[1][2];
    acc[12] += n[2][0]; acc[13] += n[2][1]; acc[14] += n[2][2];
    // S
    acc[15] += Sm[0][0]; acc[16] += Sm[0][1]; acc[17] += Sm[0][2]; acc[18] += Sm[1][1]; acc[19] += Sm[1][2]; acc[20] += Sm[2][2];
    // u x g and -g
    acc[21] = fma(u1, g[2], fma(-u2, g[1], acc[21]));
    acc[22] = fma(u2, g[0], fma(-u0, g[2], acc[22]));
    acc[23] = fma(u0, g[1], fma(-u1, g[0], acc[23]));
    acc[24] -= g[0]; acc[25] -= g[1]; acc[26] -= g[2];
}

// Accumulates one source point into acc[0..29].  Returns the hit mask; *cell_key gets the centre key (or EMPTY).
template <int K>
__device__ __forceinline__ unsigned point_terms(const Pose& P, float xf, float yf, float zf, const int64_t* __restrict__ keys,
                                                const VoxSlot* __restrict__ slots, uint32_t mask, const EvalParams& ep,
                                                double* acc, int64_t* cell_key) {
    double q[3];
    ndt_xform(P.R, P.t, xf, yf, zf, q);
    int32_t c[3];
    if (!ndt_source_cell(ep.kp, q, c)) { *cell_key = NDT_KEY_EMPTY; return 0; }   // non-finite / out of key range
    const int64_t ckey = ndt_pack(c[0], c[1], c[2]);
    *cell_key = ckey;
    const uint32_t hx = (uint32_t)c[0] * NDT_HP0, hy = (uint32_t)c[1] * NDT_HP1, hz = (uint32_t)c[2] * NDT_HP2;

    // ---- round 1: four keys per probe (two 16-byte reads), all probes in flight together
    int64_t want[K];
    uint32_t s0[K];
    Key2 ka[K], kb[K];
    bool live[K];
#pragma unroll
    for (int pr = 0; pr < K; ++pr) {
        const int dx = pr == 1 ? -1 : pr == 2 ? 1 : 0, dy = pr == 3 ? -1 : pr == 4 ? 1 : 0, dz = pr == 5 ? -1 : pr == 6 ? 1 : 0;
        live[pr] = cell_ok(c[0] + dx) && cell_ok(c[1] + dy) && cell_ok(c[2] + dz);
        want[pr] = ckey + ((int64_t)dx << 42) + ((int64_t)dy << 21) + (int64_t)dz;
        const uint32_t h = ndt_hash_fin((hx + (uint32_t)dx * NDT_HP0) ^ (hy + (uint32_t)dy * NDT_HP1) ^ (hz + (uint32_t)dz * NDT_HP2));
        s0[pr] = h & mask;
        const Key2* kp = (const Key2*)(keys + s0[pr]);   // keys[cap .. cap+3] mirror keys[0 .. 3]: no wrap
        ka[pr] = kp[0];
        kb[pr] = kp[1];
    }
    // resolve, branch-free: fs = slot that holds the wanted key (SLOT_NONE if absent), pending = chain goes on
    uint32_t fs[K];
    uint32_t cur[K];
    unsigned pending = 0;
#pragma unroll
    for (int pr = 0; pr < K; ++pr) {
        const int64_t w = want[pr];
        const bool m0 = ka[pr].x == w, m1 = ka[pr].y == w, m2 = kb[pr].x == w, m3 = kb[pr].y == w;
        const bool e0 = ka[pr].x == NDT_KEY_EMPTY, e1 = ka[pr].y == NDT_KEY_EMPTY, e2 = kb[pr].x == NDT_KEY_EMPTY,
                   e3 = kb[pr].y == NDT_KEY_EMPTY;
        // first position that ends the chain (match or empty); the map never holds a key twice
        const uint32_t off = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u;
        const bool found = m0 || (!e0 && (m1 || (!e1 && (m2 || (!e2 && m3)))));
        const bool more = !(m0 || e0 || m1 || e1 || m2 || e2 || m3 || e3);
        fs[pr] = (live[pr] && found) ? (s0[pr] + off) & mask : SLOT_NONE;
        pending |= (live[pr] && more) ? 1u << pr : 0u;
        cur[pr] = (s0[pr] + 4) & mask;
    }
    // ---- probe chains longer than four slots (rare): all probes advance together, one round trip per level
    while (__any(pending != 0)) {
        int64_t kx[K];
#pragma unroll
        for (int pr = 0; pr < K; ++pr) kx[pr] = keys[cur[pr]];
#pragma unroll
        for (int pr = 0; pr < K; ++pr) {
            const bool pend = (pending >> pr) & 1u;
            const bool hit = pend && kx[pr] == want[pr], stop = pend && (hit || kx[pr] == NDT_KEY_EMPTY);
            fs[pr] = hit ? cur[pr] : fs[pr];
            pending &= stop ? ~(1u << pr) : ~0u;
            cur[pr] = (pend && !stop) ? (cur[pr] + 1) & mask : cur[pr];
        }
    }
    // ---- round 2: payload of every matched slot (slot 0 stands in for a miss), all in flight together
    Payload pay[K];
#pragma unroll
    for (int pr = 0; pr < K; ++pr) pay[pr] = load_payload(slots, fs[pr]);
    double S[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    const unsigned hitmask = gate_and_sum<K>(pay, fs, q, c, ep, S, g, acc);
    fold_point(P, q, hitmask, S, g, acc);
    return hitmask;
}

// NEARBY7 through the neighbour links: only the point's own cell is hashed; its slot holds the slots of the six
// face neighbours (ghost cells make sure a cell next to a voxel is always in the table).  Same probe order and the
// same arithmetic as point_terms<7>, so both give identical bits.
// first half: transform, cell, hash, and the two 16-byte key reads of the point's own cell (issued, not yet used)
struct CentreProbe { uint32_t s0; Key2 ka, kb; double q[3]; int32_t c[3]; bool ok; };
__device__ __forceinline__ CentreProbe centre_probe(const Pose& P, float xf, float yf, float zf, const int64_t* __restrict__ keys, uint32_t mask,
                                                    const EvalParams& ep) {
    CentreProbe pr;
    ndt_xform(P.R, P.t, xf, yf, zf, pr.q);
    pr.ok = ndt_source_cell(ep.kp, pr.q, pr.c);
    pr.s0 = pr.ok ? ndt_hash_cell(pr.c[0], pr.c[1], pr.c[2]) & mask : 0u;
    const Key2* kp = (const Key2*)(keys + pr.s0);
    pr.ka = kp[0]; pr.kb = kp[1];
    return pr;
}
// second half: resolve the centre entry from the keys already read, then links, payloads, gate and fold.  The
// transformed point and its cell travel with the probe (10 registers across the software-pipeline stage; recomputing
// them cost 40 instructions per point).
__device__ __forceinline__ unsigned point_terms_from_probe(const Pose& P, float xf, float yf, float zf, const CentreProbe& pr,
                                                           const int64_t* __restrict__ keys, const VoxSlot* __restrict__ slots, uint32_t mask,
                                                           const EvalParams& ep, double* acc, int64_t* cell_key) {
    if (!pr.ok) { *cell_key = NDT_KEY_EMPTY; return 0; }
    const double* q = pr.q;
    const int32_t* c = pr.c;
    const int64_t ckey = ndt_pack(c[0], c[1], c[2]);
    *cell_key = ckey;
    const uint32_t s0 = pr.s0;
    const Key2 ka = pr.ka, kb = pr.kb;
    const bool m0 = ka.x == ckey, m1 = ka.y == ckey, m2 = kb.x == ckey, m3 = kb.y == ckey;
    const bool e0 = ka.x == NDT_KEY_EMPTY, e1 = ka.y == NDT_KEY_EMPTY, e2 = kb.x == NDT_KEY_EMPTY, e3 = kb.y == NDT_KEY_EMPTY;
    const uint32_t off = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u;
    const bool found = m0 || (!e0 && (m1 || (!e1 && (m2 || (!e2 && m3)))));
    uint32_t fs0 = found ? (s0 + off) & mask : SLOT_NONE;
    bool pending = !(m0 || e0 || m1 || e1 || m2 || e2 || m3 || e3);
    uint32_t cur = (s0 + 4) & mask;
    while (__any(pending)) {          // chains longer than four slots (rare)
        const int64_t k = keys[cur];
        const bool hit = pending && k == ckey, stop = pending && (hit || k == NDT_KEY_EMPTY);
        fs0 = hit ? cur : fs0;
        pending = pending && !stop;
        cur = pending ? (cur + 1) & mask : cur;
    }
    // centre slot: payload and the six links in one 64-byte read
    const int4* sp = (const int4*)(slots + (fs0 != SLOT_NONE ? fs0 : 0u));
    const int4 a = sp[0];
    const float4 b = *(const float4*)(sp + 1);
    const int4 c2 = sp[2], d = sp[3];
    const bool have = fs0 != SLOT_NONE;
    uint32_t fs[7];
    fs[0] = fs0;
    fs[1] = have ? (uint32_t)c2.z : SLOT_NONE; fs[2] = have ? (uint32_t)c2.w : SLOT_NONE;
    fs[3] = have ? (uint32_t)d.x : SLOT_NONE;  fs[4] = have ? (uint32_t)d.y : SLOT_NONE;
    fs[5] = have ? (uint32_t)d.z : SLOT_NONE;  fs[6] = have ? (uint32_t)d.w : SLOT_NONE;
    Payload pay[7];
    pay[0].a = a; pay[0].b = b; pay[0].c = make_float2(__int_as_float(c2.x), __int_as_float(c2.y));
#pragma unroll
    for (int prb = 1; prb < 7; ++prb) pay[prb] = load_payload(slots, fs[prb]);
    double S[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    const unsigned hitmask = gate_and_sum<7>(pay, fs, q, c, ep, S, g, acc);
    fold_point(P, q, hitmask, S, g, acc);
    return hitmask;
}
__device__ __forceinline__ unsigned point_terms_linked(const Pose& P, float xf, float yf, float zf, const int64_t* __restrict__ keys,
                                                       const VoxSlot* __restrict__ slots, uint32_t mask, const EvalParams& ep,
                                                       double* acc, int64_t* cell_key) {
    const CentreProbe pr = centre_probe(P, xf, yf, zf, keys, mask, ep);
    return point_terms_from_probe(P, xf, yf, zf, pr, keys, slots, mask, ep, acc, cell_key);
}

// A lane's points i, i + stride, ... < hi through the linked NEARBY7 path with one software-pipeline stage: the centre keys
// of the NEXT point are requested before the current point is worked on, so a lane waits for three dependent round trips
// per point instead of four.  (xf, yf, zf) = point i, already loaded.
__device__ __forceinline__ void lane_points_linked(const Pose& P, const float* __restrict__ xs, const float* __restrict__ ys,
                                                   const float* __restrict__ zs, int64_t i, int64_t hi, int64_t stride, float xf, float yf, float zf,
                                                   const int64_t* __restrict__ keys, const VoxSlot* __restrict__ slots, uint32_t mask,
                                                   const EvalParams& ep, double* acc) {
    CentreProbe pr = centre_probe(P, xf, yf, zf, keys, mask, ep);
    while (i < hi) {
        const int64_t in = i + stride;
        float xn = 0.f, yn = 0.f, zn = 0.f;
        if (in < hi) { xn = xs[in]; yn = ys[in]; zn = zs[in]; }
        const CentreProbe prn = centre_probe(P, xn, yn, zn, keys, mask, ep);
        int64_t ck;
        point_terms_from_probe(P, xf, yf, zf, pr, keys, slots, mask, ep, acc, &ck);
        pr = prn; xf = xn; yf = yn; zf = zn; i = in;
    }
}

template <int K, bool CELLS, bool LINKED>
__global__ __launch_bounds__(RES1_THREADS) void k_residual(const float* __restrict__ xs, const float* __restrict__ ys,
                                                          const float* __restrict__ zs, int n, const int64_t* __restrict__ keys,
                                                          const VoxSlot* __restrict__ slots, uint32_t mask, const GnState* __restrict__ st,
                                                          EvalParams ep, double* __restrict__ partials, int force, int64_t* keys_out,
                                                          uint8_t* mask_out) {
    // the first point's coordinates are requested before the state is read, so the two misses overlap
    int i = blockIdx.x * RES1_THREADS + threadIdx.x;
    float xf = 0.f, yf = 0.f, zf = 0.f;
    if (i < n) { xf = xs[i]; yf = ys[i]; zf = zs[i]; }
    if (!force && st->done) return;
    Pose P;
#pragma unroll
    for (int j = 0; j < 9; ++j) P.R[j] = st->R[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) P.t[j] = st->t[j];
    double acc[ACC_N];
#pragma unroll
    for (int j = 0; j < ACC_N; ++j) acc[j] = 0.0;
    if (LINKED && !CELLS) {
        lane_points_linked(P, xs, ys, zs, i, n, (int64_t)gridDim.x * RES1_THREADS, xf, yf, zf, keys, slots, mask, ep, acc);
    } else {
        while (i < n) {
            int64_t ck;
            const unsigned hm = LINKED ? point_terms_linked(P, xf, yf, zf, keys, slots, mask, ep, acc, &ck)
                                       : point_terms<K>(P, xf, yf, zf, keys, slots, mask, ep, acc, &ck);
            if (CELLS) { keys_out[i] = ck; mask_out[i] = (uint8_t)hm; }
            i += gridDim.x * RES1_THREADS;
            if (i < n) { xf = xs[i]; yf = ys[i]; zf = zs[i]; }
        }
    }
    block_store_row<RES1_THREADS>(acc, partials + (size_t)blockIdx.x * ACC_N);
}

// Chained form for the single-scan Gauss-Newton loop: every block first redoes, redundantly and with identical bits,
// what k_reduce_solve would have done with the PREVIOUS launch's rows (fixed-order sum + step, state ping-ponged
// between st_in and st_out; block 0 publishes st_out), then runs the residual pass at the new pose.  One launch per
// iteration instead of two: the 4 us one-block kernel, its launch gap and its end-of-kernel write-back leave the
// critical path; the price is ~60 KB of L2 reads and one serial solve per block.  n_prev = 0: no pending step.
template <int RG> __device__ __forceinline__ void wave0_finish(double (*sm)[ACC_N + 1], GnState* st, const SolveParams& sp, bool do_solve);
template <class F>
__device__ __forceinline__ void row_group_sums32_on512(const double* __restrict__ partials, int n_partials, double (*sm)[ACC_N + 1], F&& after_loads);

struct InitPose { double Rt[12]; int use; };

// Sharded loop, peer exchange: waits (all threads of the block call it) until every rank's row of all-reduce `seq` has
// arrived in this rank's mailbox — thread r polls rank r's flag (relaxed polls, then one system-scope ACQUIRE load that pairs with the
// release store the writer put after its rows) — then adds the rows up in rank order into sm[0] (the other RG - 1 groups are zero, so
// the column tree that follows returns exactly this sum: every loop form and every rank gets the same bits).  The rows are
// read with system-scope atomic loads, which bypass the caches, after the block barrier that follows the polls.  No
// system-scope FENCE here: one in 230 blocks invalidates the L2 under the residual pass that follows (measured: 112 ->
// 380 us per align); the acquire is on the flag load itself.  Bounded: returns false (uniformly) when a peer never
// delivers; the caller marks the state NDT_ERR_COMM instead of hanging.
#define PX_SPIN_LIMIT (1 << 21)
template <int RG>
__device__ __forceinline__ bool peer_gather(const PeerWait& pw, const double* rows, double (*sm)[ACC_N + 1]) {
    __shared__ int s_px_ok;
    if (threadIdx.x == 0) s_px_ok = 1;
    __syncthreads();
    if ((int)threadIdx.x < pw.n) {
        int spins = 0;
        bool seen = true;
        while (__hip_atomic_load(pw.flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < pw.seq) {
            if (++spins > PX_SPIN_LIMIT) { s_px_ok = 0; seen = false; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        // the polls are relaxed; ONE acquire load of the flag once it has been seen orders the row reads below after the
        // writer's release store (NDT_PX_NO_ACQUIRE: A/B switch of the build, tools/exp_comm.py)
#ifndef NDT_PX_NO_ACQUIRE
        if (seen) (void)__hip_atomic_load(pw.flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    }
    __syncthreads();                        // the flags were seen before any row is requested
    if (threadIdx.x < ACC_N) {
        const unsigned long long* src = (const unsigned long long*)rows + threadIdx.x;
        double v = 0.0;
        for (int r = 0; r < pw.n; ++r)
            v += __longlong_as_double((long long)__hip_atomic_load(src + (size_t)r * ACC_N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        sm[0][threadIdx.x] = v;
#pragma unroll
        for (int g = 1; g < RG; ++g) sm[g][threadIdx.x] = 0.0;
    }
    __syncthreads();
    return s_px_ok != 0;
}

// Diagnostic build only (-DNDT_STAMPS, tools/exp_stamps.py): s_memtime at the phase boundaries of the chained launch, from
// thread 0 of two blocks, into a buffer no other code reads.  The product library is built without it.
#ifdef NDT_STAMPS
__device__ unsigned long long g_ndt_wstamps[8][4];      // block 100, per wave: pass start, pass end, butterfly end
#define WSTAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 100) g_ndt_wstamps[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int ndt_debug_wstamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ndt_wstamps), sizeof(g_ndt_wstamps)); }
__device__ unsigned long long g_ndt_stamps[2][8];
#define STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) g_ndt_stamps[blockIdx.x ? 1 : 0][k] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int ndt_debug_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ndt_stamps), sizeof(g_ndt_stamps)); }
#else
#define STAMP(k) do { } while (0)
#define WSTAMP(k) do { } while (0)
#endif
static_assert(offsetof(GnState, R) == 0 && offsetof(GnState, t) == 72, "InitPose fills the first twelve words of GnState");

template <int K, bool LINKED>
__global__ __launch_bounds__(RES1_THREADS) void k_residual_chain(const double* __restrict__ prev_partials, const GnState* __restrict__ st_in,
                                                                const float* __restrict__ xs, const float* __restrict__ ys,
                                                                const float* __restrict__ zs, int n, int n_prev, int mode, const int64_t* __restrict__ keys,
                                                                const VoxSlot* __restrict__ slots, uint32_t mask,
                                                                GnState* __restrict__ st_out,
                                                                SolveParams sp, EvalParams ep, double* __restrict__ partials, InitPose init, PeerWait pw, AosIn aos) {
    static_assert(sizeof(GnState) % 8 == 0, "GnState is copied as 8-byte words");
    constexpr int ST_WORDS = (int)(sizeof(GnState) / 8);
    __shared__ double sm[32][ACC_N + 1];
    __shared__ GnState sst;
    STAMP(0);
    // The first thirteen words of the argument block — everything up to `mode` — arrive in scalar registers with the wave
    // (kernel-argument preload, Makefile); the point, state and row reads below depend on nothing else, so they are on
    // their way before the scalar loads of the remaining arguments return.  mode: bit 0 = init.use, bit 1 = aos.p set,
    // bit 2 = pw.flags set (the structs themselves sit behind those loads).
    const bool use_init = mode & 1, has_aos = mode & 2, has_pw = mode & 4;
    int i = blockIdx.x * RES1_THREADS + threadIdx.x;
    float xf = 0.f, yf = 0.f, zf = 0.f;
    // (aos.p: the first launch after ndt_set_source_async reads the caller's records and leaves the three-array copy for the
    // launches after it — every point is read by exactly one lane of one block — instead of an unpack kernel in front)
    if (i < n) {
        if (has_aos) {
            const float* r = (const float*)(aos.p + (size_t)i * aos.stride);
            xf = r[0]; yf = r[1]; zf = r[2];
            aos.ox[i] = xf; aos.oy[i] = yf; aos.oz[i] = zf;
        } else { xf = xs[i]; yf = ys[i]; zf = zs[i]; }
    }
    // the state: from st_in, or (first launch of an align) a fresh one at the pose that came as a kernel argument,
    // which saves the host-to-device copy in front of the loop (R[9], t[3] are the first twelve words; the rest is zero)
    double sword = 0.0;
    if (threadIdx.x < ST_WORDS && !use_init) sword = ((const double*)st_in)[threadIdx.x];       // requested here, stored below
    auto store_state = [&]() {
        if (threadIdx.x < ST_WORDS) ((double*)&sst)[threadIdx.x] = use_init ? (threadIdx.x < 12 ? init.Rt[threadIdx.x] : 0.0) : sword;
    };
    bool rows_ok = true;                // (sharded loop: the rows are the ranks' rows in this rank's mailbox, once they are all there)
    if (n_prev > 0 && has_pw) { store_state(); rows_ok = peer_gather<32>(pw, prev_partials, sm); }      // (ends with a barrier too)
    else if (n_prev > 0) row_group_sums32_on512(prev_partials, n_prev, sm, store_state);      // ends with a barrier: sst is complete too
    else { store_state(); __syncthreads(); }
    STAMP(1);
    // (uniform) the rest of the previous step runs in the block's first wave, with no block barrier until the pose is out:
    // column tree, world-frame sums -> H, b (sst still holds the pose the rows were computed at), Gauss-Newton step
    if (n_prev > 0 && threadIdx.x < 64 && !sst.done) {
        if (rows_ok) wave0_finish<32>(sm, &sst, sp, true);
        else if (threadIdx.x == 0) { sst.status = NDT_ERR_COMM; sst.done = 1; }
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    if (blockIdx.x == 0 && threadIdx.x < ST_WORDS) ((double*)st_out)[threadIdx.x] = ((const double*)&sst)[threadIdx.x];
    if (sst.done) return;
    WSTAMP(0);
    Pose P;
#pragma unroll
    for (int j = 0; j < 9; ++j) P.R[j] = sst.R[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) P.t[j] = sst.t[j];
    double acc[ACC_N];
#pragma unroll
    for (int j = 0; j < ACC_N; ++j) acc[j] = 0.0;
    // (the plain loop: with the prologue's state this kernel has no registers left for lane_points_linked's extra stage —
    // it spills 18 and an align of 117 594 points takes 0.165 ms instead of 0.120)
    while (i < n) {
        int64_t ck;
        if (LINKED) point_terms_linked(P, xf, yf, zf, keys, slots, mask, ep, acc, &ck);
        else point_terms<K>(P, xf, yf, zf, keys, slots, mask, ep, acc, &ck);
        i += gridDim.x * RES1_THREADS;
        if (i < n) {
            if (has_aos) {
                const float* r = (const float*)(aos.p + (size_t)i * aos.stride);
                xf = r[0]; yf = r[1]; zf = r[2];
                aos.ox[i] = xf; aos.oy[i] = yf; aos.oz[i] = zf;
            } else { xf = xs[i]; yf = ys[i]; zf = zs[i]; }
        }
    }
    STAMP(4);
    WSTAMP(1);
    block_store_row<RES1_THREADS>(acc, partials + (size_t)blockIdx.x * ACC_N);
    STAMP(5);
    WSTAMP(2);
}

// Offline replay (BASELINE.json configs[4]): blockIdx.y = alignment problem ("pair"), blockIdx.x = block of that
// pair's scan.  Scans are stored back to back in xs/ys/zs; pair p reads scan pair_scan[p] (or p) and its own GnState.
template <int K, bool LINKED>
__global__ __launch_bounds__(RES_THREADS) void k_residual_batch(const float* __restrict__ xs, const float* __restrict__ ys,
                                                                const float* __restrict__ zs, const int64_t* __restrict__ scan_off,
                                                                const int32_t* __restrict__ pair_scan, const int64_t* __restrict__ keys,
                                                                const VoxSlot* __restrict__ slots, uint32_t mask,
                                                                const GnState* __restrict__ states, EvalParams ep,
                                                                double* __restrict__ partials, int rows_per_pair, int force) {
    const int pair = blockIdx.y;
    const int scan = pair_scan ? pair_scan[pair] : pair;
    const int64_t lo = scan_off[scan], hi = scan_off[scan + 1];
    const GnState* st = states + pair;
    int64_t i = lo + (int64_t)blockIdx.x * RES_THREADS + threadIdx.x;
    float xf = 0.f, yf = 0.f, zf = 0.f;
    if (i < hi) { xf = xs[i]; yf = ys[i]; zf = zs[i]; }
    if (!force && st->done) return;
    Pose P;
#pragma unroll
    for (int j = 0; j < 9; ++j) P.R[j] = st->R[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) P.t[j] = st->t[j];
    double acc[ACC_N];
#pragma unroll
    for (int j = 0; j < ACC_N; ++j) acc[j] = 0.0;
    if (LINKED) {
        lane_points_linked(P, xs, ys, zs, i, hi, (int64_t)gridDim.x * RES_THREADS, xf, yf, zf, keys, slots, mask, ep, acc);
    } else {
        while (i < hi) {
            int64_t ck;
            point_terms<K>(P, xf, yf, zf, keys, slots, mask, ep, acc, &ck);
            i += (int64_t)gridDim.x * RES_THREADS;
            if (i < hi) { xf = xs[i]; yf = ys[i]; zf = zs[i]; }
        }
    }
    block_store_row<RES_THREADS>(acc, partials + ((size_t)pair * rows_per_pair + blockIdx.x) * ACC_N);
}

// Blocks for a launch over n points of one problem, `problems` of them side by side.  One point per lane while
// the whole launch is small (latency-bound: every wave should be resident at once); more points per lane once the
// launch would not fit the GPU in one go, which amortises the per-block reduction tail and shrinks the row count.
int residual_grid(size_t n, int n_cu, size_t problems, int nearby) {
    if (n_cu <= 0) n_cu = 256;
    if (problems <= 1) {                              // k_residual: 512-thread blocks
        // one point per lane while that fits one round of resident blocks (one 512-thread block per CU at 256 VGPRs); beyond
        // that the lanes loop, the launch stays one round and its rows stay few enough for the chained loop (2 M points:
        // 0.72 -> 0.53 ms per 10-iteration align against a 2048-block grid)
        size_t g = (n + RES1_THREADS - 1) / RES1_THREADS;
        const size_t cap = (size_t)n_cu;
        // (rounding a nearly full grid up to one block per CU with the points dealt evenly — 256 blocks of 460 instead of 230 of
        // 512 — measured worse in rounds 2 and 3: chained launch 8.12 -> 8.38 us, the pass alone unchanged)
        return (int)(g < 1 ? 1 : g > cap ? cap : g);
    }
    // Batched launches: 256-thread blocks, two resident per CU (241 VGPRs), each lane works through `ppt` points.  The
    // grid is chosen against the tail effect: time ~ (rounds of resident blocks) x (points per lane + 2 for the block's
    // reduction and row store).  In effect: the fewest points per lane for which the whole launch is ONE round of resident
    // blocks — 8 pairs of 118 k points run as 464 blocks of 8 points per lane rather than 528 blocks of 7 (two rounds),
    // 32 pairs as 512 blocks of 29, 128 pairs as 512 blocks of 115 (measured sweeps: DESIGN.md §5).
    // (NEARBY1 needs 138 VGPRs and is bandwidth-bound: measured best with four blocks per CU in flight)
    const size_t resident = (size_t)(nearby == 1 ? 4 : 2) * (size_t)n_cu;
    size_t best_ppt = 1, best_cost = ~(size_t)0;
    for (size_t ppt = 128; ppt >= 1; --ppt) {
        const size_t rows = (n + RES_THREADS * ppt - 1) / (RES_THREADS * ppt);
        const size_t rounds = (rows * problems + resident - 1) / resident;
        const size_t cost = rounds * (ppt + 2);
        if (cost < best_cost) { best_cost = cost; best_ppt = ppt; }
    }
    size_t g = (n + RES_THREADS * best_ppt - 1) / (RES_THREADS * best_ppt);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

void launch_residual(const float* x, const float* y, const float* z, int n, const int64_t* keys, const VoxSlot* slots, uint32_t mask,
                     const GnState* st, const EvalParams& ep, double* partials, int grid, int force,
                     int64_t* keys_out, uint8_t* mask_out, int linked, hipStream_t s) {
#define LAUNCH(KK, CC, LL) k_residual<KK, CC, LL><<<grid, RES1_THREADS, 0, s>>>(x, y, z, n, keys, slots, mask, st, ep, partials, force, keys_out, mask_out)
    if (ep.nearby == 7) {
        if (linked) { if (keys_out) LAUNCH(7, true, true); else LAUNCH(7, false, true); }
        else { if (keys_out) LAUNCH(7, true, false); else LAUNCH(7, false, false); }
    } else { if (keys_out) LAUNCH(1, true, false); else LAUNCH(1, false, false); }
#undef LAUNCH
}

void launch_residual_chain(const float* x, const float* y, const float* z, int n, const int64_t* keys, const VoxSlot* slots, uint32_t mask,
                           const GnState* st_in, GnState* st_out, const double* prev_partials, int n_prev, const SolveParams& sp,
                           const EvalParams& ep, double* partials, int grid, int linked, const GnState* init_state, const PeerWait& pw,
                           const AosIn& aos, hipStream_t s) {
    InitPose init{};
    if (init_state) {
        for (int j = 0; j < 9; ++j) init.Rt[j] = init_state->R[j];
        for (int j = 0; j < 3; ++j) init.Rt[9 + j] = init_state->t[j];
        init.use = 1;
    }
    const int mode = (init.use ? 1 : 0) | (aos.p ? 2 : 0) | (pw.flags ? 4 : 0);
#define LAUNCH(KK, LL) k_residual_chain<KK, LL><<<grid, RES1_THREADS, 0, s>>>(prev_partials, st_in, x, y, z, n, n_prev, mode, keys, slots, mask, st_out, sp, ep, partials, init, pw, aos)
    if (ep.nearby == 7) { if (linked) LAUNCH(7, true); else LAUNCH(7, false); }
    else LAUNCH(1, false);
#undef LAUNCH
}

// ------------------------------------------------------------------ fixed-order final sum + Gauss-Newton step

// 1 / x for a finite positive x: hardware estimate + two Newton steps (full double precision to an ulp or two; the IEEE
// division expands to five times as many instructions, and the Gauss-Newton step sits on every iteration's critical
// path: one lane, one dependent chain).  The step is compared with the oracle at 1e-4 m / 1e-5 rad, not bit for bit.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// Reduced world-frame sums (fold_point) -> H and b at the pose with rotation R (row-major):
// H_rr = R^T M R, H_rt = R^T N, b_rot = -R^T (sum u x g); S, -sum g, chi2 and the counters are already final.
// hb_entry gives entry e of the converted row in one fixed operation order, so a block can convert with one lane per
// entry (the latency-critical single-scan kernels) or one lane for all of them (raw_to_Hb) and get the same bits.
// decode tables as immediates (a table in memory would put a dependent load chain into the single-scan prologue):
// entry e -> kind (1: H_rr(i,j), 2: H_rt(i,j), 3: b_rot(i), 0: as is), i, j, two bits each; slot of M(k,l), four bits each
#define HB_KIND_BITS 0x0000fc002a6a5a95ull
#define HB_I_BITS 0x000090002a955000ull
#define HB_J_BITS 0x0000000024a49924ull
#define HB_M_BITS 0x0000000b72761210ull
__device__ __forceinline__ int hb_m_slot(int k, int l) { return (int)((HB_M_BITS >> (4 * (3 * k + l))) & 15ull); }
__device__ __forceinline__ int hb_n_slot(int k, int j) { return (k == 0 ? 3 : k == 1 ? 8 : 12) + j; }
// (branch-free: every lane reads the operands of all three kinds — some twenty LDS reads issued together, one wait — and
// selects; three divergent paths with a wait each took three times as long on the single-scan loop's critical path)
__device__ __forceinline__ double hb_entry(const double* R, const double* raw, int e) {
    const int kind = (int)((HB_KIND_BITS >> (2 * e)) & 3ull), i = (int)((HB_I_BITS >> (2 * e)) & 3ull), j = (int)((HB_J_BITS >> (2 * e)) & 3ull);
    const double own = raw[e];
    const double Rj0 = R[j], Rj1 = R[3 + j], Rj2 = R[6 + j], Ri0 = R[i], Ri1 = R[3 + i], Ri2 = R[6 + i];
    double t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double mr = (raw[hb_m_slot(k, 0)] * Rj0 + raw[hb_m_slot(k, 1)] * Rj1) + raw[hb_m_slot(k, 2)] * Rj2;   // (M R)(k,j)
        const double nk = raw[hb_n_slot(k, j)], wk = raw[21 + k];
        t[k] = kind == 1 ? mr : kind == 2 ? nk : wk;
    }
    const double v = (Ri0 * t[0] + Ri1 * t[1]) + Ri2 * t[2];
    return kind == 0 ? own : kind == 3 ? -v : v;
}

// ---- the Gauss-Newton step, in the block's first wave.  dx = H^-1 b, R <- R Exp(dx[0:3]), t <- t + dx[3:6] (Exp: Rodrigues as
// src/slam/reference/math_utils.h:256-269, identity below 1e-7 rad).  It sits on the critical path of every iteration of the
// single-scan loop (one dependent chain, every block waiting).  Every lane solves the 6x6 system for itself by block
// elimination (below); lane e < 9 then computes one entry of the new rotation.  Every form of the loop (chained, two
// launches, batched, sharded, static) calls this one function, so they all take identical steps.
// The step is taken in the WORLD frame, straight from the reduced sums (fold_point): with J = diag(R, I) — orthogonal —
// H_body = J^T H_w J and b_body = J^T b_w, so dx_body = J^T dx_w, and R Exp(R^T phi_w) = Exp(phi_w) R: the same step
// and the same norm, without converting 27 numbers through R first (1 060 cycles of this chain, every iteration).  H and b
// in the caller's frame ([rotation, translation], right perturbation: factor.h:243-251) are only needed when a state is
// handed back, so st->acc gets them (hb_entry, at the pose the sums were taken at) when the state finishes here.
// all 64 lanes of the wave call it; raw = the 32 reduced accumulators (LDS); st may be LDS or global
__device__ __forceinline__ void gn_step_wave(GnState* st, const double* raw, double* xch, const SolveParams& sp) {
    const int lane = threadIdx.x & 63;
    const double* hb = raw;               // H_w = [M N; N^T S] sits in the same slots as the converted row's blocks
    if ((long long)raw[ACC_NEFF] < (long long)sp.min_effective_pts) {
        const double v = lane < ACC_N ? hb_entry(st->R, raw, lane) : 0.0;
        if (lane < ACC_N) st->acc[lane] = v;
        if (lane == 0) { st->status = NDT_ERR_TOO_FEW_POINTS; st->done = 1; }
        return;
    }
    // dx = H^-1 b by block elimination, H = [A B; B^T C] (A: rotation, C: translation block = sum of the information
    // matrices): C^-1 and the Schur complement's inverse as 3x3 adjugates.  Every lane computes the whole solve for itself
    // from the 27 numbers in LDS (broadcast reads): ~130 float64 operations with dependent chains a dozen deep and no
    // cross-lane traffic, against the six pivot-to-pivot round trips (reciprocal, v_readlane broadcasts, an LDS transpose)
    // of the lane-parallel LDL^T this replaces (3 360 -> see DESIGN.md §5 cycles for the step).  H is positive definite
    // exactly when C and the Schur complement are (leading minors), which is the failure test.
    const double A00 = hb[0], A01 = hb[1], A02 = hb[2], A11 = hb[6], A12 = hb[7], A22 = hb[11];
    const double B00 = hb[3], B01 = hb[4], B02 = hb[5], B10 = hb[8], B11 = hb[9], B12 = hb[10], B20 = hb[12], B21 = hb[13], B22 = hb[14];
    const double C00 = hb[15], C01 = hb[16], C02 = hb[17], C11 = hb[18], C12 = hb[19], C22 = hb[20];
    const double r0 = -hb[21], r1 = -hb[22], r2 = -hb[23], r3 = hb[24], r4 = hb[25], r5 = hb[26];      // b_w = [-(sum u x g); -sum g]
    // C^-1
    const double k00 = fma(C11, C22, -(C12 * C12)), k01 = fma(C02, C12, -(C01 * C22)), k02 = fma(C01, C12, -(C02 * C11));
    const double k11 = fma(C00, C22, -(C02 * C02)), k12 = fma(C01, C02, -(C00 * C12)), k22 = fma(C00, C11, -(C01 * C01));
    const double detC = fma(C02, k02, fma(C01, k01, C00 * k00));
    bool spd = C00 > 0.0 && k22 > 0.0 && detC > 0.0;          // (false for NaN)
    const double iC = fast_rcp(detC);
    const double i00 = k00 * iC, i01 = k01 * iC, i02 = k02 * iC, i11 = k11 * iC, i12 = k12 * iC, i22 = k22 * iC;
    // T = B C^-1
    const double T00 = fma(B02, i02, fma(B01, i01, B00 * i00)), T01 = fma(B02, i12, fma(B01, i11, B00 * i01)), T02 = fma(B02, i22, fma(B01, i12, B00 * i02));
    const double T10 = fma(B12, i02, fma(B11, i01, B10 * i00)), T11 = fma(B12, i12, fma(B11, i11, B10 * i01)), T12 = fma(B12, i22, fma(B11, i12, B10 * i02));
    const double T20 = fma(B22, i02, fma(B21, i01, B20 * i00)), T21 = fma(B22, i12, fma(B21, i11, B20 * i01)), T22 = fma(B22, i22, fma(B21, i12, B20 * i02));
    // S = A - T B^T (symmetric)
    const double S00 = A00 - fma(T02, B02, fma(T01, B01, T00 * B00)), S01 = A01 - fma(T02, B12, fma(T01, B11, T00 * B10));
    const double S02 = A02 - fma(T02, B22, fma(T01, B21, T00 * B20)), S11 = A11 - fma(T12, B12, fma(T11, B11, T10 * B10));
    const double S12 = A12 - fma(T12, B22, fma(T11, B21, T10 * B20)), S22 = A22 - fma(T22, B22, fma(T21, B21, T20 * B20));
    const double m00 = fma(S11, S22, -(S12 * S12)), m01 = fma(S02, S12, -(S01 * S22)), m02 = fma(S01, S12, -(S02 * S11));
    const double m11 = fma(S00, S22, -(S02 * S02)), m12 = fma(S01, S02, -(S00 * S12)), m22 = fma(S00, S11, -(S01 * S01));
    const double detS = fma(S02, m02, fma(S01, m01, S00 * m00));
    spd = spd && S00 > 0.0 && m22 > 0.0 && detS > 0.0;
    if (!spd) {                            // (uniform: every lane computed the same numbers)
        const double v = lane < ACC_N ? hb_entry(st->R, raw, lane) : 0.0;
        if (lane < ACC_N) st->acc[lane] = v;
        if (lane == 0) { st->status = NDT_ERR_SINGULAR; st->done = 1; }
        return;
    }
    const double iS = fast_rcp(detS);
    // x1 = S^-1 (b1 - T b2),  x2 = C^-1 b2 - T^T x1
    const double w0 = r0 - fma(T02, r5, fma(T01, r4, T00 * r3)), w1 = r1 - fma(T12, r5, fma(T11, r4, T10 * r3)), w2 = r2 - fma(T22, r5, fma(T21, r4, T20 * r3));
    const double dx0 = fma(m02, w2, fma(m01, w1, m00 * w0)) * iS, dx1 = fma(m12, w2, fma(m11, w1, m01 * w0)) * iS, dx2 = fma(m22, w2, fma(m12, w1, m02 * w0)) * iS;
    const double u0 = fma(i02, r5, fma(i01, r4, i00 * r3)), u1 = fma(i12, r5, fma(i11, r4, i01 * r3)), u2 = fma(i22, r5, fma(i12, r4, i02 * r3));
    const double dx3 = u0 - fma(T20, dx2, fma(T10, dx1, T00 * dx0)), dx4 = u1 - fma(T21, dx2, fma(T11, dx1, T01 * dx0)), dx5 = u2 - fma(T22, dx2, fma(T12, dx1, T02 * dx0));
    const double x = lane == 0 ? dx0 : lane == 1 ? dx1 : lane == 2 ? dx2 : lane == 3 ? dx3 : lane == 4 ? dx4 : dx5;      // lanes 0..5: dx[lane]
    const double nrm2 = ((((dx0 * dx0 + dx1 * dx1) + dx2 * dx2) + dx3 * dx3) + dx4 * dx4) + dx5 * dx5;
    (void)xch;
    // new rotation Exp(phi_w) R, Exp = I + sin(th) K + (1 - cos(th)) (a a^T - I), K = hat(a), a = phi_w / th:
    // lane e = 3 r + c computes ROW r of Exp and from it entry (r, c) of the product
    const int e = lane < 9 ? lane : 8, r = e / 3, c = e - 3 * r;
    const double th2 = (dx0 * dx0 + dx1 * dx1) + dx2 * dx2;
    double E0 = r == 0 ? 1.0 : 0.0, E1 = r == 1 ? 1.0 : 0.0, E2 = r == 2 ? 1.0 : 0.0;      // E[r][m], m = 0, 1, 2
    if (th2 > 1e-14) {
        // Exp = I + A hat(phi) + B (phi phi^T - th^2 I) with A = sin(th)/th, B = (1 - cos(th))/th^2: both are series in th^2,
        // so no square root, no reciprocal and no unit axis on this chain (Taylor to th^18 / th^16 up to 0.64 = 0.8^2:
        // truncation below 1e-17; beyond, a step of more than 45 degrees, the closed forms)
        double A, B;
        if (th2 <= 0.64) {
            double pa = -1.0 / 121645100408832000.0;           // -1/19!
            pa = fma(pa, th2, 1.0 / 355687428096000.0);          // 1/17!
            pa = fma(pa, th2, -1.0 / 1307674368000.0);           // -1/15!
            pa = fma(pa, th2, 1.0 / 6227020800.0);               // 1/13!
            pa = fma(pa, th2, -1.0 / 39916800.0);                // -1/11!
            pa = fma(pa, th2, 1.0 / 362880.0);                   // 1/9!
            pa = fma(pa, th2, -1.0 / 5040.0);                    // -1/7!
            pa = fma(pa, th2, 1.0 / 120.0);                      // 1/5!
            pa = fma(pa, th2, -1.0 / 6.0);                       // -1/3!
            A = fma(pa, th2, 1.0);
            double pb = 1.0 / 6402373705728000.0;              // 1/18!
            pb = fma(pb, th2, -1.0 / 20922789888000.0);          // -1/16!
            pb = fma(pb, th2, 1.0 / 87178291200.0);              // 1/14!
            pb = fma(pb, th2, -1.0 / 479001600.0);               // -1/12!
            pb = fma(pb, th2, 1.0 / 3628800.0);                  // 1/10!
            pb = fma(pb, th2, -1.0 / 40320.0);                   // -1/8!
            pb = fma(pb, th2, 1.0 / 720.0);                      // 1/6!
            pb = fma(pb, th2, -1.0 / 24.0);                      // -1/4!
            B = fma(pb, th2, 0.5);                               // 1/2!
        } else {
            const double th = sqrt(th2);
            double sn, cs;
            sincos(th, &sn, &cs);
            A = sn / th; B = (1.0 - cs) / th2;
        }
        const double pc = r == 0 ? dx0 : r == 1 ? dx1 : dx2;
        // row r of hat(phi): r = 0: (0, -p2, p1); r = 1: (p2, 0, -p0); r = 2: (-p1, p0, 0)
        const double k0 = r == 1 ? dx2 : r == 2 ? -dx1 : 0.0, k1 = r == 0 ? -dx2 : r == 2 ? dx0 : 0.0, k2 = r == 0 ? dx1 : r == 1 ? -dx0 : 0.0;
        E0 += fma(A, k0, B * (dx0 * pc - (r == 0 ? th2 : 0.0)));
        E1 += fma(A, k1, B * (dx1 * pc - (r == 1 ? th2 : 0.0)));
        E2 += fma(A, k2, B * (dx2 * pc - (r == 2 ? th2 : 0.0)));
    }
    const double Rn = (E0 * st->R[c] + E1 * st->R[3 + c]) + E2 * st->R[6 + c];
    const double tn = st->t[lane >= 3 && lane < 6 ? lane - 3 : 0] + x;      // lanes 3..5 hold dx[3..5]
    const int it = st->iter + 1;
    // (uniform) the state finishes with this step: H, b of this iteration in the caller's frame, at the pose they were summed at
    if (nrm2 < sp.eps * sp.eps || it >= sp.max_iterations) {
        const double v = lane < ACC_N ? hb_entry(st->R, raw, lane) : 0.0;
        if (lane < ACC_N) st->acc[lane] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();       // every lane has read the old pose before any lane writes the new one
    if (lane < 9) st->R[lane] = Rn;
    if (lane >= 3 && lane < 6) st->t[lane - 3] = tn;
    if (lane < 6) st->dx[lane] = x;
    if (lane == 0) {
        st->dx_norm2 = nrm2;
        st->iter = it;
        if (nrm2 < sp.eps * sp.eps) { st->converged = 1; st->done = 1; }
        if (it >= sp.max_iterations) st->done = 1;
    }
}

#define RED_THREADS 1024
// thread (row group r = tid / 32, column c = tid % 32) adds partial rows r, r+RG, ... in order into sm[r][c]; the RG sums
// of a column are then combined by one lane in a fixed binary tree (wave0_finish).  Ends with a block barrier.
template <int RG>
__device__ __forceinline__ void row_group_sums(const double* __restrict__ partials, int n_partials, double (*sm)[ACC_N + 1], bool live) {
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    double v = 0.0;
    if (live) {
        // rows r, r+RG, ... : eight loads in flight at a time (they are L2 misses: other CUs just wrote them),
        // added strictly in row order
        for (int i0 = r; i0 < n_partials; i0 += RG * 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + RG * u;
                x[u] = i < n_partials ? partials[(size_t)i * ACC_N + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += x[u];
        }
    }
    sm[r][c] = v;
    __syncthreads();
}
// row_group_sums<32> on a 512-thread block, same bits: thread (r = tid / 16, c2 = tid % 16) adds rows r, r + 32, ... of the
// column PAIR (2 c2, 2 c2 + 1) — the same rows in the same order as thread (r, c) of the 1024-thread form, two columns at a
// time, so that every row read is a 16-byte load: 8-byte accesses go through the L1 at 0.54-0.70x the 16-byte rate
// (MI355X_MICROARCH.md), and these 59 KB per block sit on every iteration's critical path (round 3; the 8-byte form kept
// the two halves of 16 groups in one thread instead).  Leaves all 32 group sums per column: the tree starts at its first level.
// `after_loads` runs once, after the first (usually the only) batch of row reads has been requested and before any of
// them is used: work that waits for an earlier read goes there without holding the row reads back.
template <class F>
__device__ __forceinline__ void row_group_sums32_on512(const double* __restrict__ partials, int n_partials, double (*sm)[ACC_N + 1], F&& after_loads) {
    const int c2 = threadIdx.x & 15, r = threadIdx.x >> 4;
    double va = 0.0, vb = 0.0;
    bool first = true;
    for (int i0 = r; i0 < n_partials; i0 += 32 * 8) {
        // (all eight reads unconditional, from a clamped row, and masked afterwards: no branch between them, one wait)
        double2 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *(const double2*)(partials + (size_t)min(i0 + 32 * u, n_partials - 1) * ACC_N + 2 * c2);
        if (first) { after_loads(); first = false; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 32 * u >= n_partials) x[u] = make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 8; ++u) va += x[u].x;
#pragma unroll
        for (int u = 0; u < 8; ++u) vb += x[u].y;
    }
    if (first) after_loads();
    sm[r][2 * c2] = va;
    sm[r][2 * c2 + 1] = vb;
    __syncthreads();
}
// the fixed binary tree over the RG sums of column `c`: level o adds entry r + o to entry r for r < o, o = RG/2 .. 1
template <int RG>
__device__ __forceinline__ double column_tree(double (*sm)[ACC_N + 1], int c) {
    double v[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) v[r] = sm[r][c];
#pragma unroll
    for (int o = RG / 2; o >= 1; o >>= 1)
#pragma unroll
        for (int r = 0; r < o; ++r) v[r] += v[r + o];
    return v[0];
}
// The block's first wave (all 64 lanes), after the barrier that made the RG sums per column visible: column tree, the
// raw row -> H, b at the state's pose (one lane per entry; also stored in st->acc), and the Gauss-Newton step.  Only
// wave-level ordering in here: LDS operations of one wave execute in issue order.  sm has at least 4 rows.
// column_tree<32> by all 64 lanes of a wave, same tree, same bits: lane c + 32 h takes the groups 8 h .. 8 h + 7 and their
// partners 16 above (the tree's first level), the two halves meet through one row swap per value (second level:
// v[k] + v[k + 8]), lanes < 32 finish the last three levels.  Half the LDS reads and adds of the one-lane-per-column form
// on the chained launch's critical path.  Returns the column total in lanes < 32.
__device__ __forceinline__ double column_tree32_wave(double (*sm)[ACC_N + 1], int lane) {
    const int c = lane & 31, h = lane >> 5;
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = sm[8 * h + k][c] + sm[8 * h + k + 16][c];      // level 16
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = swap_add32(v[k], v[k]);                          // level 8: lanes < 32 get v_h0[k] + v_h1[k]
#pragma unroll
    for (int o = 4; o >= 1; o >>= 1)
#pragma unroll
        for (int k = 0; k < o; ++k) v[k] += v[k + o];
    return v[0];
}
template <int RG>
__device__ __forceinline__ void wave0_finish(double (*sm)[ACC_N + 1], GnState* st, const SolveParams& sp, bool do_solve) {
    const int lane = threadIdx.x & 63;
    if constexpr (RG == 32) {
        const double tot = column_tree32_wave(sm, lane);               // (all 64 lanes; a lane reads its own column only)
        if (lane < ACC_N) sm[0][lane] = tot;
    } else {
        if (lane < ACC_N) sm[0][lane] = column_tree<RG>(sm, lane);      // a lane reads and writes its own column only
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    STAMP(6);
    STAMP(7);
    if (do_solve) gn_step_wave(st, sm[0], &sm[2][0], sp);      // (converts into st->acc itself when the state finishes)
    else {
        const double v = lane < ACC_N ? hb_entry(st->R, sm[0], lane) : 0.0;
        if (lane < ACC_N) st->acc[lane] = v;
    }
}

// The first wave leaves a copy of the state in `mirror` (pinned host memory the device can write; nullptr: no copy), so
// the host reads the result after its stream wait without a copy command in between.
static_assert(sizeof(GnState) % 8 == 0, "mirror_state copies 8-byte words");
// The word after the state gets `seq` once the copy is out (system-scope release): a host that polls it sees the result a
// few microseconds before the stream's completion signal would tell it.
__device__ __forceinline__ void mirror_state(const GnState* st, GnState* mirror, unsigned long long seq) {
    if (!mirror) return;
    __threadfence();                          // the lanes' stores to *st are done before other lanes read them back
    const unsigned long long* s = (const unsigned long long*)st;
    unsigned long long* d = (unsigned long long*)mirror;
    for (int i = threadIdx.x & 63; i < (int)(sizeof(GnState) / 8); i += 64) d[i] = s[i];
    __threadfence_system();                   // (every lane: its words are out before lane 0's flag)
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(d + sizeof(GnState) / 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(RED_THREADS) void k_reduce_solve(const double* __restrict__ partials, int n_partials, GnState* st,
                                                              SolveParams sp, int do_solve, int force, GnState* mirror, unsigned long long mirror_seq, PeerWait pw) {
    __shared__ double sm[32][ACC_N + 1];
    // the rows are summed unconditionally (their loads overlap the read of st->done); a finished state just
    // discards the result
    const int done = st->done;
    bool rows_ok = true;
    if (pw.flags) rows_ok = peer_gather<32>(pw, partials, sm);      // (sharded loop: the ranks' rows in this rank's mailbox)
    else row_group_sums<32>(partials, n_partials, sm, true);
    if (threadIdx.x >= 64) return;
    if (force || !done) {
        if (rows_ok) wave0_finish<32>(sm, st, sp, do_solve != 0);
        else if (threadIdx.x == 0) { st->status = NDT_ERR_COMM; st->done = 1; }
    }
    mirror_state(st, mirror, mirror_seq);
}
// sharded loop without RCCL: the local fixed-order sum goes straight into every rank's mailbox (this rank's too), row
// `rank` of the parity in use, with system-scope stores over xGMI; the sequence flags follow the rows (release fence by
// the one wave that wrote them).  Every rank then sums the same rows in rank order: identical bits everywhere.
__global__ __launch_bounds__(RED_THREADS) void k_reduce_rows_p2p(const double* __restrict__ partials, int n_partials, const GnState* st, PeerRows peers,
                                                                 int n_ranks, int rank, unsigned long long seq, int force) {
    __shared__ double sm[32][ACC_N + 1];
    row_group_sums<32>(partials, n_partials, sm, force || !st->done);   // a finished state contributes zeros (stale partials)
    if (threadIdx.x >= 64) return;
    double v = 0.0;
    if (threadIdx.x < ACC_N) v = column_tree<32>(sm, threadIdx.x);
    for (int p = 0; p < n_ranks; ++p)
        if (threadIdx.x < ACC_N)
            __hip_atomic_store((unsigned long long*)(peers.rows[p] + (size_t)rank * ACC_N + threadIdx.x), (unsigned long long)__double_as_longlong(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // the flags follow the rows: a system-scope release store (this one wave wrote all the rows; waiting for the rows'
    // acknowledgements alone, without the release, measured the same)
    if ((int)threadIdx.x < n_ranks) __hip_atomic_store(peers.flags[threadIdx.x] + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// multi-GPU form: the local sum goes to a row that RCCL all-reduces, and the step reads the reduced row
__global__ __launch_bounds__(RED_THREADS) void k_reduce_rows(const double* __restrict__ partials, int n_partials, const GnState* st,
                                                             double* row_out, int force) {
    __shared__ double sm[32][ACC_N + 1];
    row_group_sums<32>(partials, n_partials, sm, force || !st->done);   // a finished state contributes zeros (stale partials)
    if (threadIdx.x < ACC_N) row_out[threadIdx.x] = column_tree<32>(sm, threadIdx.x);
}
__global__ __launch_bounds__(64) void k_solve_only(GnState* st, const double* reduced, SolveParams sp, int do_solve, int force) {
    __shared__ double sm[4][ACC_N + 1];
    if (!force && st->done) return;
    if (threadIdx.x < ACC_N) { sm[0][threadIdx.x] = reduced[threadIdx.x]; sm[1][threadIdx.x] = 0.0; }
    __syncthreads();
    wave0_finish<2>(sm, st, sp, do_solve != 0);      // (the second "group" is zero: the tree adds +0)
}

#define BRED_THREADS 256
// batched form: one block per pair; counts the pairs that finish in *n_done
__device__ __forceinline__ void reduce_solve_pair(const double* partials, int rows_per_pair, GnState* st, const SolveParams& sp, int do_solve, int force, int32_t* n_done) {
    __shared__ double sm[BRED_THREADS / 32][ACC_N + 1];
    row_group_sums<BRED_THREADS / 32>(partials, rows_per_pair, sm, true);
    if (!force && st->done) return;            // (uniform over the block: nobody has written the state yet)
    if (threadIdx.x < 64) {
        wave0_finish<BRED_THREADS / 32>(sm, st, sp, do_solve != 0);
        if (do_solve && threadIdx.x == 0 && st->done) atomicAdd(n_done, 1);
    }
}
__global__ __launch_bounds__(BRED_THREADS) void k_reduce_solve_batch(const double* __restrict__ partials, int rows_per_pair, GnState* states,
                                                                     SolveParams sp, int do_solve, int force, int32_t* n_done) {
    reduce_solve_pair(partials + (size_t)blockIdx.x * rows_per_pair * ACC_N, rows_per_pair, states + blockIdx.x, sp, do_solve, force, n_done);
}

void launch_residual_batch(const float* x, const float* y, const float* z, const int64_t* scan_off, const int32_t* pair_scan,
                           const int64_t* keys, const VoxSlot* slots, uint32_t mask, const GnState* states, const EvalParams& ep,
                           double* partials, int rows_per_pair, int pair0, int n_pairs, int force, int linked, hipStream_t s) {
    dim3 grid((unsigned)rows_per_pair, (unsigned)n_pairs);
    const int32_t* ps = pair_scan ? pair_scan + pair0 : nullptr;
    // without pair_scan pair p reads scan p, so the offsets are shifted together with the states
    const int64_t* so = pair_scan ? scan_off : scan_off + pair0;
    double* pp = partials + (size_t)pair0 * rows_per_pair * ACC_N;
    if (ep.nearby == 7 && linked) k_residual_batch<7, true><<<grid, RES_THREADS, 0, s>>>(x, y, z, so, ps, keys, slots, mask, states + pair0, ep, pp, rows_per_pair, force);
    else if (ep.nearby == 7) k_residual_batch<7, false><<<grid, RES_THREADS, 0, s>>>(x, y, z, so, ps, keys, slots, mask, states + pair0, ep, pp, rows_per_pair, force);
    else k_residual_batch<1, false><<<grid, RES_THREADS, 0, s>>>(x, y, z, so, ps, keys, slots, mask, states + pair0, ep, pp, rows_per_pair, force);
}
void launch_reduce_solve_batch(const double* partials, int rows_per_pair, GnState* states, int n_pairs, const SolveParams& sp,
                               int do_solve, int force, int32_t* n_done, hipStream_t s) {
    k_reduce_solve_batch<<<n_pairs, BRED_THREADS, 0, s>>>(partials, rows_per_pair, states, sp, do_solve, force, n_done);
}
void launch_reduce_solve(const double* partials, int n_partials, GnState* st, const SolveParams& sp, int do_solve, int force, GnState* mirror,
                         unsigned long long mirror_seq, const PeerWait& pw, hipStream_t s) {
    k_reduce_solve<<<1, RED_THREADS, 0, s>>>(partials, n_partials, st, sp, do_solve, force, mirror, mirror_seq, pw);
}
void launch_reduce_rows_p2p(const double* partials, int n_partials, const GnState* st, const PeerRows& peers, int n_ranks, int rank,
                            unsigned long long seq, int force, hipStream_t s) {
    k_reduce_rows_p2p<<<1, RED_THREADS, 0, s>>>(partials, n_partials, st, peers, n_ranks, rank, seq, force);
}
void launch_reduce_rows(const double* partials, int n_partials, const GnState* st, double* row_out, int force, hipStream_t s) {
    k_reduce_rows<<<1, RED_THREADS, 0, s>>>(partials, n_partials, st, row_out, force);
}
void launch_solve_only(GnState* st, const double* reduced, const SolveParams& sp, int do_solve, int force, hipStream_t s) {
    k_solve_only<<<1, 64, 0, s>>>(st, reduced, sp, do_solve, force);
}
