/*
 * ndt_oracle.c — CPU restatement of the NDT scan-matching path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build,
 * load or call this file.  The product (lio-slam_amd/, include/) never does.
 *
 * PARITY UNPINNED against the reference: ergevv/lio-slam holds no NDT source, no
 * tests and no golden vectors.  Its live path is third-party PCL 1.14
 * pcl::NormalDistributionsTransform (src/slam/include/process.h:86; calls at
 * src/slam/src/process.cc:78,87,94,95,147,373-376), absent from this image; the
 * algorithm family the north star names (voxel-hash Gauss-Newton NDT with
 * NEARBY1/NEARBY7) is the SLAM-book IncNdt3d/Ndt3d whose header is also absent
 * (src/slam/reference/lio_preinteg.h:13) — only its call sites
 * (src/slam/reference/lio_preinteg.cc:37-38,108,116-121,133) and math helpers
 * (src/slam/reference/math_utils.h) are in the tree.  This file restates that
 * published algorithm and follows the in-tree pieces line by line where they
 * exist; it is pinned by the independent float64 NumPy vectors in tests/golden/.
 *
 * What follows what:
 *   voxel key, FLOOR_F32     PCL VoxelGridCovariance grid rule, ijk = floor(p_f32 * inv_leaf_f32)
 *                            (the cell assignment process.cc:78/147 executes)
 *   voxel key, ROUND_F64     SLAM-book CastToInt(round(p * inv_res))  
 *   batch mean / covariance  ComputeMeanAndCov, src/slam/reference/math_utils.h:55-72
 *                            (mean; sample covariance over (n-1))
 *   incremental merge        UpdateMeanAndCov, src/slam/reference/math_utils.h:87-98
 *   hat / Exp                SKEW_SYM_MATRIX and Exp, src/slam/reference/math_utils.h:242-269
 *   keyframe transform       pcl::transformPointCloud at src/slam/src/process.cc:134-136
 *   Gauss-Newton align       IncNdt3d::AlignNdt as called at src/slam/reference/lio_preinteg.cc:116-121;
 *                            parameters from src/slam/src/process.cc:373-376
 *   H ordering               [rotation, translation], right perturbation: src/slam/include/factor.h:243-251
 *
 * Arithmetic contract shared with the HIP engine (this is what makes cell
 * assignment bit-exact; build with -ffp-contract=off):
 *   q' = ((R[j][0]*x + R[j][1]*y) + R[j][2]*z) + t[j]      in float64, that order
 *   source key  = floor(q' * (double)inv_leaf_f32)          (FLOOR) or round() (ROUND)
 *   target key  = floorf(x_f32 * inv_leaf_f32)              (FLOOR) or round((double)x * inv)
 *   keyframe pt = (float)q'                                  then keyed as a target point
 *   Sigma^-1 e  = fma(I2, e2, fma(I1, e1, I0*e0)) per row, chi2 = fma(e2, g2, fma(e1, g1, e0*g0))   (the gate is bit-exact)
 * The voxel table holds, per voxel, what the engine's residual kernel reads:
 * mean as a float32 offset from the voxel origin key*res, and the six unique
 * entries of the regularised inverse covariance as float32 (SURVEY.md §8(d):
 * 40-byte payload).  `quantize = 0` keeps them in float64 instead, which is the
 * looser "pure double" cross-check in the tests (quantize = 2, 3: fixed-point
 * offsets, an experiment no engine path uses — see finalize_voxel).
 */
#include "../include/ndt_mi355.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>

#define KEY_BIAS (1 << 20)
#define KEY_EMPTY INT64_MIN

typedef struct {
    int64_t key;
    int32_t n;
    int32_t valid;
    double mean_off[3]; /* mean - key*res */
    double cov[6];      /* xx xy xz yy yz zz */
    double pay_off[3];  /* what the residual pass reads (float32-rounded if quantize) */
    double pay_info[6];
    uint64_t touch;     /* (cloud number << 32) | index of the last point of that cloud that fell into the voxel */
} ora_voxel;

typedef struct ora_handle {
    ndt_config cfg;
    int quantize;
    /* hash: slot -> voxel index */
    int64_t* slot_key;
    int32_t* slot_vox;
    size_t cap; /* power of two */
    ora_voxel* vox;
    size_t n_vox, vox_cap;
    float* src; /* xyz packed */
    size_t n_src;
    float inv_leaf_f32;
    double inv_res_d, res_d;
    int n_threads;
    uint32_t cloud_no;  /* clouds merged so far (the LRU order's major key) */
} ora_handle;

/* ---------------------------------------------------------------- keys */

static int64_t pack_key(int32_t ix, int32_t iy, int32_t iz) {
    return ((int64_t)(ix + KEY_BIAS) << 42) | ((int64_t)(iy + KEY_BIAS) << 21) | (int64_t)(iz + KEY_BIAS);
}
static void unpack_key(int64_t k, int32_t* ix, int32_t* iy, int32_t* iz) {
    *ix = (int32_t)((k >> 42) & 0x1FFFFF) - KEY_BIAS;
    *iy = (int32_t)((k >> 21) & 0x1FFFFF) - KEY_BIAS;
    *iz = (int32_t)(k & 0x1FFFFF) - KEY_BIAS;
}
static int in_range(double v) { return v >= -(double)KEY_BIAS && v < (double)KEY_BIAS; }

/* PCL grid rule on a float32 map point; returns 0 if out of key range */
static int target_cell(const ora_handle* h, const float p[3], int32_t c[3]) {
    for (int j = 0; j < 3; ++j) {
        double v;
        if (h->cfg.key_rule == NDT_KEY_FLOOR_F32) {
            float f = floorf(p[j] * h->inv_leaf_f32);
            v = (double)f;
        } else {
            v = round((double)p[j] * h->inv_res_d);
        }
        if (!in_range(v)) return 0;
        c[j] = (int32_t)v;
    }
    return 1;
}
/* same rule on a float64 transformed source point */
static int source_cell(const ora_handle* h, const double q[3], int32_t c[3]) {
    for (int j = 0; j < 3; ++j) {
        double v = q[j] * h->inv_res_d;
        v = (h->cfg.key_rule == NDT_KEY_FLOOR_F32) ? floor(v) : round(v);
        if (!in_range(v)) return 0;
        c[j] = (int32_t)v;
    }
    return 1;
}

/* ---------------------------------------------------------------- hash */

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
static int32_t map_find(const ora_handle* h, int64_t key) {
    if (!h->cap) return -1;
    size_t m = h->cap - 1, s = (size_t)mix64((uint64_t)key) & m;
    for (;;) {
        int64_t k = h->slot_key[s];
        if (k == key) return h->slot_vox[s];
        if (k == KEY_EMPTY) return -1;
        s = (s + 1) & m;
    }
}
static void map_rehash(ora_handle* h, size_t cap) {
    int64_t* nk = (int64_t*)malloc(cap * sizeof(int64_t));
    int32_t* nv = (int32_t*)malloc(cap * sizeof(int32_t));
    for (size_t i = 0; i < cap; ++i) nk[i] = KEY_EMPTY;
    for (size_t v = 0; v < h->n_vox; ++v) {
        size_t s = (size_t)mix64((uint64_t)h->vox[v].key) & (cap - 1);
        while (nk[s] != KEY_EMPTY) s = (s + 1) & (cap - 1);
        nk[s] = h->vox[v].key; nv[s] = (int32_t)v;
    }
    free(h->slot_key); free(h->slot_vox);
    h->slot_key = nk; h->slot_vox = nv; h->cap = cap;
}
static int32_t map_insert(ora_handle* h, int64_t key) {
    if ((h->n_vox + 1) * 2 > h->cap) map_rehash(h, h->cap ? h->cap * 2 : 1024);
    if (h->n_vox == h->vox_cap) {
        h->vox_cap = h->vox_cap ? h->vox_cap * 2 : 1024;
        h->vox = (ora_voxel*)realloc(h->vox, h->vox_cap * sizeof(ora_voxel));
    }
    size_t m = h->cap - 1, s = (size_t)mix64((uint64_t)key) & m;
    while (h->slot_key[s] != KEY_EMPTY) s = (s + 1) & m;
    h->slot_key[s] = key; h->slot_vox[s] = (int32_t)h->n_vox;
    memset(&h->vox[h->n_vox], 0, sizeof(ora_voxel));
    h->vox[h->n_vox].key = key;
    return (int32_t)h->n_vox++;
}

/* ---------------------------------------------------------------- 3x3 symmetric eigen (cyclic Jacobi) */

static void jacobi3(const double c[6], double lam[3], double V[3][3]) {
    double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-34 * dia || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) { /* A <- A * G */
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) { /* A <- G^T * A */
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    lam[0] = a[0][0]; lam[1] = a[1][1]; lam[2] = a[2][2];
}

/* Regularised inverse: eigenvalues below ratio*lambda_max are raised to it
 * (PCL min_covar_eigvalue_mult_). Returns 0 if the
 * voxel cannot be used. */
/* float32 rounding of a table entry (quantize = 1).  Through a volatile on purpose: gcc 11.4 at -O3 -march=x86-64-v3 vectorises
 * `q ? (double)(float)x : x` in a short loop and DROPS the conversion in the vector part (the first two of three offsets came out
 * unrounded until round 3; tests/test_oracle_golden.py::test_payload_rounding_is_applied holds the fix). */
static double f32_round(double x) { volatile float f = (float)x; return (double)f; }

static int finalize_voxel(const ora_handle* h, ora_voxel* v) {
    v->valid = 0;
    memset(v->pay_info, 0, sizeof v->pay_info);
    for (int j = 0; j < 3; ++j) {
        if (h->quantize >= 2) {
            /* EXPERIMENT ONLY (tools/exp_payload_quant.py, DESIGN.md §5 "32-byte payload"): the mean offset as unsigned fixed
             * point over [-res/2, 3 res/2) with 21 (quantize = 2) or 24 (quantize = 3) bits.  No engine path uses it. */
            const double scale = h->quantize == 2 ? 1048576.0 : 8388608.0;
            double r = floor((v->mean_off[j] / h->res_d + 0.5) * scale + 0.5);
            if (r < 0.0) r = 0.0;
            if (r > 2.0 * scale - 1.0) r = 2.0 * scale - 1.0;
            v->pay_off[j] = (r / scale - 0.5) * h->res_d;
        } else v->pay_off[j] = h->quantize ? f32_round(v->mean_off[j]) : v->mean_off[j];
    }
    if (v->n < h->cfg.min_pts_in_voxel) return 0;
    if (h->cfg.info_rule == NDT_INFO_ADD_1E3) {
        /* IncNdt3d: info = (Sigma + 1e-3 I)^-1; a single point gets 100 I */
        double inf[6] = {100.0, 0.0, 0.0, 100.0, 0.0, 100.0};
        if (v->n > 1) {
            const double* cv = v->cov;
            const double a00 = cv[0] + 1e-3, a01 = cv[1], a02 = cv[2], a11 = cv[3] + 1e-3, a12 = cv[4], a22 = cv[5] + 1e-3;
            const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
            const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
            const double det = (a00 * c00 + a01 * c01) + a02 * c02;
            if (!(det > 0.0) || !isfinite(det)) return 0;
            inf[0] = c00 / det; inf[1] = c01 / det; inf[2] = c02 / det; inf[3] = c11 / det; inf[4] = c12 / det; inf[5] = c22 / det;
        }
        for (int t = 0; t < 6; ++t) if (!isfinite(inf[t])) return 0;
        for (int t = 0; t < 6; ++t) v->pay_info[t] = h->quantize ? f32_round(inf[t]) : inf[t];
        v->valid = 1;
        return 1;
    }
    double lam[3], V[3][3];
    jacobi3(v->cov, lam, V);
    double lmax = fmax(lam[0], fmax(lam[1], lam[2]));
    if (!(lmax > 0.0) || !isfinite(lmax)) return 0;
    double fl = h->cfg.eig_floor_ratio * lmax;
    double il[3];
    for (int j = 0; j < 3; ++j) il[j] = 1.0 / (lam[j] < fl ? fl : lam[j]);
    double info[6];
    int t = 0;
    for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c, ++t)
            info[t] = (V[r][0] * il[0] * V[c][0] + V[r][1] * il[1] * V[c][1]) + V[r][2] * il[2] * V[c][2];
    for (t = 0; t < 6; ++t) {
        if (!isfinite(info[t])) return 0;
        v->pay_info[t] = h->quantize ? f32_round(info[t]) : info[t];
    }
    v->valid = 1;
    return 1;
}

/* ---------------------------------------------------------------- map build / merge */

typedef struct { int64_t key; uint32_t idx; } key_idx;

/* stable LSD radix sort on the key (ties keep input order = ascending idx) */
static void sort_key_idx(key_idx* a, size_t n) {
    key_idx* b = (key_idx*)malloc(n * sizeof(key_idx));
    for (int pass = 0; pass < 8; ++pass) {
        size_t cnt[257] = {0};
        int sh = pass * 8;
        for (size_t i = 0; i < n; ++i) cnt[(((uint64_t)a[i].key >> sh) & 255) + 1]++;
        int skip = 0;
        for (int d = 0; d < 256; ++d) if (cnt[d + 1] == n) skip = 1;
        if (skip) continue;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; ++i) b[cnt[((uint64_t)a[i].key >> sh) & 255]++] = a[i];
        memcpy(a, b, n * sizeof(key_idx));
    }
    free(b);
}

/* IncNdt3d's voxel capacity [UPSTREAM-MEM: slam_in_autonomous_driving ch7/ndt_inc.cc AddCloud; SURVEY.md App. B]: voxels
 * sit in an LRU list, every point moves its voxel to the front, a new voxel is pushed to the front and the tail is
 * popped when the size reaches capacity_.  At the end of a cloud the list is ordered by "last point that touched the
 * voxel", and it holds the capacity - 1 most recently touched voxels — as long as the cloud did not come back to a voxel it
 * had evicted itself (IncNdt3d then dereferences an erased entry; not reproduced).  So: after the merge, drop the least
 * recently touched voxels until capacity - 1 are left. */
static int cmp_u64(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
static void evict_lru(ora_handle* h) {
    const size_t capn = (size_t)h->cfg.capacity;
    if (h->cfg.capacity <= 0 || h->n_vox < capn) return;
    const size_t k = h->n_vox - (capn - 1);
    uint64_t* t = (uint64_t*)malloc(h->n_vox * sizeof(uint64_t));
    for (size_t v = 0; v < h->n_vox; ++v) t[v] = h->vox[v].touch;
    qsort(t, h->n_vox, sizeof(uint64_t), cmp_u64);
    const uint64_t thr = t[k - 1];          /* stamps are distinct: exactly k voxels are at or below it */
    free(t);
    size_t w = 0;
    for (size_t v = 0; v < h->n_vox; ++v)
        if (h->vox[v].touch > thr) h->vox[w++] = h->vox[v];
    h->n_vox = w;
    size_t cap = 1024;
    while (cap < 2 * (w + 1)) cap *= 2;
    free(h->slot_key); free(h->slot_vox);
    h->slot_key = NULL; h->slot_vox = NULL; h->cap = 0;
    {   /* map_rehash is defined above and rebuilds the slots from vox[] */
        int64_t* nk = (int64_t*)malloc(cap * sizeof(int64_t));
        int32_t* nv = (int32_t*)malloc(cap * sizeof(int32_t));
        for (size_t i = 0; i < cap; ++i) nk[i] = KEY_EMPTY;
        for (size_t v = 0; v < h->n_vox; ++v) {
            size_t sl = (size_t)mix64((uint64_t)h->vox[v].key) & (cap - 1);
            while (nk[sl] != KEY_EMPTY) sl = (sl + 1) & (cap - 1);
            nk[sl] = h->vox[v].key; nv[sl] = (int32_t)v;
        }
        h->slot_key = nk; h->slot_vox = nv; h->cap = cap;
    }
}

/* merge `n` float32 world points into the map */
static int merge_points(ora_handle* h, const float* p, size_t n) {
    h->cloud_no += 1;
    key_idx* ki = (key_idx*)malloc((n ? n : 1) * sizeof(key_idx));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        const float* q = p + 3 * i;
        if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
        int32_t c[3];
        if (!target_cell(h, q, c)) continue;
        ki[m].key = pack_key(c[0], c[1], c[2]);
        ki[m].idx = (uint32_t)i;
        ++m;
    }
    sort_key_idx(ki, m);
    size_t s = 0;
    while (s < m) {
        size_t e = s;
        int64_t key = ki[s].key;
        while (e < m && ki[e].key == key) ++e;
        int32_t c[3];
        unpack_key(key, &c[0], &c[1], &c[2]);
        double org[3] = {c[0] * h->res_d, c[1] * h->res_d, c[2] * h->res_d};
        /* batch statistics in voxel-local coordinates, ComputeMeanAndCov (math_utils.h:55-72) */
        double s1[3] = {0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = s; i < e; ++i) {
            const float* q = p + 3 * (size_t)ki[i].idx;
            double d0 = (double)q[0] - org[0], d1 = (double)q[1] - org[1], d2 = (double)q[2] - org[2];
            s1[0] += d0; s1[1] += d1; s1[2] += d2;
            s2[0] += d0 * d0; s2[1] += d0 * d1; s2[2] += d0 * d2;
            s2[3] += d1 * d1; s2[4] += d1 * d2; s2[5] += d2 * d2;
        }
        int32_t nb = (int32_t)(e - s);
        double mu[3] = {s1[0] / nb, s1[1] / nb, s1[2] / nb};
        double cv[6] = {0, 0, 0, 0, 0, 0};
        if (nb > 1) {
            double dn = (double)(nb - 1);
            cv[0] = (s2[0] - s1[0] * mu[0]) / dn; cv[1] = (s2[1] - s1[0] * mu[1]) / dn;
            cv[2] = (s2[2] - s1[0] * mu[2]) / dn; cv[3] = (s2[3] - s1[1] * mu[1]) / dn;
            cv[4] = (s2[4] - s1[1] * mu[2]) / dn; cv[5] = (s2[5] - s1[2] * mu[2]) / dn;
        }
        int32_t vi = map_find(h, key);
        if (vi < 0) {
            vi = map_insert(h, key);
            ora_voxel* v = &h->vox[vi];
            v->n = nb;
            memcpy(v->mean_off, mu, sizeof mu);
            memcpy(v->cov, cv, sizeof cv);
        } else if (h->cfg.max_pts_in_voxel > 0 && h->vox[vi].n > h->cfg.max_pts_in_voxel) {
            /* frozen (IncNdt3d): the voxel keeps what it has */
        } else {
            /* UpdateMeanAndCov (math_utils.h:87-98) */
            ora_voxel* v = &h->vox[vi];
            double hm = (double)v->n, cn = (double)nb, tot = hm + cn;
            double nm[3], dh[3], dc[3];
            for (int j = 0; j < 3; ++j) {
                nm[j] = (hm * v->mean_off[j] + cn * mu[j]) / tot;
                dh[j] = v->mean_off[j] - nm[j];
                dc[j] = mu[j] - nm[j];
            }
            int t = 0;
            for (int r = 0; r < 3; ++r)
                for (int cc = r; cc < 3; ++cc, ++t)
                    v->cov[t] = (hm * (v->cov[t] + dh[r] * dh[cc]) + cn * (cv[t] + dc[r] * dc[cc])) / tot;
            memcpy(v->mean_off, nm, sizeof nm);
            v->n += nb;
        }
        finalize_voxel(h, &h->vox[vi]);
        h->vox[vi].touch = ((uint64_t)h->cloud_no << 32) | (uint64_t)ki[e - 1].idx;   /* the stable sort keeps a voxel's points in input order */
        s = e;
    }
    free(ki);
    evict_lru(h);
    return NDT_OK;
}

static void gather_xyz(const void* pts, size_t n, size_t stride, float* out) {
    const unsigned char* b = (const unsigned char*)pts;
    for (size_t i = 0; i < n; ++i) memcpy(out + 3 * i, b + i * stride, 12);
}

static void mat_from_T(const double T[16], double R[3][3], double t[3]) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[r][c] = T[c * 4 + r];
        t[r] = T[12 + r];
    }
}
static void T_from_mat(const double R[3][3], const double t[3], double T[16]) {
    memset(T, 0, 16 * sizeof(double));
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[c * 4 + r] = R[r][c];
        T[12 + r] = t[r];
    }
    T[15] = 1.0;
}
static inline void xform(const double R[3][3], const double t[3], const float p[3], double q[3]) {
    double x = p[0], y = p[1], z = p[2];
    for (int j = 0; j < 3; ++j) q[j] = ((R[j][0] * x + R[j][1] * y) + R[j][2] * z) + t[j];
}

/* ---------------------------------------------------------------- public API */

ora_handle* ora_create(const ndt_config* cfg, int quantize) {
    ora_handle* h = (ora_handle*)calloc(1, sizeof(ora_handle));
    h->cfg = *cfg;
    h->quantize = quantize;
    h->inv_leaf_f32 = 1.0f / cfg->resolution;
    h->inv_res_d = (double)h->inv_leaf_f32;
    h->res_d = (double)cfg->resolution;
    h->n_threads = 1;
    return h;
}
static void map_clear(ora_handle* h) {
    free(h->slot_key); free(h->slot_vox); free(h->vox);
    h->slot_key = NULL; h->slot_vox = NULL; h->vox = NULL;
    h->cap = h->n_vox = h->vox_cap = 0;
    h->cloud_no = 0;
}
void ora_destroy(ora_handle* h) {
    if (!h) return;
    map_clear(h);
    free(h->src);
    free(h);
}
void ora_set_threads(ora_handle* h, int n) { h->n_threads = n < 1 ? 1 : n; }
void ora_set_params(ora_handle* h, const ndt_config* cfg) {
    h->cfg.max_iterations = cfg->max_iterations; h->cfg.eps = cfg->eps; h->cfg.nearby = cfg->nearby;
    h->cfg.res_outlier_th = cfg->res_outlier_th; h->cfg.min_effective_pts = cfg->min_effective_pts;
    h->cfg.final_eval = cfg->final_eval;
}

int ora_add_target(ora_handle* h, const void* pts, size_t n, size_t stride, const double T[16]) {
    float* p = (float*)malloc((n ? n : 1) * 12);
    gather_xyz(pts, n, stride, p);
    if (T) { /* pcl::transformPointCloud, process.cc:134-136; result stored as float32 */
        double R[3][3], t[3];
        mat_from_T(T, R, t);
        for (size_t i = 0; i < n; ++i) {
            double q[3];
            xform(R, t, p + 3 * i, q);
            p[3 * i] = (float)q[0]; p[3 * i + 1] = (float)q[1]; p[3 * i + 2] = (float)q[2];
        }
    }
    int rc = merge_points(h, p, n);
    free(p);
    return rc;
}
int ora_set_target(ora_handle* h, const void* pts, size_t n, size_t stride) {
    map_clear(h);
    return ora_add_target(h, pts, n, stride, NULL);
}
int ora_set_source(ora_handle* h, const void* pts, size_t n, size_t stride) {
    free(h->src);
    h->src = (float*)malloc((n ? n : 1) * 12);
    const unsigned char* b = (const unsigned char*)pts;
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        float q[3];
        memcpy(q, b + i * stride, 12);
        if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
        memcpy(h->src + 3 * m, q, 12);
        ++m;
    }
    h->n_src = m;
    return NDT_OK;
}
int64_t ora_source_size(const ora_handle* h) { return (int64_t)h->n_src; }
int64_t ora_num_voxels(const ora_handle* h) { return (int64_t)h->n_vox; }

static const int NB7[7][3] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};

/* acc: 21 upper-tri H (row-major), 6 b, chi2, n_eff, hits(total voxel hits incl. gated) */
static void eval_range(const ora_handle* h, const double R[3][3], const double t[3], size_t i0, size_t i1,
                       double acc[32], int64_t* keys, uint8_t* masks) {
    memset(acc, 0, 32 * sizeof(double));
    const int k = h->cfg.nearby == 7 ? 7 : 1;
    const double th = h->cfg.res_outlier_th;
    for (size_t i = i0; i < i1; ++i) {
        const float* p = h->src + 3 * i;
        double q[3];
        xform(R, t, p, q);
        int32_t c[3];
        if (!source_cell(h, q, c)) {
            if (keys) { keys[i] = KEY_EMPTY; masks[i] = 0; }
            continue;
        }
        double S[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        uint8_t mask = 0;
        for (int pr = 0; pr < k; ++pr) {
            int32_t n0 = c[0] + NB7[pr][0], n1 = c[1] + NB7[pr][1], n2 = c[2] + NB7[pr][2];
            if (!in_range(n0) || !in_range(n1) || !in_range(n2)) continue;
            int32_t vi = map_find(h, pack_key(n0, n1, n2));
            if (vi < 0) continue;
            const ora_voxel* v = &h->vox[vi];
            if (!v->valid) continue;
            acc[30] += 1.0;
            double e0 = (q[0] - (double)n0 * h->res_d) - v->pay_off[0];
            double e1 = (q[1] - (double)n1 * h->res_d) - v->pay_off[1];
            double e2 = (q[2] - (double)n2 * h->res_d) - v->pay_off[2];
            const double* I = v->pay_info;
            double g0 = fma(I[2], e2, fma(I[1], e1, I[0] * e0));
            double g1 = fma(I[4], e2, fma(I[3], e1, I[1] * e0));
            double g2 = fma(I[5], e2, fma(I[4], e1, I[2] * e0));
            double chi2 = fma(e2, g2, fma(e1, g1, e0 * g0));
            if (isnan(chi2) || chi2 > th) continue;
            mask |= (uint8_t)(1u << pr);
            for (int j = 0; j < 6; ++j) S[j] += I[j];
            g[0] += g0; g[1] += g1; g[2] += g2;
            acc[27] += chi2;
            acc[28] += 1.0;
        }
        if (keys) { keys[i] = pack_key(c[0], c[1], c[2]); masks[i] = mask; }
        if (!mask) continue;
        /* J = [ -R*hat(p) | I ]  (right perturbation; SKEW_SYM_MATRIX math_utils.h:242-247) */
        double x = p[0], y = p[1], z = p[2];
        double A[3][3];
        for (int r = 0; r < 3; ++r) {
            A[r][0] = -(R[r][1] * z - R[r][2] * y);
            A[r][1] = -(R[r][2] * x - R[r][0] * z);
            A[r][2] = -(R[r][0] * y - R[r][1] * x);
        }
        double Sm[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
        double SA[3][3];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) SA[r][cc] = (Sm[r][0] * A[0][cc] + Sm[r][1] * A[1][cc]) + Sm[r][2] * A[2][cc];
        /* H layout (upper tri, row-major): rows 0..2 = rotation, 3..5 = translation */
        double Hf[6][6];
        memset(Hf, 0, sizeof Hf);
        for (int r = 0; r < 3; ++r)
            for (int cc = r; cc < 3; ++cc)
                Hf[r][cc] = (A[0][r] * SA[0][cc] + A[1][r] * SA[1][cc]) + A[2][r] * SA[2][cc];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) Hf[r][3 + cc] = SA[cc][r]; /* A^T S */
        for (int r = 0; r < 3; ++r)
            for (int cc = r; cc < 3; ++cc) Hf[3 + r][3 + cc] = Sm[r][cc];
        int tix = 0;
        for (int r = 0; r < 6; ++r)
            for (int cc = r; cc < 6; ++cc, ++tix) acc[tix] += Hf[r][cc];
        for (int r = 0; r < 3; ++r) acc[21 + r] += -((A[0][r] * g[0] + A[1][r] * g[1]) + A[2][r] * g[2]);
        for (int r = 0; r < 3; ++r) acc[24 + r] += -g[r];
    }
}

typedef struct { const ora_handle* h; const double (*R)[3]; const double* t; size_t i0, i1; double acc[32]; } eval_job;
static void* eval_thread(void* a) {
    eval_job* j = (eval_job*)a;
    eval_range(j->h, j->R, j->t, j->i0, j->i1, j->acc, NULL, NULL);
    return NULL;
}

static void eval_pose(const ora_handle* h, const double R[3][3], const double t[3], double acc[32]) {
    int nt = h->n_threads;
    if (nt <= 1 || h->n_src < 4096) { eval_range(h, R, t, 0, h->n_src, acc, NULL, NULL); return; }
    eval_job* jobs = (eval_job*)calloc((size_t)nt, sizeof(eval_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nt, sizeof(pthread_t));
    for (int i = 0; i < nt; ++i) {
        jobs[i].h = h; jobs[i].R = R; jobs[i].t = t;
        jobs[i].i0 = h->n_src * (size_t)i / (size_t)nt;
        jobs[i].i1 = h->n_src * (size_t)(i + 1) / (size_t)nt;
        pthread_create(&th[i], NULL, eval_thread, &jobs[i]);
    }
    memset(acc, 0, 32 * sizeof(double));
    for (int i = 0; i < nt; ++i) { /* fixed-order final sum */
        pthread_join(th[i], NULL);
        for (int j = 0; j < 32; ++j) acc[j] += jobs[i].acc[j];
    }
    free(jobs); free(th);
}

static void unpack_acc(const double acc[32], double H[36], double b[6]) {
    int t = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c, ++t) { H[r * 6 + c] = acc[t]; H[c * 6 + r] = acc[t]; }
    for (int r = 0; r < 6; ++r) b[r] = acc[21 + r];
}

int ora_eval(ora_handle* h, const double T[16], double H[36], double b[6], double* chi2, int64_t* n_eff,
             double* hits) {
    if (!h->n_vox) return NDT_ERR_NO_TARGET;
    if (!h->src) return NDT_ERR_NO_SOURCE;
    double R[3][3], t[3], acc[32], Hf[36], bf[6];
    mat_from_T(T, R, t);
    eval_pose(h, R, t, acc);
    unpack_acc(acc, Hf, bf);
    if (H) memcpy(H, Hf, sizeof Hf);
    if (b) memcpy(b, bf, sizeof bf);
    if (chi2) *chi2 = acc[27];
    if (n_eff) *n_eff = (int64_t)acc[28];
    if (hits) *hits = acc[30];
    return NDT_OK;
}

int ora_source_cells(ora_handle* h, const double T[16], int64_t* keys, uint8_t* masks) {
    if (!h->n_vox) return NDT_ERR_NO_TARGET;
    if (!h->src) return NDT_ERR_NO_SOURCE;
    double R[3][3], t[3], acc[32];
    mat_from_T(T, R, t);
    eval_range(h, R, t, 0, h->n_src, acc, keys, masks);
    return NDT_OK;
}

/* 6x6 SPD solve by Cholesky; returns 0 if a pivot is not positive */
static int chol_solve6(const double Hin[36], const double b[6], double x[6]) {
    double L[6][6];
    memset(L, 0, sizeof L);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = Hin[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0.0)) return 0;
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
    return 1;
}

/* Exp of math_utils.h:256-269 (Rodrigues; identity below 1e-7 rad) */
static void so3_exp(const double w[3], double E[3][3]) {
    double th = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E[i][j] = (i == j);
    if (!(th > 1e-7)) return;
    double a[3] = {w[0] / th, w[1] / th, w[2] / th};
    double K[3][3] = {{0, -a[2], a[1]}, {a[2], 0, -a[0]}, {-a[1], a[0], 0}};
    double s = sin(th), c1 = 1.0 - cos(th);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double kk = (K[i][0] * K[0][j] + K[i][1] * K[1][j]) + K[i][2] * K[2][j];
            E[i][j] += s * K[i][j] + c1 * kk;
        }
}

int ora_align(ora_handle* h, const double guess[16], ndt_result* out) {
    if (!h->n_vox) return NDT_ERR_NO_TARGET;
    if (!h->src) return NDT_ERR_NO_SOURCE;
    double R[3][3], t[3], acc[32], H[36], b[6];
    mat_from_T(guess, R, t);
    memset(out, 0, sizeof *out);
    int rc = NDT_OK;
    memset(acc, 0, sizeof acc);
    for (int it = 0; it < h->cfg.max_iterations; ++it) {
        eval_pose(h, R, t, acc);
        if ((int64_t)acc[28] < h->cfg.min_effective_pts) { rc = NDT_ERR_TOO_FEW_POINTS; break; }
        unpack_acc(acc, H, b);
        double dx[6];
        if (!chol_solve6(H, b, dx)) { rc = NDT_ERR_SINGULAR; break; }
        double E[3][3], Rn[3][3];
        so3_exp(dx, E);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rn[i][j] = (R[i][0] * E[0][j] + R[i][1] * E[1][j]) + R[i][2] * E[2][j];
        memcpy(R, Rn, sizeof Rn);
        for (int j = 0; j < 3; ++j) t[j] += dx[3 + j];
        out->iterations = it + 1;
        double nrm = 0;
        for (int j = 0; j < 6; ++j) nrm += dx[j] * dx[j];
        out->last_dx_norm = sqrt(nrm);
        if (out->last_dx_norm < h->cfg.eps) { out->converged = 1; break; }
    }
    if (rc == NDT_OK && h->cfg.final_eval) eval_pose(h, R, t, acc);
    unpack_acc(acc, out->H, out->b);
    out->chi2 = acc[27];
    out->n_effective = (int64_t)acc[28];
    T_from_mat(R, t, out->T);
    out->status = rc;
    return rc;
}

static int cmp_vox_key(const void* a, const void* b) {
    int64_t ka = ((const ora_voxel*)a)->key, kb = ((const ora_voxel*)b)->key;
    return (ka > kb) - (ka < kb);
}

int ora_export_voxels(ora_handle* h, int64_t* keys, int32_t* counts, double* mean, double* cov, double* info,
                      uint8_t* valid, size_t cap, size_t* n_out) {
    if (n_out) *n_out = h->n_vox;
    if (cap < h->n_vox) return NDT_ERR_CAPACITY;
    ora_voxel* s = (ora_voxel*)malloc((h->n_vox ? h->n_vox : 1) * sizeof(ora_voxel));
    memcpy(s, h->vox, h->n_vox * sizeof(ora_voxel));
    qsort(s, h->n_vox, sizeof(ora_voxel), cmp_vox_key);
    static const int U[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (size_t i = 0; i < h->n_vox; ++i) {
        int32_t c[3];
        unpack_key(s[i].key, &c[0], &c[1], &c[2]);
        if (keys) keys[i] = s[i].key;
        if (counts) counts[i] = s[i].n;
        if (valid) valid[i] = (uint8_t)s[i].valid;
        for (int r = 0; r < 3; ++r) {
            if (mean) mean[3 * i + r] = (double)c[r] * h->res_d + s[i].mean_off[r];
            for (int cc = 0; cc < 3; ++cc) {
                if (cov) cov[9 * i + 3 * r + cc] = s[i].cov[U[r][cc]];
                if (info) info[9 * i + 3 * r + cc] = s[i].pay_info[U[r][cc]];
            }
        }
    }
    free(s);
    return NDT_OK;
}

/* pcl::transformPointCloud + pcl::VoxelGrid(leaf) of src/slam/src/process.cc:62-73, xyz only: one centroid per occupied
 * leaf cell (grid rule floor(p_f32 * inv_leaf_f32), as PCL's VoxelGrid), cells ordered by PCL's linear index
 * (x fastest, z slowest), centroid = float64 mean of the cell's points in input order, rounded once to float32. */
typedef struct { int32_t c[3]; uint32_t idx; } cell_idx;
static int cmp_cell_zyx(const void* a, const void* b) {
    const cell_idx *p = (const cell_idx*)a, *q = (const cell_idx*)b;
    for (int j = 2; j >= 0; --j) if (p->c[j] != q->c[j]) return p->c[j] < q->c[j] ? -1 : 1;
    return p->idx < q->idx ? -1 : p->idx > q->idx;
}
int ora_voxel_downsample(const void* pts, size_t n, size_t stride, float leaf, const double T[16], float* out, size_t cap, size_t* n_out) {
    ora_handle hh;
    memset(&hh, 0, sizeof hh);
    hh.cfg.key_rule = NDT_KEY_FLOOR_F32;
    hh.inv_leaf_f32 = 1.0f / leaf; hh.inv_res_d = (double)hh.inv_leaf_f32; hh.res_d = (double)leaf;
    float* p = (float*)malloc((n ? n : 1) * 12);
    gather_xyz(pts, n, stride, p);
    cell_idx* ci = (cell_idx*)malloc((n ? n : 1) * sizeof(cell_idx));
    double R[3][3], t[3];
    if (T) mat_from_T(T, R, t);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        float* q = p + 3 * i;
        if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
        if (T) { double w[3]; xform(R, t, q, w); q[0] = (float)w[0]; q[1] = (float)w[1]; q[2] = (float)w[2]; }
        if (!target_cell(&hh, q, ci[m].c)) continue;
        ci[m].idx = (uint32_t)i;
        ++m;
    }
    qsort(ci, m, sizeof(cell_idx), cmp_cell_zyx);
    size_t no = 0, s = 0;
    int rc = NDT_OK;
    while (s < m) {
        size_t e = s;
        double a[3] = {0, 0, 0};
        while (e < m && ci[e].c[0] == ci[s].c[0] && ci[e].c[1] == ci[s].c[1] && ci[e].c[2] == ci[s].c[2]) {
            const float* q = p + 3 * (size_t)ci[e].idx;
            a[0] += q[0]; a[1] += q[1]; a[2] += q[2];
            ++e;
        }
        if (out) {
            if (no >= cap) { rc = NDT_ERR_CAPACITY; break; }
            for (int j = 0; j < 3; ++j) out[3 * no + j] = (float)(a[j] / (double)(e - s));
        }
        ++no;
        s = e;
    }
    if (n_out) *n_out = no;
    free(ci); free(p);
    return rc;
}

int64_t ora_pack_key(int32_t x, int32_t y, int32_t z) { return pack_key(x, y, z); }

/* key of a single map point / of a single transformed source coordinate, for the key-rule vectors */
int ora_target_key(ora_handle* h, const float p[3], int32_t c[3]) { return target_cell(h, p, c); }
int ora_source_key(ora_handle* h, const double q[3], int32_t c[3]) { return source_cell(h, q, c); }

/* ================================================================================================
 * PCL compatibility mode.  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED: PCL's sources are not
 * in this image; what follows restates pcl::NormalDistributionsTransform (PCL >= 1.12 registration/impl/ndt.hpp:
 * computeTransformation, computeDerivatives, computeAngleDerivatives, computePointDerivatives, updateDerivatives,
 * computeHessian/updateHessian, computeStepLengthMT, trialValueSelectionMT, updateIntervalMT) and the radius search
 * of VoxelGridCovariance as written down in SURVEY.md Appendix A.3-A.6 — the code src/slam/src/process.cc:94
 * runs.  It is written the way PCL writes it (per neighbour, full 3x6 point Jacobian and 18x6 point Hessian,
 * 6x6 double loop); the HIP engine sums the same terms per point in factored form, so this file also checks that
 * algebra.  Arithmetic shared with the engine: the float32 transform matrix and transformed points, the cell rule
 * and the float32 centroid distance test.
 * ================================================================================================ */

#include <float.h>

void ora_pcl_default_params(ndt_pcl_params* p) {
    p->outlier_ratio = 0.55; p->step_size = 0.1; p->rotation_eps = 0.0; p->hessian_d1_as_pcl = 1; p->reserved = 0;
}

/* eq. 6.8 [Magnusson 2009], computeTransformation() */
static void pcl_gauss(double res, double outlier_ratio, double* d1, double* d2) {
    const double c1 = 10.0 * (1.0 - outlier_ratio);
    const double c2 = outlier_ratio / pow(res, 3);
    const double d3 = -log(c2);
    *d1 = -log(c1 + c2) - d3;
    *d2 = -2.0 * log((-log(c1 * exp(-0.5) + c2) - d3) / *d1);
}

/* R = Rx(p3) Ry(p4) Rz(p5), t = p0..2  (convertTransform) */
static void pcl_rt_from_vector(const double p[6], double R[3][3], double t[3]) {
    const double cx = cos(p[3]), sx = sin(p[3]), cy = cos(p[4]), sy = sin(p[4]), cz = cos(p[5]), sz = sin(p[5]);
    R[0][0] = cy * cz;                  R[0][1] = -(cy * sz);               R[0][2] = sy;
    R[1][0] = cx * sz + (sx * sy) * cz; R[1][1] = cx * cz - (sx * sy) * sz; R[1][2] = -(sx * cy);
    R[2][0] = sx * sz - (cx * sy) * cz; R[2][1] = sx * cz + (cx * sy) * sz; R[2][2] = cx * cy;
    t[0] = p[0]; t[1] = p[1]; t[2] = p[2];
}
void ora_pcl_matrix_from_vector(const double p[6], double T[16]) {
    double R[3][3], t[3];
    pcl_rt_from_vector(p, R, t);
    T_from_mat(R, t, T);
}
/* Eigen's MatrixBase::eulerAngles(0,1,2): first angle in [0, pi] */
void ora_pcl_vector_from_matrix(const double T[16], double p[6]) {
    double m[3][3], t[3];
    mat_from_T(T, m, t);
    const double PI = 3.14159265358979323846;
    double r0 = atan2(m[1][2], m[2][2]);
    const double c2 = sqrt(m[0][0] * m[0][0] + m[0][1] * m[0][1]);
    double r1;
    if (r0 > 0.0) { r0 -= PI; r1 = atan2(-m[0][2], -c2); }
    else r1 = atan2(-m[0][2], c2);
    const double s1 = sin(r0), c1 = cos(r0);
    const double r2 = atan2(s1 * m[2][0] - c1 * m[1][0], c1 * m[1][1] - s1 * m[2][1]);
    p[0] = t[0]; p[1] = t[1]; p[2] = t[2];
    p[3] = -r0; p[4] = -r1; p[5] = -r2;
}

static void pcl_cos_sin(double a, double* c, double* s) {
    if (fabs(a) < 10e-5) { *c = 1.0; *s = 0.0; } else { *c = cos(a); *s = sin(a); }
}
/* computeAngleDerivatives: rows a..h of eq. 6.19 and a2..f3 of eq. 6.21 */
static void pcl_angle_jacobian(const double p[6], double aj[8][3]) {
    double cx, cy, cz, sx, sy, sz;
    pcl_cos_sin(p[3], &cx, &sx); pcl_cos_sin(p[4], &cy, &sy); pcl_cos_sin(p[5], &cz, &sz);
    const double v[8][3] = {
        {(-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy)},   /* a */
        {(cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy)},     /* b */
        {(-sy * cz), sy * sz, cy},                                            /* c */
        {sx * cy * cz, (-sx * cy * sz), sx * sy},                             /* d */
        {(-cx * cy * cz), cx * cy * sz, (-cx * sy)},                          /* e */
        {(-cy * sz), (-cy * cz), 0},                                          /* f */
        {(cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0},             /* g */
        {(sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0}};             /* h */
    memcpy(aj, v, sizeof v);
}
static void pcl_angle_hessian(const double p[6], int d1_as_pcl, double ah[15][3]) {
    double cx, cy, cz, sx, sy, sz;
    pcl_cos_sin(p[3], &cx, &sx); pcl_cos_sin(p[4], &cy, &sy); pcl_cos_sin(p[5], &cz, &sz);
    const double v[15][3] = {
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy},      /* a2 */
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy)},   /* a3 */
        {(cx * cy * cz), (-cx * cy * sz), (cx * sy)},                         /* b2 */
        {(sx * cy * cz), (-sx * cy * sz), (sx * sy)},                         /* b3 */
        {(-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0},             /* c2 */
        {(cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0},             /* c3 */
        {(-cy * cz), (cy * sz), d1_as_pcl ? (sy) : (-sy)},                    /* d1: PCL writes +sy; eq. 6.21 has -sy */
        {(-sx * sy * cz), (sx * sy * sz), (sx * cy)},                         /* d2 */
        {(cx * sy * cz), (-cx * sy * sz), (-cx * cy)},                        /* d3 */
        {(sy * sz), (sy * cz), 0},                                            /* e1 */
        {(-sx * cy * sz), (-sx * cy * cz), 0},                                /* e2 */
        {(cx * cy * sz), (cx * cy * cz), 0},                                  /* e3 */
        {(-cy * cz), (cy * sz), 0},                                           /* f1 */
        {(-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0},            /* f2 */
        {(-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0}};           /* f3 */
    memcpy(ah, v, sizeof v);
}

/* One pass over the source: computeDerivatives (with_gradient) or computeHessian (Hessian only).
 * aj / ah are passed in because PCL refreshes the angular Hessian only when compute_hessian is true. */
static double pcl_pass(const ora_handle* h, double d1, double d2, const double p[6], const double aj[8][3], const double ah[15][3],
                       int with_gradient, int with_hessian, double g[6], double H[36], int64_t* n_nb) {
    double R[3][3], t[3];
    pcl_rt_from_vector(p, R, t);
    float M[3][4];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M[r][c] = (float)R[r][c]; M[r][3] = (float)t[r]; }
    const float r2f = (float)(h->res_d * h->res_d);
    double score = 0.0;
    int64_t nb = 0;
    if (with_gradient) memset(g, 0, 6 * sizeof(double));
    if (with_hessian) memset(H, 0, 36 * sizeof(double));
    for (size_t i = 0; i < h->n_src; ++i) {
        const float* x = h->src + 3 * i;
        float q[3];
        for (int r = 0; r < 3; ++r) q[r] = ((M[r][0] * x[0] + M[r][1] * x[1]) + M[r][2] * x[2]) + M[r][3];
        if (!isfinite(q[0]) || !isfinite(q[1]) || !isfinite(q[2])) continue;
        int32_t c[3];
        if (!target_cell(h, q, c)) continue;
        /* computePointDerivatives(x): J_E (3x6) and H_E (18x6) from the ORIGINAL point */
        double J[3][6], HE[18][6];
        memset(J, 0, sizeof J); memset(HE, 0, sizeof HE);
        J[0][0] = J[1][1] = J[2][2] = 1.0;
        double xa[8], xh[15];
        for (int k = 0; k < 8; ++k) xa[k] = (aj[k][0] * x[0] + aj[k][1] * x[1]) + aj[k][2] * x[2];
        J[1][3] = xa[0]; J[2][3] = xa[1]; J[0][4] = xa[2]; J[1][4] = xa[3]; J[2][4] = xa[4]; J[0][5] = xa[5]; J[1][5] = xa[6]; J[2][5] = xa[7];
        if (with_hessian) {
            for (int k = 0; k < 15; ++k) xh[k] = (ah[k][0] * x[0] + ah[k][1] * x[1]) + ah[k][2] * x[2];
            const double va[3] = {0, xh[0], xh[1]}, vb[3] = {0, xh[2], xh[3]}, vc[3] = {0, xh[4], xh[5]};
            const double* vd = xh + 6; const double* ve = xh + 9; const double* vf = xh + 12;
            for (int r = 0; r < 3; ++r) {
                HE[9 + r][3] = va[r];  HE[12 + r][3] = vb[r]; HE[15 + r][3] = vc[r];
                HE[9 + r][4] = vb[r];  HE[12 + r][4] = vd[r]; HE[15 + r][4] = ve[r];
                HE[9 + r][5] = vc[r];  HE[12 + r][5] = ve[r]; HE[15 + r][5] = vf[r];
            }
        }
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int32_t n0 = c[0] + dx, n1 = c[1] + dy, n2 = c[2] + dz;
                    if (!in_range(n0) || !in_range(n1) || !in_range(n2)) continue;
                    const int32_t vi = map_find(h, pack_key(n0, n1, n2));
                    if (vi < 0) continue;
                    const ora_voxel* v = &h->vox[vi];
                    if (!v->valid) continue;
                    /* radiusSearch(point, resolution) on the centroid cloud: float32 squared distance, strictly inside */
                    const float cen[3] = {(float)((double)n0 * h->res_d + v->pay_off[0]), (float)((double)n1 * h->res_d + v->pay_off[1]),
                                          (float)((double)n2 * h->res_d + v->pay_off[2])};
                    const float e0 = q[0] - cen[0], e1 = q[1] - cen[1], e2 = q[2] - cen[2];
                    const float dist2 = (e0 * e0 + e1 * e1) + e2 * e2;
                    if (!(dist2 < r2f)) continue;
                    ++nb;
                    const double xt[3] = {((double)q[0] - (double)n0 * h->res_d) - v->pay_off[0], ((double)q[1] - (double)n1 * h->res_d) - v->pay_off[1],
                                          ((double)q[2] - (double)n2 * h->res_d) - v->pay_off[2]};
                    const double* I = v->pay_info;
                    const double Ci[3][3] = {{I[0], I[1], I[2]}, {I[1], I[3], I[4]}, {I[2], I[4], I[5]}};
                    double cx[3];
                    for (int r = 0; r < 3; ++r) cx[r] = (Ci[r][0] * xt[0] + Ci[r][1] * xt[1]) + Ci[r][2] * xt[2];
                    const double qv = (xt[0] * cx[0] + xt[1] * cx[1]) + xt[2] * cx[2];
                    double e = exp(-d2 * qv / 2.0);
                    const double score_inc = -d1 * e;
                    e = d2 * e;
                    if (e > 1.0 || e < 0.0 || isnan(e)) continue;       /* updateDerivatives returns 0: no score either */
                    if (with_gradient) score += score_inc;
                    e *= d1;
                    for (int a = 0; a < 6; ++a) {
                        double cj[3];
                        for (int r = 0; r < 3; ++r) cj[r] = (Ci[r][0] * J[0][a] + Ci[r][1] * J[1][a]) + Ci[r][2] * J[2][a];
                        const double xcj = (xt[0] * cj[0] + xt[1] * cj[1]) + xt[2] * cj[2];
                        if (with_gradient) g[a] += xcj * e;
                        if (!with_hessian) continue;
                        for (int b = 0; b < 6; ++b) {
                            double cjb[3], ch[3];
                            for (int r = 0; r < 3; ++r) {
                                cjb[r] = (Ci[r][0] * J[0][b] + Ci[r][1] * J[1][b]) + Ci[r][2] * J[2][b];
                                ch[r] = (Ci[r][0] * HE[3 * a][b] + Ci[r][1] * HE[3 * a + 1][b]) + Ci[r][2] * HE[3 * a + 2][b];
                            }
                            const double xcjb = (xt[0] * cjb[0] + xt[1] * cjb[1]) + xt[2] * cjb[2];
                            const double xch = (xt[0] * ch[0] + xt[1] * ch[1]) + xt[2] * ch[2];
                            const double jcj = (J[0][b] * cj[0] + J[1][b] * cj[1]) + J[2][b] * cj[2];
                            H[a * 6 + b] += e * (-d2 * xcj * xcjb + xch + jcj);
                        }
                    }
                }
    }
    if (n_nb) *n_nb = nb;
    return score;
}

int ora_pcl_eval(ora_handle* h, const ndt_pcl_params* prm, const double p[6], double* score, double g[6], double H[36], int64_t* n_nb) {
    if (!h->n_vox) return NDT_ERR_NO_TARGET;
    if (!h->src) return NDT_ERR_NO_SOURCE;
    double d1, d2, aj[8][3], ah[15][3], gg[6], HH[36];
    pcl_gauss(h->res_d, prm->outlier_ratio, &d1, &d2);
    pcl_angle_jacobian(p, aj); pcl_angle_hessian(p, prm->hessian_d1_as_pcl, ah);
    const double s = pcl_pass(h, d1, d2, p, aj, ah, 1, H != NULL, gg, HH, n_nb);
    if (score) *score = s;
    if (g) memcpy(g, gg, sizeof gg);
    if (H) memcpy(H, HH, sizeof HH);
    return NDT_OK;
}

/* cyclic Jacobi eigen-decomposition of a symmetric 6x6 (A = V diag(lam) V^T; V row-major, columns = vectors) */
static void sym6_eig(const double Ain[36], double lam[6], double V[36]) {
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
                const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 6; ++k) { const double a = A[k][pp], b = A[k][qq]; A[k][pp] = c * a - s * b; A[k][qq] = s * a + c * b; }
                for (int k = 0; k < 6; ++k) { const double a = A[pp][k], b = A[qq][k]; A[pp][k] = c * a - s * b; A[qq][k] = s * a + c * b; }
                for (int k = 0; k < 6; ++k) { const double a = V[k * 6 + pp], b = V[k * 6 + qq]; V[k * 6 + pp] = c * a - s * b; V[k * 6 + qq] = s * a + c * b; }
            }
    }
    for (int i = 0; i < 6; ++i) lam[i] = A[i][i];
}
/* JacobiSVD(H).solve(b) for a symmetric H: pseudo-inverse over the eigenvalues above Eigen's default rank threshold */
static void pcl_svd_solve(const double H[36], const double b[6], double x[6]) {
    double lam[6], V[36], mx = 0.0;
    sym6_eig(H, lam, V);
    for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(lam[i]));
    const double thr = 6.0 * DBL_EPSILON * mx;
    memset(x, 0, 6 * sizeof(double));
    for (int i = 0; i < 6; ++i) {
        if (!(fabs(lam[i]) > thr)) continue;
        double d = 0.0;
        for (int k = 0; k < 6; ++k) d += V[k * 6 + i] * b[k];
        d /= lam[i];
        for (int k = 0; k < 6; ++k) x[k] += d * V[k * 6 + i];
    }
}

/* More-Thuente pieces, ndt.hpp */
static double mt_psi(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
static double mt_dpsi(double g_a, double g_0, double mu) { return g_a - mu * g_0; }
static int mt_update_interval(double* a_l, double* f_l, double* g_l, double* a_u, double* f_u, double* g_u, double a_t, double f_t, double g_t) {
    if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }                                           /* U1 / a */
    if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }                               /* U2 / b */
    if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }   /* U3 / c */
    return 1;
}
static double mt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t) {
    if (a_t == a_l && a_t == a_u) return a_t;
    int cond;
    if (a_t == a_l) cond = 4;
    else if (f_t > f_l) cond = 1;
    else if (g_t * g_l < 0) cond = 2;
    else if (fabs(g_t) <= fabs(g_l)) cond = 3;
    else cond = 4;
    if (cond == 1) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        const double w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
        return 0.5 * (a_q + a_c);
    }
    if (cond == 2) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        const double w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
        return a_s;
    }
    if (cond == 3) {
        const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        const double w = sqrt(z * z - g_t * g_l);
        const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        if (a_t > a_l) return fmin(a_t + 0.66 * (a_u - a_t), a_n);
        return fmax(a_t + 0.66 * (a_u - a_t), a_n);
    }
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
    const double w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
}

typedef struct {
    const ora_handle* h; const ndt_pcl_params* prm; double d1, d2;
    double aj[8][3], ah[15][3];     /* angular_jacobian_ / angular_hessian_ members */
    int evals; int64_t n_nb;
} pcl_ctx;
static double pcl_derivs(pcl_ctx* c, const double p[6], int with_hessian, double g[6], double H[36]) {
    pcl_angle_jacobian(p, c->aj);
    if (with_hessian) pcl_angle_hessian(p, c->prm->hessian_d1_as_pcl, c->ah);
    ++c->evals;
    return pcl_pass(c->h, c->d1, c->d2, p, c->aj, c->ah, 1, with_hessian, g, H, &c->n_nb);
}

static double pcl_step_length_mt(pcl_ctx* c, const double x[6], double dir[6], double step_init, double step_max, double step_min,
                                 double* score, double g[6], double H[36], double x_out[6]) {
    const double phi_0 = -*score;
    double d_phi_0 = 0.0;
    for (int k = 0; k < 6; ++k) d_phi_0 += g[k] * dir[k];
    d_phi_0 = -d_phi_0;
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) { memcpy(x_out, x, 6 * sizeof(double)); return 0; }
        d_phi_0 *= -1;
        for (int k = 0; k < 6; ++k) dir[k] *= -1;
    }
    const int max_step_iterations = 10;
    int step_iterations = 0;
    const double mu = 1.e-4, nu = 0.9;
    double a_l = 0, a_u = 0;
    double f_l = mt_psi(a_l, phi_0, phi_0, d_phi_0, mu), g_l = mt_dpsi(d_phi_0, d_phi_0, mu);
    double f_u = mt_psi(a_u, phi_0, phi_0, d_phi_0, mu), g_u = mt_dpsi(d_phi_0, d_phi_0, mu);
    int interval_converged = (step_max - step_min) < 0, open_interval = 1;
    double a_t = step_init;
    a_t = fmin(a_t, step_max);
    a_t = fmax(a_t, step_min);
    double x_t[6];
    for (int k = 0; k < 6; ++k) x_t[k] = x[k] + dir[k] * a_t;
    *score = pcl_derivs(c, x_t, 1, g, H);
    double phi_t = -*score, d_phi_t = 0.0;
    for (int k = 0; k < 6; ++k) d_phi_t += g[k] * dir[k];
    d_phi_t = -d_phi_t;
    double psi_t = mt_psi(a_t, phi_t, phi_0, d_phi_0, mu), d_psi_t = mt_dpsi(d_phi_t, d_phi_0, mu);
    while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        if (open_interval) a_t = mt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else a_t = mt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = fmin(a_t, step_max);
        a_t = fmax(a_t, step_min);
        for (int k = 0; k < 6; ++k) x_t[k] = x[k] + dir[k] * a_t;
        *score = pcl_derivs(c, x_t, 0, g, H);
        phi_t = -*score; d_phi_t = 0.0;
        for (int k = 0; k < 6; ++k) d_phi_t += g[k] * dir[k];
        d_phi_t = -d_phi_t;
        psi_t = mt_psi(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = mt_dpsi(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = 0;
            f_l += phi_0 - mu * d_phi_0 * a_l; g_l += mu * d_phi_0;
            f_u += phi_0 - mu * d_phi_0 * a_u; g_u += mu * d_phi_0;
        }
        if (open_interval) interval_converged = mt_update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
        else interval_converged = mt_update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
    }
    if (step_iterations) {       /* computeHessian: angular_jacobian_ of x_t, angular_hessian_ still of the first trial */
        ++c->evals;
        pcl_pass(c->h, c->d1, c->d2, x_t, c->aj, c->ah, 0, 1, NULL, H, NULL);
    }
    memcpy(x_out, x_t, sizeof x_t);
    return a_t;
}

int ora_pcl_align(ora_handle* h, const ndt_pcl_params* prm, const double guess[16], ndt_pcl_result* out) {
    memset(out, 0, sizeof *out);
    if (!h->n_vox) return out->status = NDT_ERR_NO_TARGET;
    if (!h->src) return out->status = NDT_ERR_NO_SOURCE;
    pcl_ctx c;
    memset(&c, 0, sizeof c);
    c.h = h; c.prm = prm;
    pcl_gauss(h->res_d, prm->outlier_ratio, &c.d1, &c.d2);
    double p[6], g[6], H[36];
    ora_pcl_vector_from_matrix(guess, p);
    double score = pcl_derivs(&c, p, 1, g, H);
    int converged = 0, iters = 0;
    while (!converged) {
        double ng[6], delta[6];
        for (int k = 0; k < 6; ++k) ng[k] = -g[k];
        pcl_svd_solve(H, ng, delta);
        double dn = 0.0;
        for (int k = 0; k < 6; ++k) dn += delta[k] * delta[k];
        dn = sqrt(dn);
        if (dn == 0 || isnan(dn)) { converged = dn == 0; break; }
        for (int k = 0; k < 6; ++k) delta[k] /= dn;
        double p_new[6];
        dn = pcl_step_length_mt(&c, p, delta, dn, prm->step_size, h->cfg.eps / 2, &score, g, H, p_new);
        for (int k = 0; k < 6; ++k) delta[k] *= dn;
        double Rd[3][3], td[3];
        pcl_rt_from_vector(delta, Rd, td);
        for (int k = 0; k < 6; ++k) p[k] += delta[k];
        const double cos_angle = 0.5 * ((Rd[0][0] + Rd[1][1] + Rd[2][2]) - 1.0);
        const double tsq = (td[0] * td[0] + td[1] * td[1]) + td[2] * td[2];
        ++iters;
        const double te = h->cfg.eps, re = prm->rotation_eps;
        if (iters >= h->cfg.max_iterations || ((te > 0 && tsq <= te) && (re > 0 && cos_angle >= re)) ||
            ((te <= 0) && (re > 0 && cos_angle >= re)) || ((te > 0 && tsq <= te) && (re <= 0)))
            converged = 1;
        (void)p_new;
    }
    out->iterations = iters; out->converged = converged; out->evaluations = c.evals;
    out->score = score; out->trans_probability = score / (double)h->n_src; out->n_neighbours = c.n_nb;
    memcpy(out->p, p, sizeof p); memcpy(out->gradient, g, sizeof g); memcpy(out->hessian, H, sizeof H);
    ora_pcl_matrix_from_vector(p, out->T);
    out->status = NDT_OK;
    return NDT_OK;
}
