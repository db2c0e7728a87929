#!/usr/bin/env python3
"""Generates tests/golden/*.npz — float64 NumPy known-answer vectors for the NDT path.

The reference (ergevv/lio-slam) ships no tests, fixtures or golden vectors for
this path and its NDT implementation (PCL) is absent, so nothing can be generated
from the reference itself.  These vectors come from this
independent NumPy implementation of the published algorithm (np.cov, np.linalg.eigh,
np.linalg.solve, dict-based voxel map — no code shared with oracle/ndt_oracle.c or the
HIP engine) and pin the oracle.  Run:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from lio_slam_amd import synth  # noqa: E402  (scene generator only; no NDT arithmetic)

NB7 = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]


# ------------------------------------------------------------------ (i) key rules
def target_keys(p32, res, rule):
    p32 = np.asarray(p32, np.float32)
    inv32 = np.float32(1.0) / np.float32(res)
    if rule == 0:   # PCL: floor(p_f32 * inv_leaf_f32) evaluated in float32
        return np.floor(p32 * inv32).astype(np.int32)
    v = p32.astype(np.float64) * np.float64(inv32)   # SLAM-book: round half away from zero
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int32)


def source_keys(q64, res, rule):
    inv = np.float64(np.float32(1.0) / np.float32(res))
    v = np.asarray(q64, np.float64) * inv
    if rule == 0:
        return np.floor(v).astype(np.int32)
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int32)


def gen_keys():
    one_m = np.nextafter(np.float32(1.0), np.float32(0.0))
    vals = [0.0, -0.0, 0.5, -0.5, 1.0, -1.0, -1e-7, 1e-7, float(one_m), -float(one_m), 1.5, -1.5, 2.5, -2.5,
            0.49999997, -0.49999997, 0.50000006, 999.99994, -999.99994, 1000.0, -1000.0, 1000.5, -1000.5,
            3.0000002, -3.0000002, 7.9999995, -7.9999995, 0.25, -0.25, 0.75, -0.75, 123.456, -123.456,
            1e-30, -1e-30, 65535.996, -65535.996, 0.3, 0.6, 0.9, 1.2, -0.3, -0.6, -0.9, -1.2, 2.0, -2.0,
            10.5, -10.5, 99.5, -99.5, 0.1, -0.1, 0.2, -0.2, 0.7, -0.7, 4.5, -4.5, 5.4999995, -5.4999995,
            12.000001, -12.000001, 31.5]
    v32 = np.array(vals, np.float32)
    pts = np.stack([v32, np.roll(v32, 1), np.roll(v32, 2)], 1)
    out = {"points": pts}
    for res in (1.0, 0.5, 0.3):
        for rule in (0, 1):
            out[f"target_r{res}_k{rule}"] = target_keys(pts, res, rule)
            q = pts.astype(np.float64) + 1e-9      # float64 values that are not float32-representable
            out[f"source_r{res}_k{rule}"] = source_keys(q, res, rule)
    out["source_points"] = pts.astype(np.float64) + 1e-9
    np.savez(os.path.join(HERE, "keys.npz"), **out)


# ------------------------------------------------------------------ (ii)/(iii) voxel statistics
def reg_info(cov, ratio):
    lam, V = np.linalg.eigh(cov)
    lam = np.maximum(lam, ratio * lam.max())
    return (V * lam) @ V.T, (V / lam) @ V.T


def gen_voxels():
    rng = np.random.Generator(np.random.PCG64(7))
    cells = [(3, -2, 0), (-5, 4, 1), (100, -100, 2), (0, 0, 0)]
    counts = [6, 7, 50, 5]
    pts, exp = [], {}
    for ci, (cell, n) in enumerate(zip(cells, counts)):
        base = np.array(cell, np.float64)
        # a thin slab (planar surface) inside the cell -> exercises the eigenvalue floor
        p = base + np.array([0.5, 0.5, 0.3]) + rng.normal(size=(n, 3)) * np.array([0.2, 0.15, 0.004])
        p = np.clip(p, base + 0.01, base + 0.99).astype(np.float32)
        pts.append(p)
        p64 = p.astype(np.float64)
        exp[f"n{ci}"] = n
        exp[f"mean{ci}"] = p64.mean(0)
        exp[f"cov{ci}"] = np.cov(p64.T)              # sample covariance /(n-1), math_utils.h:55-72
        creg, info = reg_info(exp[f"cov{ci}"], 0.01)
        exp[f"covreg{ci}"], exp[f"info{ci}"] = creg, info
        exp[f"cell{ci}"] = np.array(cell, np.int32)
    exp["points"] = np.concatenate(pts)
    exp["counts"] = np.array(counts, np.int32)
    # (iii) merge: second batch into cells 0..2, UpdateMeanAndCov math_utils.h:87-98
    pts2 = []
    for ci, (cell, n2) in enumerate(zip(cells[:3], [9, 1, 20])):
        base = np.array(cell, np.float64)
        p = base + np.array([0.4, 0.6, 0.32]) + rng.normal(size=(n2, 3)) * np.array([0.15, 0.2, 0.006])
        p = np.clip(p, base + 0.01, base + 0.99).astype(np.float32)
        pts2.append(p)
        p64 = p.astype(np.float64)
        m, n = counts[ci], n2
        mu_h, cov_h = exp[f"mean{ci}"], exp[f"cov{ci}"]
        mu_c = p64.mean(0)
        cov_c = np.cov(p64.T) if n2 > 1 else np.zeros((3, 3))
        mu_n = (m * mu_h + n * mu_c) / (m + n)
        cov_n = (m * (cov_h + np.outer(mu_h - mu_n, mu_h - mu_n)) +
                 n * (cov_c + np.outer(mu_c - mu_n, mu_c - mu_n))) / (m + n)
        exp[f"merged_n{ci}"] = m + n
        exp[f"merged_mean{ci}"] = mu_n
        exp[f"merged_cov{ci}"] = cov_n
        exp[f"merged_info{ci}"] = reg_info(cov_n, 0.01)[1]
    exp["points2"] = np.concatenate(pts2)
    np.savez(os.path.join(HERE, "voxels.npz"), **exp)


# ------------------------------------------------------------------ (iv) scan pair: H, b, chi2, poses
def build_map(pts32, res=1.0, min_pts=6, ratio=0.01, quantize=True):
    keys = target_keys(pts32, res, 0)
    vox = {}
    order = np.lexsort((np.arange(len(keys)), keys[:, 2], keys[:, 1], keys[:, 0]))
    ks = keys[order]
    bounds = np.flatnonzero(np.any(np.diff(ks, axis=0) != 0, axis=1)) + 1
    for seg in np.split(order, bounds):
        k = tuple(int(v) for v in keys[seg[0]])
        p = pts32[seg].astype(np.float64)
        n = len(seg)
        mean = p.mean(0)
        off = mean - np.array(k, np.float64) * res
        valid = False
        info = np.zeros((3, 3))
        if n >= min_pts:
            cov = np.cov(p.T)
            lam = np.linalg.eigvalsh(cov)
            if lam.max() > 0:
                info = reg_info(cov, ratio)[1]
                valid = True
        if quantize:
            off = off.astype(np.float32).astype(np.float64)
            info = info.astype(np.float32).astype(np.float64)
        vox[k] = (off, info, valid)
    return vox


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def evaluate(vox, src32, T, nearby, res=1.0, th=20.0):
    R, t = T[:3, :3], T[:3, 3]
    s = src32.astype(np.float64)
    q = np.empty_like(s)
    for j in range(3):   # the engine's operation order: ((R0 x + R1 y) + R2 z) + t
        q[:, j] = ((R[j, 0] * s[:, 0] + R[j, 1] * s[:, 1]) + R[j, 2] * s[:, 2]) + t[j]
    cells = source_keys(q, res, 0)
    H = np.zeros((6, 6)); b = np.zeros(6); chi2s = 0.0; neff = 0
    masks = np.zeros(len(s), np.uint8)
    for i in range(len(s)):
        J = np.hstack([-R @ hat(s[i]), np.eye(3)])
        for pr in range(nearby):
            k = (int(cells[i, 0]) + NB7[pr][0], int(cells[i, 1]) + NB7[pr][1], int(cells[i, 2]) + NB7[pr][2])
            v = vox.get(k)
            if v is None or not v[2]:
                continue
            e = (q[i] - np.array(k, np.float64) * res) - v[0]
            c2 = e @ v[1] @ e
            if np.isnan(c2) or c2 > th:
                continue
            masks[i] |= 1 << pr
            H += J.T @ v[1] @ J
            b += -J.T @ v[1] @ e
            chi2s += c2
            neff += 1
    return H, b, chi2s, neff, cells, masks


def exp_so3(w):
    th = np.linalg.norm(w)
    if th <= 1e-7:
        return np.eye(3)
    K = hat(w / th)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def gen_pair():
    rng = np.random.Generator(np.random.PCG64(11))
    scene = synth.room_scene(rng)
    Tm = synth.se3([0, 0, 0.0], [0.0, 0.0, 0.6])
    Ts = synth.se3([0.01, -0.02, 0.05], [0.35, -0.1, 0.62])
    m = synth.transform(Tm, synth.scan(scene, "vlp16", Tm, rng))
    s = synth.scan(scene, "vlp16", Ts, rng)
    m = m[np.sort(rng.choice(len(m), 6000, replace=False))]
    s = s[np.sort(rng.choice(len(s), 2000, replace=False))]
    guess = Ts @ synth.se3([0.01, 0.02, -0.03], [0.15, -0.1, 0.05])
    out = {"map": m, "scan": s, "guess": guess, "T_true": Ts}
    for quant in (1, 0):
        vox = build_map(m, quantize=bool(quant))
        for nearby in (1, 7):
            H, b, c2, ne, cells, masks = evaluate(vox, s, guess, nearby)
            tag = f"q{quant}_n{nearby}"
            out[f"H_{tag}"], out[f"b_{tag}"], out[f"chi2_{tag}"], out[f"neff_{tag}"] = H, b, c2, ne
            out[f"cells_{tag}"], out[f"masks_{tag}"] = cells, masks
        # Gauss-Newton trajectory, NEARBY7, 6 iterations
        T = guess.copy()
        poses = []
        for _ in range(6):
            H, b, _, _, _, _ = evaluate(vox, s, T, 7)
            dx = np.linalg.solve(H, b)
            T = T.copy()
            T[:3, :3] = T[:3, :3] @ exp_so3(dx[:3])
            T[:3, 3] = T[:3, 3] + dx[3:]
            poses.append(T)
        out[f"poses_q{quant}"] = np.stack(poses)
    np.savez_compressed(os.path.join(HERE, "pair.npz"), **out)


# ------------------------------------------------------------------ (vii) configs[1] at full size: 30k-pt scan vs 200k-pt map
def gen_config1():
    """BASELINE.json configs[1] as synth.make_config(1) generates it (the inputs are regenerated by the tests from the
    seed; a checksum guards against drift): every voxel's key and count, the regularised information matrix of the
    first 1500 valid voxels (np.linalg.eigh), H / b / chi2 / n_eff at the guess for NEARBY7, and the same sums over
    the two halves of the source (what two ranks of the sharded mode contribute)."""
    c = synth.make_config(1)
    m, s, guess = c["map"], c["scan"], c["guess"]
    vox = build_map(m, quantize=True)
    keys = np.array(sorted(vox.keys()), np.int32)
    k3 = target_keys(m, 1.0, 0)
    uk, cnt = np.unique(k3, axis=0, return_counts=True)
    assert np.array_equal(uk.astype(np.int32), keys)
    valid = np.array([vox[tuple(k)][2] for k in keys], np.uint8)
    sel = np.flatnonzero(valid)[:1500]
    info = np.stack([vox[tuple(keys[i])][1] for i in sel])
    off = np.stack([vox[tuple(keys[i])][0] for i in sel])
    out = {"keys": keys, "counts": cnt.astype(np.int32), "valid": valid, "info_idx": sel.astype(np.int32), "info": info, "offset": off,
           "input_checksum": np.array([float(m.astype(np.float64).sum()), float(s.astype(np.float64).sum()), float(guess.sum())])}
    H, b, c2, ne, _, masks = evaluate(vox, s, guess, 7)
    out.update(H=H, b=b, chi2=c2, neff=ne, masks=masks)
    h = len(s) // 2
    for tag, part in (("a", s[:h]), ("b", s[h:])):
        Hs, bs, c2s, nes, _, _ = evaluate(vox, part, guess, 7)
        out.update({f"H_{tag}": Hs, f"b_{tag}": bs, f"chi2_{tag}": c2s, f"neff_{tag}": nes})
    np.savez_compressed(os.path.join(HERE, "config1.npz"), **out)


def gen_yard():
    """The 200k-point yard map of synth.make_config(2, small=True) (thousands of voxels): every key and count, and the
    regularised information matrix (np.linalg.eigh) + mean offset of 1500 valid voxels spread over the map."""
    m = synth.make_config(2, small=True)["map"]
    vox = build_map(m, quantize=True)
    keys = np.array(sorted(vox.keys()), np.int32)
    uk, cnt = np.unique(target_keys(m, 1.0, 0), axis=0, return_counts=True)
    assert np.array_equal(uk.astype(np.int32), keys)
    valid = np.array([vox[tuple(k)][2] for k in keys], np.uint8)
    vi = np.flatnonzero(valid)
    sel = vi[np.linspace(0, len(vi) - 1, 1500).astype(int)]
    np.savez_compressed(os.path.join(HERE, "yard.npz"), keys=keys, counts=cnt.astype(np.int32), valid=valid, info_idx=sel.astype(np.int32),
                        info=np.stack([vox[tuple(keys[i])][1] for i in sel]), offset=np.stack([vox[tuple(keys[i])][0] for i in sel]),
                        input_checksum=np.array([float(m.astype(np.float64).sum())]))


# ------------------------------------------------------------------ (vi) VoxelGrid centroid downsample (§8(f) N1)
def gen_downsample():
    rng = np.random.Generator(np.random.PCG64(13))
    scene = synth.room_scene(rng)
    Ts = synth.se3([0.0, 0.0, 0.3], [1.0, 0.5, 0.6])
    s = synth.scan(scene, "vlp16", Ts, rng)
    s = s[np.sort(rng.choice(len(s), 6000, replace=False))]
    s[17] = [np.nan, 1, 1]; s[99] = [0, np.inf, 0]
    T_il = synth.se3([0.01, -0.02, 0.03], [0.05, 0.0, 0.28])           # a LiDAR -> IMU extrinsic
    out = {"scan": s, "T": T_il}
    for leaf in (0.5, 0.3):
        fin = np.isfinite(s).all(axis=1)
        p = synth.transform(T_il, s[fin])                                  # float64 math, one rounding to float32
        cells = np.floor(p * (np.float32(1.0) / np.float32(leaf))).astype(np.int64)   # PCL: float32 multiply, floor
        order = np.lexsort((np.arange(len(p)), cells[:, 0], cells[:, 1], cells[:, 2]))   # z slowest, x fastest
        cs = cells[order]
        bounds = np.flatnonzero(np.any(np.diff(cs, axis=0) != 0, axis=1)) + 1
        out[f"centroids_{leaf}"] = np.array([p[g].astype(np.float64).mean(0) for g in np.split(order, bounds)])   # float64
        out[f"cells_{leaf}"] = np.array([cells[g[0]] for g in np.split(order, bounds)], np.int32)
    np.savez_compressed(os.path.join(HERE, "downsample.npz"), **out)


# ------------------------------------------------------------------ (v) PCL compatibility mode (§8(f) N4)
def pcl_vector_from_matrix(T):
    """Eigen's eulerAngles(0,1,2) (first angle in [0, pi]) + translation -> [t, angles]."""
    m = T[:3, :3]
    r0 = np.arctan2(m[1, 2], m[2, 2])
    c2 = np.hypot(m[0, 0], m[0, 1])
    if r0 > 0:
        r0 -= np.pi
        r1 = np.arctan2(-m[0, 2], -c2)
    else:
        r1 = np.arctan2(-m[0, 2], c2)
    s1, c1 = np.sin(r0), np.cos(r0)
    r2 = np.arctan2(s1 * m[2, 0] - c1 * m[1, 0], c1 * m[1, 1] - s1 * m[2, 1])
    return np.concatenate([T[:3, 3], -np.array([r0, r1, r2])])


def rot_xyz(a):
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def pcl_derivatives(vox, src32, p, res=1.0, outlier_ratio=0.55, d1_as_pcl=True, float32_transform=True):
    """score, gradient, Hessian of pcl::NormalDistributionsTransform::computeDerivatives:
    radius search over the centroids of the valid voxels with a kd-tree, Magnusson eq. 6.9-6.13 and 6.17-6.21."""
    from scipy.spatial import cKDTree
    c1 = 10 * (1 - outlier_ratio); c2 = outlier_ratio / res ** 3; d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    keys = [k for k, v in vox.items() if v[2]]
    means = np.array([np.array(k, np.float64) * res + vox[k][0] for k in keys])
    infos = np.array([vox[k][1] for k in keys])
    R = rot_xyz(p[3:])
    if float32_transform:
        M = np.hstack([R, p[:3, None]]).astype(np.float32)
        x32 = src32.astype(np.float32)
        q = (((M[:, 0] * x32[:, :1] + M[:, 1] * x32[:, 1:2]) + M[:, 2] * x32[:, 2:3]) + M[:, 3]).astype(np.float64)
        tree = cKDTree(means.astype(np.float32).astype(np.float64))
    else:
        q = src32.astype(np.float64) @ R.T + p[:3]
        tree = cKDTree(means)
    cs = [1.0 if abs(a) < 10e-5 else np.cos(a) for a in p[3:]]
    sn = [0.0 if abs(a) < 10e-5 else np.sin(a) for a in p[3:]]
    (cx, cy, cz), (sx, sy, sz) = cs, sn
    aj = np.array([[-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy], [cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy],
                   [-sy * cz, sy * sz, cy], [sx * cy * cz, -sx * cy * sz, sx * sy], [-cx * cy * cz, cx * cy * sz, -cx * sy],
                   [-cy * sz, -cy * cz, 0], [cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0], [sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0]])
    ah = np.array([[-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy], [-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy],
                   [cx * cy * cz, -cx * cy * sz, cx * sy], [sx * cy * cz, -sx * cy * sz, sx * sy],
                   [-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0], [cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0],
                   [-cy * cz, cy * sz, sy if d1_as_pcl else -sy], [-sx * sy * cz, sx * sy * sz, sx * cy], [cx * sy * cz, -cx * sy * sz, -cx * cy],
                   [sy * sz, sy * cz, 0], [-sx * cy * sz, -sx * cy * cz, 0], [cx * cy * sz, cx * cy * cz, 0],
                   [-cy * cz, cy * sz, 0], [-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0], [-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0]])
    score = 0.0; g = np.zeros(6); H = np.zeros((6, 6)); nb = 0; margin = np.inf
    x64 = src32.astype(np.float64)
    for i, idxs in enumerate(tree.query_ball_point(q, res)):
        if not idxs:
            continue
        xa = aj @ x64[i]; xh = ah @ x64[i]
        J = np.zeros((3, 6)); J[:, :3] = np.eye(3)
        J[1, 3], J[2, 3], J[0, 4], J[1, 4], J[2, 4], J[0, 5], J[1, 5], J[2, 5] = xa
        HE = np.zeros((6, 6, 3))
        a, b, c = np.array([0, xh[0], xh[1]]), np.array([0, xh[2], xh[3]]), np.array([0, xh[4], xh[5]])
        d, e, f = xh[6:9], xh[9:12], xh[12:15]
        HE[3, 3], HE[4, 3], HE[5, 3] = a, b, c
        HE[3, 4], HE[4, 4], HE[5, 4] = b, d, e
        HE[3, 5], HE[4, 5], HE[5, 5] = c, e, f
        for k in idxs:
            dist = np.linalg.norm(q[i] - tree.data[k])
            margin = min(margin, abs(dist - res))
            nb += 1
            xt = q[i] - means[k]
            Ci = infos[k]
            ex = np.exp(-d2 * (xt @ Ci @ xt) / 2)
            w = d2 * ex
            if w > 1 or w < 0 or np.isnan(w):
                continue
            score += -d1 * ex
            w *= d1
            xcJ = xt @ Ci @ J
            g += xcJ * w
            H += w * (-d2 * np.outer(xcJ, xcJ) + np.einsum("r,abr->ab", Ci @ xt, HE) + J.T @ Ci @ J)
    return score, g, H, nb, margin


def gen_pcl():
    pair = np.load(os.path.join(HERE, "pair.npz"))
    m, s, guess = pair["map"], pair["scan"], pair["guess"]
    vox = build_map(m, quantize=True)
    p = pcl_vector_from_matrix(guess)
    assert np.allclose(rot_xyz(p[3:]), guess[:3, :3], atol=1e-12)
    out = {"p": p, "guess": guess}
    # formula check in float64 (no float32 rounding noise): analytic gradient / Hessian (eq. 6.21 sign) against central
    # differences of the score / gradient.  Neighbour sets are frozen by the tiny step.
    h = 1e-6
    _, g0, H0, _, _ = pcl_derivatives(vox, s, p, d1_as_pcl=False, float32_transform=False)
    gn = np.zeros(6); Hn = np.zeros((6, 6))
    for k in range(6):
        dp = np.zeros(6); dp[k] = h
        sp, gp, _, _, _ = pcl_derivatives(vox, s, p + dp, d1_as_pcl=False, float32_transform=False)
        sm, gm, _, _, _ = pcl_derivatives(vox, s, p - dp, d1_as_pcl=False, float32_transform=False)
        gn[k] = (sp - sm) / (2 * h); Hn[:, k] = (gp - gm) / (2 * h)
    assert np.allclose(gn, g0, rtol=1e-5, atol=1e-5 * np.abs(g0).max()), (gn, g0)
    assert np.allclose(Hn, H0, rtol=1e-4, atol=1e-5 * np.abs(H0).max()), (Hn - H0)
    for tag, flag in (("pcl", True), ("eq621", False)):
        sc, g, H, nb, margin = pcl_derivatives(vox, s, p, d1_as_pcl=flag)
        assert margin > 1e-5, margin       # no neighbour sits on the radius: float32 / float64 distance tests agree
        out[f"score_{tag}"], out[f"g_{tag}"], out[f"H_{tag}"], out[f"nb_{tag}"] = sc, g, H, nb
    # the same against float64 voxel payloads (no float32 rounding of the table: pins the oracle to 1e-9)
    vox0 = build_map(m, quantize=False)
    sc, g, H, nb, margin = pcl_derivatives(vox0, s, p, d1_as_pcl=True)
    assert margin > 1e-5, margin
    out["score_q0"], out["g_q0"], out["H_q0"], out["nb_q0"] = sc, g, H, nb
    # a second transform vector with large angles (exercises every angular term)
    p2 = p + np.array([0.3, -0.2, 0.1, 0.4, -0.3, 0.5])
    sc, g, H, nb, margin = pcl_derivatives(vox, s, p2, d1_as_pcl=True)
    assert margin > 1e-5, margin
    out["p2"], out["score2"], out["g2"], out["H2"], out["nb2"] = p2, sc, g, H, nb
    np.savez_compressed(os.path.join(HERE, "pcl.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pcl":
        gen_pcl()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "config1":
        gen_config1()
        gen_yard()
        sys.exit(0)
    gen_downsample()
    gen_keys()
    gen_voxels()
    gen_pair()
    gen_pcl()
    gen_config1()
    gen_yard()
    print("golden vectors written to", HERE)
