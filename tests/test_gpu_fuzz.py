"""Randomised differential test: the HIP engine against the CPU oracle over seeded random scenes and configurations
(resolution, key rule, NEARBY1/7, voxel validity and gate thresholds, eigenvalue floor), with ragged input thrown in
(non-finite points, duplicates, points outside the key range, planar and collinear voxels, empty merges).  Same bars
as test_gpu_parity.py for one pass: keys, counts, validity, cell assignment and gate masks bit for bit; H, b, chi2 at 1e-10.
Aligned poses of the Gauss-Newton loop stay at the north star's 1e-4 m / 1e-5 rad here: random scenes with a dozen effective
points reach Hessians with condition numbers of 1e13, which turn a last-bit difference into micrometres (tools/fuzz_diag.py)."""
import numpy as np
import pytest

from conftest import pose_error, rel_err, TIGHT_H, TIGHT_T, TIGHT_R

pytestmark = pytest.mark.gpu


def _scene(rng, n, extent):
    """Points on a few random planes and boxes plus clutter, inside a box of `extent` metres around a random origin."""
    origin = rng.uniform(-200, 200, 3) * [1, 1, 0.05]
    parts = []
    for _ in range(rng.integers(3, 7)):
        m = n // 6
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        a = np.cross(nrm, [0.3, 0.5, 0.8]); a /= np.linalg.norm(a); b = np.cross(nrm, a)
        uv = rng.uniform(-extent / 2, extent / 2, (m, 2))
        parts.append(uv[:, :1] * a + uv[:, 1:] * b + rng.uniform(-extent / 4, extent / 4, 3) + rng.normal(0, 0.02, (m, 3)))
    parts.append(rng.uniform(-extent / 2, extent / 2, (n // 10, 3)))                        # clutter: sparse voxels below min_pts
    line = np.linspace(-extent / 3, extent / 3, n // 20)[:, None] * np.array([[1.0, 0.2, 0.0]])   # collinear voxels
    parts.append(line)
    return (np.concatenate(parts) + origin).astype(np.float32), origin


def _se3(rng, dt, dr):
    w = rng.normal(size=3); w *= dr / np.linalg.norm(w)
    th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / max(th, 1e-300)
    T = np.eye(4); T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    t = rng.normal(size=3); T[:3, 3] = t * dt / np.linalg.norm(t)
    return T


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_and_config(seed):
    import lio_slam_amd as la
    from oracle.oracle import Oracle
    rng = np.random.default_rng(1000 + seed)
    res = float(rng.choice([0.5, 0.8, 1.0, 1.5, 3.0]))
    cfg = dict(resolution=res, key_rule=int(rng.integers(0, 2)), nearby=int(rng.choice([1, 7])), min_pts_in_voxel=int(rng.integers(3, 9)),
               res_outlier_th=float(rng.choice([5.0, 20.0, 100.0])), eig_floor_ratio=float(rng.choice([1e-3, 1e-2, 5e-2])),
               max_iterations=int(rng.integers(3, 16)), eps=float(rng.choice([0.0, 1e-3, 1e-2])), min_effective_pts=int(rng.choice([10, 50])))
    pts, origin = _scene(rng, int(rng.integers(4000, 30000)), extent=float(rng.uniform(10, 40)) * res)
    pts = rng.permutation(pts)
    # ragged input: non-finite coordinates, exact duplicates, a point outside the packed-key range, one on a cell boundary
    bad = rng.choice(len(pts), 40, replace=False)
    pts[bad[:15], rng.integers(0, 3, 15)] = np.nan
    pts[bad[15:25], rng.integers(0, 3, 10)] = -np.inf
    pts[bad[25:35]] = pts[(bad[25:35] + 7) % len(pts)]
    pts[bad[35]] = [4e6 * res, 0, 0]
    pts[bad[36]] = np.float32(np.round(origin / res) * res)
    n = la.NDT(**cfg); o = Oracle(**cfg)
    cut = sorted(rng.choice(np.arange(1, len(pts)), 2, replace=False))
    n.set_target(pts[: cut[0]]); o.set_target(pts[: cut[0]])
    Tm = _se3(rng, 0.4, 0.02)
    n.add_target(pts[cut[0]: cut[1]]); o.add_target(pts[cut[0]: cut[1]])
    n.add_target(pts[cut[1]:], Tm); o.add_target(pts[cut[1]:], Tm)                   # a merge through a pose
    n.add_target(np.zeros((0, 3), np.float32)); o.add_target(np.zeros((0, 3), np.float32))
    vg, vo = n.export_voxels(), o.export_voxels()
    assert n.num_voxels() == o.num_voxels() == len(vo["keys"])
    assert np.array_equal(vg["keys"], vo["keys"]) and np.array_equal(vg["counts"], vo["counts"]) and np.array_equal(vg["valid"], vo["valid"])
    np.testing.assert_allclose(vg["mean"], vo["mean"], rtol=0, atol=1e-12 * max(1.0, res))
    assert np.max(np.abs(vg["cov"] - vo["cov"]) / np.maximum(np.abs(vo["cov"]).max(axis=(1, 2), keepdims=True), 1e-300)) < 1e-10
    # source: a subset of the scene seen from a perturbed pose
    T_true = _se3(rng, 2.0, 0.3); T_true[:3, 3] += origin
    sel = rng.choice(len(pts), min(len(pts), int(rng.integers(500, 6000))), replace=False)
    world = pts[sel].astype(np.float64)
    with np.errstate(invalid="ignore"):
        src = ((world - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)          # non-finite points stay non-finite
    guess = T_true @ _se3(rng, 0.15 * res, 0.01)
    n.set_source(src); o.set_source(src)
    kg, mg = n.source_cells(guess); ko, mo = o.source_cells(guess)
    fin = np.isfinite(src).all(axis=1)                                                # the oracle drops non-finite points
    assert np.all(kg[~fin] == np.iinfo(np.int64).min) and np.all(mg[~fin] == 0)
    assert np.array_equal(kg[fin], ko) and np.array_equal(mg[fin], mo)
    eg, eo = n.eval(guess), o.eval(guess)
    assert eg["n_effective"] == eo["n_effective"]
    if eo["n_effective"] > 0:
        assert rel_err(eg["H"], eo["H"]) < TIGHT_H and rel_err(eg["b"], eo["b"]) < 1e-8 and abs(eg["chi2"] - eo["chi2"]) <= TIGHT_H * abs(eo["chi2"])
    rg, ro = n.align(guess, check=False), o.align(guess)
    assert rg["rc"] == ro["rc"]
    if ro["rc"] == 0:
        dt, dr = pose_error(rg["T"], ro["T"])
        assert rg["iterations"] == ro["iterations"] and rg["converged"] == ro["converged"]
        assert dt < 1e-4 and dr < 1e-5, (dt, dr)
        assert rel_err(rg["H"], ro["H"]) < 1e-5
    # the same pairs as a replay batch (three hypotheses, two scans)
    if ro["rc"] == 0:
        half = len(src) // 2
        guesses = [guess, T_true, T_true @ _se3(rng, 0.1 * res, 0.005)]
        rb = n.align_batch([src[:half], src[half:]], guesses, [0, 1, 0])
        for p, (g, sc) in enumerate(zip(guesses, (src[:half], src[half:], src[:half]))):
            o.set_source(sc)
            rp = o.align(g)
            assert rb[p]["status"] == rp["rc"]
            if rp["rc"] == 0:
                dt, dr = pose_error(rb[p]["T"], rp["T"])
                assert dt < 1e-4 and dr < 1e-5 and rb[p]["iterations"] == rp["iterations"], (p, dt, dr)
    n.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_downsample(seed):
    """pcl::VoxelGrid on the device against the oracle: random leaf sizes and extrinsics, non-finite points, negative
    coordinates, from one point per cell to hundreds.  Same cells in the same (PCL index) order; centroids are float64
    sums in a different order, rounded once to float32: equal to 1 ulp, and almost always bit for bit."""
    import lio_slam_amd as la
    from oracle.oracle import voxel_downsample
    rng = np.random.default_rng(2000 + seed)
    leaf = float(rng.choice([0.1, 0.2, 0.5, 1.0, 2.0]))
    pts, origin = _scene(rng, int(rng.integers(2000, 60000)), extent=float(rng.uniform(20, 200)) * leaf)
    pts = rng.permutation(pts - origin.astype(np.float32))           # a scan in its sensor frame, both signs
    bad = rng.choice(len(pts), 30, replace=False)
    pts[bad[:20], rng.integers(0, 3, 20)] = np.nan
    pts[bad[20:], rng.integers(0, 3, 10)] = np.inf
    T = None if seed % 3 == 0 else _se3(rng, 1.5, 0.4)
    n = la.NDT()
    n.set_source_downsampled(pts, leaf, T)
    g = n.get_source()
    o = voxel_downsample(pts, leaf, T)
    assert g.shape == o.shape and len(o) > 100
    inv = np.float32(1.0) / np.float32(leaf)
    assert np.array_equal(np.floor(g * inv).astype(np.int32), np.floor(o * inv).astype(np.int32))
    assert np.max(np.abs(g.astype(np.float64) - o) / np.maximum(np.abs(o), 1.0)) < 2.5e-7
    assert np.mean(np.all(g == o, axis=1)) > 0.98
    n.set_source_downsampled(pts[:1], leaf, T)                        # one point: its own centroid
    assert n.source_size() == (1 if np.isfinite(pts[0]).all() else 0)
    n.close()


@pytest.mark.parametrize("seed", range(40))
def test_random_pcl_mode(seed):
    """PCL compatibility mode against the oracle's PCL-literal restatement on random scenes: resolutions, outlier ratios,
    step sizes, both forms of the angular Hessian row, merges before the evaluation, non-finite source points."""
    import lio_slam_amd as la
    from oracle.oracle import Oracle
    rng = np.random.default_rng(3000 + seed)
    res = float(rng.choice([0.5, 1.0, 2.0]))
    cfg = dict(resolution=res, max_iterations=int(rng.integers(2, 12)), eps=float(rng.choice([1e-2, 1e-4, 1e-6])))
    prm = dict(outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])), step_size=float(rng.choice([0.05, 0.1, 0.5])) * res,
               hessian_d1_as_pcl=int(rng.integers(0, 2)))
    pts, origin = _scene(rng, int(rng.integers(6000, 20000)), extent=float(rng.uniform(10, 25)) * res)
    pts = rng.permutation(pts - origin.astype(np.float32))           # PCL mode works in the map frame: keep the angles' lever arm short
    n = la.NDT(**cfg); o = Oracle(**cfg)
    h = len(pts) // 2
    n.set_target(pts[:h]); o.set_target(pts[:h]); n.add_target(pts[h:]); o.add_target(pts[h:])
    T_true = _se3(rng, 1.0, 0.2)
    sel = rng.choice(len(pts), int(rng.integers(800, 4000)), replace=False)
    src = ((pts[sel].astype(np.float64) - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
    src[3] = np.nan; src[11, 2] = np.inf
    guess = T_true @ _se3(rng, 0.1 * res, 0.01)
    n.set_source(src); o.set_source(src)
    p = la.pcl_vector_from_matrix(guess)
    rn, ro = n.pcl_eval(p, **prm), o.pcl_eval(p, **prm)
    assert rn["n_neighbours"] == ro["n_neighbours"] > 0
    assert abs(rn["score"] - ro["score"]) <= TIGHT_H * abs(ro["score"])
    assert rel_err(rn["g"], ro["g"]) < 1e-8 and rel_err(rn["H"], ro["H"]) < TIGHT_H
    ra, rb = n.pcl_align(guess, **prm), o.pcl_align(guess, **prm)
    assert ra["rc"] == rb["rc"] == 0 and ra["iterations"] == rb["iterations"] and ra["evaluations"] == rb["evaluations"]
    dt, dr = pose_error(ra["T"], rb["T"])
    assert dt < 1e-8 and dr < 1e-8, (dt, dr)
    n.close()
