"""Pins the CPU oracle (oracle/ndt_oracle.c) to the independent NumPy float64 golden vectors.

The reference holds no fixtures for this path: parity with the
reference itself is unpinned; these vectors are the pin.
"""
import numpy as np
import pytest

from conftest import pose_error, rel_err
from oracle.oracle import Oracle


def _unpack(keys):
    b = 1 << 20
    return np.stack([((keys >> 42) & 0x1FFFFF) - b, ((keys >> 21) & 0x1FFFFF) - b, (keys & 0x1FFFFF) - b], 1)


@pytest.mark.parametrize("res", [1.0, 0.5, 0.3])
@pytest.mark.parametrize("rule", [0, 1])
def test_key_rules(golden, res, rule):
    g = golden["keys"]
    o = Oracle(resolution=res, key_rule=rule)
    for p, q, kt, ks in zip(g["points"], g["source_points"], g[f"target_r{res}_k{rule}"], g[f"source_r{res}_k{rule}"]):
        assert np.array_equal(o.target_key(p), kt), (p, res, rule)
        assert np.array_equal(o.source_key(q), ks), (q, res, rule)


def test_voxel_statistics(golden):
    g = golden["voxels"]
    o = Oracle(quantize=False)
    o.set_target(g["points"])
    v = o.export_voxels()
    cells = _unpack(v["keys"])
    assert len(cells) == 4
    for ci in range(4):
        i = int(np.flatnonzero(np.all(cells == g[f"cell{ci}"], axis=1))[0])
        assert v["counts"][i] == g[f"n{ci}"]
        np.testing.assert_allclose(v["mean"][i], g[f"mean{ci}"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(v["cov"][i], g[f"cov{ci}"], rtol=1e-9, atol=1e-15)
        if g[f"n{ci}"] >= 6:
            assert v["valid"][i] == 1
            assert rel_err(v["info"][i], g[f"info{ci}"]) < 1e-9
        else:
            assert v["valid"][i] == 0 and not v["info"][i].any()


def test_incremental_merge(golden):
    g = golden["voxels"]
    o = Oracle(quantize=False)
    o.set_target(g["points"])
    o.add_target(g["points2"])
    v = o.export_voxels()
    cells = _unpack(v["keys"])
    for ci in range(3):
        i = int(np.flatnonzero(np.all(cells == g[f"cell{ci}"], axis=1))[0])
        assert v["counts"][i] == g[f"merged_n{ci}"]
        np.testing.assert_allclose(v["mean"][i], g[f"merged_mean{ci}"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(v["cov"][i], g[f"merged_cov{ci}"], rtol=1e-9, atol=1e-15)
        assert rel_err(v["info"][i], g[f"merged_info{ci}"]) < 1e-9


@pytest.mark.parametrize("quant", [1, 0])
@pytest.mark.parametrize("nearby", [1, 7])
def test_pair_eval(golden, quant, nearby):
    g = golden["pair"]
    o = Oracle(quantize=bool(quant), nearby=nearby)
    o.set_target(g["map"])
    o.set_source(g["scan"])
    tag = f"q{quant}_n{nearby}"
    keys, masks = o.source_cells(g["guess"])
    assert np.array_equal(_unpack(keys), g[f"cells_{tag}"])        # bit-exact cell assignment
    assert np.array_equal(masks, g[f"masks_{tag}"])
    e = o.eval(g["guess"])
    assert e["n_effective"] == int(g[f"neff_{tag}"])
    # measured 2e-15 in both modes.  (Until round 3 the quantised mode needed 1e-6 here: gcc had dropped the float32 rounding
    # of two of the three mean offsets in the oracle, see f32_round in ndt_oracle.c.)  A last-bit difference between two
    # float64 eigen-solvers can still flip one float32 rounding of one voxel's information entry: 6e-8 of one term in thousands
    tol = 1e-10 if quant else 1e-12
    assert rel_err(e["H"], g[f"H_{tag}"]) < tol
    assert rel_err(e["b"], g[f"b_{tag}"]) < tol
    assert abs(e["chi2"] - float(g[f"chi2_{tag}"])) < tol * float(g[f"chi2_{tag}"])


@pytest.mark.parametrize("quant", [1, 0])
def test_pair_gauss_newton_poses(golden, quant):
    g = golden["pair"]
    for it in range(1, 7):
        o = Oracle(quantize=bool(quant), nearby=7, max_iterations=it, eps=0.0, final_eval=0)
        o.set_target(g["map"])
        o.set_source(g["scan"])
        r = o.align(g["guess"])
        assert r["rc"] == 0 and r["iterations"] == it
        dt, dr = pose_error(r["T"], g[f"poses_q{quant}"][it - 1])
        assert dt < 1e-10 and dr < 1e-7, (it, dt, dr)  # measured 6e-16 m; arccos floor ~1.5e-8 near identity
    t_err, r_err = pose_error(r["T"], g["T_true"])
    assert t_err < 0.06 and r_err < 0.01       # sparse 2k-point scan: NDT bias, not a parity bar


def test_payload_rounding_is_applied():
    """quantize = 1 means: mean offsets and information entries rounded to float32, all of them.  One voxel, one source point,
    chi^2 at the identity against NumPy with the rounding done here; the unrounded offsets give a measurably different value,
    so a rounding that the compiler dropped (as gcc 11.4 did for the x and y offsets until round 3) fails this test."""
    rng = np.random.default_rng(17)
    pts = (np.array([3.0, -2.0, 1.0]) + 0.5 + rng.normal(0, [0.11, 0.07, 0.03], (40, 3))).astype(np.float32)
    pts = pts[np.all(np.floor(pts) == [3, -2, 1], axis=1)]
    q = np.array([[3.61, -1.37, 1.52]], np.float32)
    chi = {}
    for quant in (0, 1):
        o = Oracle(quantize=bool(quant), nearby=1, res_outlier_th=1e9, min_effective_pts=1)
        o.set_target(pts); o.set_source(q)
        e = o.eval(np.eye(4))
        assert e["n_effective"] == 1
        chi[quant] = e["chi2"]
    v = Oracle(quantize=False); v.set_target(pts); vox = v.export_voxels()
    mean, info = vox["mean"][0], vox["info"][0]
    off32 = (mean - [3, -2, 1]).astype(np.float32).astype(np.float64)
    info32 = info.astype(np.float32).astype(np.float64)
    assert np.all(off32 != mean - [3, -2, 1])                       # none of the three offsets is a float32 number
    e1 = (q[0].astype(np.float64) - [3, -2, 1]) - off32
    e0 = q[0].astype(np.float64) - mean
    assert abs(chi[1] - e1 @ info32 @ e1) < 1e-12 * chi[1]
    assert abs(chi[0] - e0 @ info @ e0) < 1e-12 * chi[0]
    for j in range(3):                                               # each offset's rounding is visible at this tolerance
        ej = e1.copy(); ej[j] = e0[j]
        assert abs(ej @ info32 @ ej - chi[1]) > 1e-10 * chi[1]


def test_edge_cases():
    o = Oracle()
    with pytest.raises(RuntimeError):
        o.eval(np.eye(4))                                         # no target
    pts = np.zeros((5, 3), np.float32)
    pts[1] = [np.nan, 0, 0]; pts[2] = [0, np.inf, 0]
    o.set_target(pts)
    assert o.num_voxels() == 1 and o.export_voxels()["counts"][0] == 3   # non-finite points skipped
    o.set_source(pts)
    assert o.source_size() == 3
    r = o.align(np.eye(4))
    assert r["rc"] == -5                                          # too few effective points
    o.set_source(np.zeros((0, 3), np.float32))
    assert o.source_size() == 0


@pytest.mark.parametrize("leaf", [0.5, 0.3])
def test_voxel_downsample(golden, leaf):
    """§8(f) N1: pcl::transformPointCloud + pcl::VoxelGrid centroids (process.cc:62-73), xyz only."""
    from oracle.oracle import voxel_downsample
    g = golden["downsample"]
    d = voxel_downsample(g["scan"], leaf, g["T"])
    ref = g[f"centroids_{leaf}"]
    assert d.shape == ref.shape                                              # same cells, same (PCL index) order
    assert np.max(np.abs(d.astype(np.float64) - ref)) < 1e-6                 # float32 rounding of a float64 mean
    assert np.array_equal(np.floor(d * (np.float32(1.0) / np.float32(leaf))).astype(np.int32)[:, 2].argsort(kind="stable"),
                          np.arange(len(d)))                                 # z is the slowest index


def _check_map_against_golden(v, g):
    """keys, counts and validity of every voxel bit for bit; the float32 payload (mean offset, regularised information)
    of the golden subset: the golden values are float64 results rounded to float32, so they must agree to a float32 ulp."""
    cells = _unpack(v["keys"])
    assert np.array_equal(cells, g["keys"]) and np.array_equal(v["counts"], g["counts"]) and np.array_equal(v["valid"], g["valid"])
    idx = g["info_idx"]
    scale = np.abs(g["info"]).max(axis=(1, 2), keepdims=True)
    assert np.max(np.abs(v["info"][idx] - g["info"]) / scale) < 3e-7
    off = v["mean"][idx] - cells[idx].astype(np.float64)
    assert np.max(np.abs(off - g["offset"])) < 2e-7


def test_config1_full_size_map_and_sums(golden):
    """BASELINE.json configs[1] at its full size (30k-point scan, 200k-point map): the oracle against the NumPy vectors —
    every voxel, H / b / chi2 / n_eff and the gate masks at the guess, and the sums of the two halves of the source
    (what two ranks contribute in the sharded mode) adding up to the whole."""
    from lio_slam_amd import synth
    g = golden["config1"]
    c = synth.make_config(1)
    chk = np.array([float(c["map"].astype(np.float64).sum()), float(c["scan"].astype(np.float64).sum()), float(c["guess"].sum())])
    assert np.array_equal(chk, g["input_checksum"]), "synthetic inputs drifted: regenerate tests/golden/config1.npz"
    o = Oracle()
    o.set_target(c["map"])
    _check_map_against_golden(o.export_voxels(), g)
    o.set_source(c["scan"])
    e = o.eval(c["guess"])
    _, masks = o.source_cells(c["guess"])
    assert np.array_equal(masks, g["masks"]) and e["n_effective"] == int(g["neff"])
    # measured 1e-14 (H), 9e-15 (b), 0 (chi2); the slack is for a float32 rounding of one table entry that two float64
    # summation orders or eigen-solvers leave on different sides (2 of the 8208 golden information entries differ that way)
    assert rel_err(e["H"], g["H"]) < 1e-10 and rel_err(e["b"], g["b"]) < 1e-10 and abs(e["chi2"] - float(g["chi2"])) < 1e-10 * float(g["chi2"])
    h = len(c["scan"]) // 2
    Hs, bs, ns = np.zeros((6, 6)), np.zeros(6), 0
    for tag, part in (("a", c["scan"][:h]), ("b", c["scan"][h:])):
        o.set_source(part)
        ep = o.eval(c["guess"])
        assert ep["n_effective"] == int(g[f"neff_{tag}"])
        assert rel_err(ep["H"], g[f"H_{tag}"]) < 1e-10 and rel_err(ep["b"], g[f"b_{tag}"]) < 1e-10
        Hs += ep["H"]; bs += ep["b"]; ns += ep["n_effective"]
    assert ns == int(g["neff"]) and rel_err(Hs, e["H"]) < 1e-12 and rel_err(bs, e["b"]) < 1e-12      # shard sums = the whole, to rounding


def test_yard_map_information_matrices(golden):
    """An 8139-voxel map (the 200k-point yard map): keys and counts of every voxel, eigen-clamped inverse of 1500 of them."""
    from lio_slam_amd import synth
    g = golden["yard"]
    m = synth.make_config(2, small=True)["map"]
    assert float(m.astype(np.float64).sum()) == float(g["input_checksum"][0]), "synthetic inputs drifted: regenerate tests/golden/yard.npz"
    o = Oracle()
    o.set_target(m)
    _check_map_against_golden(o.export_voxels(), g)


def test_incndt3d_options_of_the_oracle():
    """max_pts_in_voxel freezes a voxel once it holds more points than that; info_rule ADD_1E3 gives (Sigma + 1e-3 I)^-1
    (SLAM-book IncNdt3d) — checked against NumPy on a handful of voxels."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(5)
    cells = np.array([[0, 0, 0], [3, 1, 0], [-2, 4, 1]], np.float32)
    def cloud(n_per, seed):
        r = np.random.default_rng(seed)
        return np.concatenate([c + 0.5 + np.clip(r.normal(0, [0.1, 0.05, 0.01], (n_per, 3)), -0.4, 0.4) for c in cells]).astype(np.float32)
    a, b = cloud(30, 1), cloud(40, 2)
    o = Oracle(min_pts_in_voxel=3, info_rule=1, quantize=0)
    o.set_target(a)
    vox = o.export_voxels()
    assert len(vox["keys"]) == 3
    for i in range(3):
        assert np.allclose(vox["info"][i], np.linalg.inv(vox["cov"][i] + 1e-3 * np.eye(3)), rtol=1e-9, atol=1e-9)
    # freeze: 30 points already > max_pts 20, so the merge changes nothing; with max_pts 50 it does
    o2 = Oracle(min_pts_in_voxel=3, max_pts_in_voxel=20, quantize=0)
    o2.set_target(a); before = o2.export_voxels(); o2.add_target(b); after = o2.export_voxels()
    assert np.array_equal(before["counts"], after["counts"]) and np.array_equal(before["mean"], after["mean"]) and np.array_equal(before["cov"], after["cov"])
    o3 = Oracle(min_pts_in_voxel=3, max_pts_in_voxel=50, quantize=0)
    o3.set_target(a); o3.add_target(b); merged = o3.export_voxels()
    assert np.array_equal(merged["counts"], before["counts"] + 40)
    o3.add_target(b); again = o3.export_voxels()          # 70 > 50: frozen now
    assert np.array_equal(again["counts"], merged["counts"]) and np.array_equal(again["cov"], merged["cov"])


def _lru_clouds(seed, n_clouds, pts_per_cloud, step):
    """Clouds of a sensor that moves along x: each covers a 12 x 6 x 2 m box whose origin advances `step` m per cloud,
    so older voxels fall out of view for good (no cloud re-enters a voxel that it evicted itself)."""
    r = np.random.default_rng(seed)
    return [(r.uniform([0, 0, 0], [12, 6, 2], (pts_per_cloud, 3)) + [step * c, 0, 0]).astype(np.float32) for c in range(n_clouds)]


def _lru_list_model(clouds, capacity):
    """IncNdt3d::AddCloud's bookkeeping, point by point: an ordered list of voxel keys, a point moves its
    voxel to the front, a new voxel is pushed to the front and the tail is dropped when the size reaches the capacity.
    Independent of the oracle: plain Python on the float32 floor rule at 1 m."""
    from collections import OrderedDict
    od = OrderedDict()                       # last = front of the list
    for c in clouds:
        for p in c:
            key = tuple(int(v) for v in np.floor(p))
            if key in od:
                od.move_to_end(key)
            else:
                od[key] = None
                if len(od) >= capacity:
                    od.popitem(last=False)
    return set(od)


def test_lru_capacity_of_the_oracle():
    """`capacity`: after a merge the least recently touched voxels go until capacity - 1 are left — the same set of voxels
    as the point-by-point LRU list of IncNdt3d keeps, checked against a plain-Python model of that list; the voxels that
    stay hold the statistics they would hold without a capacity."""
    from oracle.oracle import Oracle
    from lio_slam_amd import unpack_keys
    for capacity, step in ((120, 3.0), (60, 5.0), (200, 1.0)):
        clouds = _lru_clouds(11 + capacity, 9, 1500, step)
        o = Oracle(min_pts_in_voxel=3, capacity=capacity, quantize=0)
        ref = Oracle(min_pts_in_voxel=3, quantize=0)
        o.set_target(clouds[0]); ref.set_target(clouds[0])
        for c in clouds[1:]:
            o.add_target(c); ref.add_target(c)
        vox, full = o.export_voxels(), ref.export_voxels()
        want = _lru_list_model(clouds, capacity)
        got = set(map(tuple, unpack_keys(vox["keys"]).tolist()))
        assert len(got) == capacity - 1 and got == want
        # survivors that were never evicted and re-created carry the unbounded map's statistics, bit for bit
        pos = {int(k): i for i, k in enumerate(full["keys"])}
        same = 0
        for i, k in enumerate(vox["keys"]):
            j = pos[int(k)]
            if vox["counts"][i] == full["counts"][j]:
                assert np.array_equal(vox["mean"][i], full["mean"][j]) and np.array_equal(vox["cov"][i], full["cov"][j])
                same += 1
        assert same >= (capacity - 1) // 2
