"""PCL compatibility mode on the GPU against the oracle's PCL-literal restatement and the NumPy
golden vectors.  Tolerances: the neighbour count is exact (same float32 transform, cell rule and distance test);
score / gradient / Hessian <= 1e-10 relative (measured 6e-15: summation order and the device exp; the bar leaves room for one
float32 rounding of one table entry landing on the other side); poses of the Newton / More-Thuente loop <= 1e-9 m / 1e-9 rad
(measured 7e-15 m).  Parity with PCL itself is unpinned (PCL is not in the image)."""
import numpy as np
import pytest

import lio_slam_amd as la
from conftest import pose_error, rel_err, TIGHT_H, TIGHT_T, TIGHT_R
from lio_slam_amd import synth
from oracle import oracle as ora
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _pair(golden, **cfg):
    n = la.NDT(**cfg); o = Oracle(**cfg)
    n.set_target(golden["pair"]["map"]); o.set_target(golden["pair"]["map"])
    n.set_source(golden["pair"]["scan"]); o.set_source(golden["pair"]["scan"])
    return n, o


def test_host_helpers_match_oracle(golden):
    g = golden["pcl"]
    assert np.array_equal(la.pcl_vector_from_matrix(g["guess"]), ora.pcl_vector_from_matrix(g["guess"]))
    assert np.array_equal(la.pcl_matrix_from_vector(g["p2"]), ora.pcl_matrix_from_vector(g["p2"]))


@pytest.mark.parametrize("flag", [1, 0])
def test_derivatives_match_oracle_and_golden(golden, flag):
    n, o = _pair(golden)
    g = golden["pcl"]
    for p in (g["p"], g["p2"]):
        rn = n.pcl_eval(p, hessian_d1_as_pcl=flag); ro = o.pcl_eval(p, hessian_d1_as_pcl=flag)
        assert rn["n_neighbours"] == ro["n_neighbours"]
        assert abs(rn["score"] - ro["score"]) <= TIGHT_H * abs(ro["score"])
        assert rel_err(rn["g"], ro["g"]) < TIGHT_H and rel_err(rn["H"], ro["H"]) < TIGHT_H
        assert np.array_equal(rn["H"], rn["H"].T)
    tag = "pcl" if flag else "eq621"
    rn = n.pcl_eval(g["p"], hessian_d1_as_pcl=flag)
    assert rn["n_neighbours"] == int(g[f"nb_{tag}"])
    assert rel_err(rn["g"], g[f"g_{tag}"]) < TIGHT_H and rel_err(rn["H"], g[f"H_{tag}"]) < TIGHT_H
    # gradient-only pass gives the same score and gradient bits
    r2 = n.pcl_eval(g["p"], hessian=False, hessian_d1_as_pcl=flag)
    assert r2["score"] == rn["score"] and np.array_equal(r2["g"], rn["g"])


def test_align_as_the_reference_configures_it(golden):
    """process.cc:373-376: resolution 1, <= 35 iterations, eps 0.01, step 0.1 -> PCL stops after one capped step."""
    n, o = _pair(golden, max_iterations=35, eps=0.01)
    rn = n.pcl_align(golden["pair"]["guess"]); ro = o.pcl_align(golden["pair"]["guess"])
    assert rn["rc"] == 0 and ro["rc"] == 0
    assert rn["iterations"] == ro["iterations"] == 1 and rn["converged"] and ro["converged"]
    assert rn["evaluations"] == ro["evaluations"]
    dt, dr = pose_error(rn["T"], ro["T"])
    assert dt < TIGHT_T and dr < TIGHT_R
    assert abs(rn["score"] - ro["score"]) <= TIGHT_H * abs(ro["score"])


def test_align_tight_epsilon_and_nonfinite_points(golden):
    n, o = _pair(golden, max_iterations=35, eps=1e-6)
    s = golden["pair"]["scan"].copy(); s[5] = np.nan; s[77, 1] = np.inf
    n.set_source(s); o.set_source(s)
    rn = n.pcl_align(golden["pair"]["guess"]); ro = o.pcl_align(golden["pair"]["guess"])
    assert rn["rc"] == 0 and rn["iterations"] == ro["iterations"] and rn["iterations"] > 3
    dt, dr = pose_error(rn["T"], ro["T"])
    assert dt < TIGHT_T and dr < TIGHT_R
    assert rel_err(rn["H"], ro["H"]) < TIGHT_H and np.all(np.isfinite(rn["H"]))
    et, _ = pose_error(rn["T"], golden["pair"]["T_true"])
    assert et < 0.05


def test_full_size_scan(golden):
    """configs[1]-sized case: 30k-pt scan against a 50k-pt map, oracle in a few seconds."""
    c = synth.make_config(1, small=True)
    n = la.NDT(max_iterations=10, eps=1e-6); o = Oracle(max_iterations=10, eps=1e-6)
    n.set_target(c["map"]); o.set_target(c["map"]); n.set_source(c["scan"]); o.set_source(c["scan"])
    p = la.pcl_vector_from_matrix(c["guess"])
    rn = n.pcl_eval(p); ro = o.pcl_eval(p)
    assert rn["n_neighbours"] == ro["n_neighbours"] and rn["n_neighbours"] > len(c["scan"])
    assert rel_err(rn["g"], ro["g"]) < TIGHT_H and rel_err(rn["H"], ro["H"]) < TIGHT_H
    ra = n.pcl_align(c["guess"]); rb = o.pcl_align(c["guess"])
    assert ra["iterations"] == rb["iterations"]
    dt, dr = pose_error(ra["T"], rb["T"])
    assert dt < TIGHT_T and dr < TIGHT_R
    assert ra["score"] > rn["score"]


def test_errors():
    n = la.NDT()
    with pytest.raises(la.NdtError):
        n.pcl_eval(np.zeros(6))                      # no target
    c = synth.make_config(0, small=True)
    n.set_target(c["map"]); n.set_source(c["scan"])
    with pytest.raises(la.NdtError):
        n.pcl_eval(np.zeros(6), outlier_ratio=1.5)
    with pytest.raises(la.NdtError):
        n.pcl_eval(np.array([0, 0, np.nan, 0, 0, 0.0]))


def test_linked_rows_match_hashed_rows(golden, monkeypatch):
    """27 hashed cells (NDT_NO_LINKS=1) and 9 hashed cells + x-links visit the same voxels in the same order: same bits."""
    c = synth.make_config(1, small=True)
    p = la.pcl_vector_from_matrix(c["guess"])
    res = []
    for no_links in (False, True):
        if no_links:
            monkeypatch.setenv("NDT_NO_LINKS", "1")
        n = la.NDT(max_iterations=10, eps=1e-6)
        half = len(c["map"]) // 2
        n.set_target(c["map"][:half]); n.add_target(c["map"][half:])          # links rebuilt after a merge
        n.set_source(c["scan"])
        res.append((n.pcl_eval(p), n.pcl_align(c["guess"])))
    (ea, ra), (eb, rb) = res
    assert ea["n_neighbours"] == eb["n_neighbours"] and ea["score"] == eb["score"]
    assert np.array_equal(ea["g"], eb["g"]) and np.array_equal(ea["H"], eb["H"])
    assert np.array_equal(ra["T"], rb["T"]) and ra["iterations"] == rb["iterations"] and ra["evaluations"] == rb["evaluations"]
