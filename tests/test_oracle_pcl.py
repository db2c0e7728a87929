"""PCL compatibility mode: the oracle's PCL-literal restatement against the independent NumPy
vectors of tests/golden/pcl.npz (kd-tree radius search, vectorised eq. 6.12 / 6.13, checked there against finite
differences).  Parity with PCL itself is unpinned: PCL is not in this image."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import oracle as ora
from oracle.oracle import Oracle


@pytest.fixture(scope="module")
def pcl_case(golden):
    o = Oracle()
    o.set_target(golden["pair"]["map"]); o.set_source(golden["pair"]["scan"])
    return o, golden["pcl"]


def test_vector_matrix_round_trip(golden):
    g = golden["pcl"]
    p = ora.pcl_vector_from_matrix(g["guess"])
    assert np.allclose(p, g["p"], atol=1e-14)
    assert np.allclose(ora.pcl_matrix_from_vector(p), g["guess"], atol=1e-14)
    # Eigen's convention: the first angle comes back in [0, pi]; a negative roll flips the other two
    T = ora.pcl_matrix_from_vector([1, 2, 3, -0.2, 0.1, 0.3])
    q = ora.pcl_vector_from_matrix(T)
    assert 0 <= q[3] <= np.pi and np.allclose(ora.pcl_matrix_from_vector(q), T, atol=1e-14)


@pytest.mark.parametrize("tag,flag", [("pcl", 1), ("eq621", 0)])
def test_derivatives_match_numpy(pcl_case, tag, flag):
    o, g = pcl_case
    r = o.pcl_eval(g["p"], hessian_d1_as_pcl=flag)
    # float32 table payload on both sides (measured 4e-16; the bar leaves room for a float32 rounding of one table entry that a
    # last-bit difference of the float64 statistics puts on the other side)
    assert r["n_neighbours"] == int(g[f"nb_{tag}"])
    assert abs(r["score"] - float(g[f"score_{tag}"])) <= 1e-10 * abs(float(g[f"score_{tag}"]))
    assert rel_err(r["g"], g[f"g_{tag}"]) < 1e-10
    assert rel_err(r["H"], g[f"H_{tag}"]) < 1e-10


def test_derivatives_match_numpy_float64_table(golden):
    g = golden["pcl"]
    o = Oracle(quantize=False)
    o.set_target(golden["pair"]["map"]); o.set_source(golden["pair"]["scan"])
    r = o.pcl_eval(g["p"])
    assert r["n_neighbours"] == int(g["nb_q0"])
    assert abs(r["score"] - float(g["score_q0"])) <= 1e-12 * abs(float(g["score_q0"]))
    assert rel_err(r["g"], g["g_q0"]) < 1e-12
    assert rel_err(r["H"], g["H_q0"]) < 1e-12


def test_derivatives_large_angles(pcl_case):
    o, g = pcl_case
    r = o.pcl_eval(g["p2"])
    assert r["n_neighbours"] == int(g["nb2"])
    assert abs(r["score"] - float(g["score2"])) <= 1e-10 * abs(float(g["score2"]))
    assert rel_err(r["g"], g["g2"]) < 1e-10 and rel_err(r["H"], g["H2"]) < 1e-10


def test_align_runs_as_pcl_does(pcl_case, golden):
    o, g = pcl_case
    o.set_params(max_iterations=35, eps=0.01)
    r = o.pcl_align(golden["pair"]["guess"])              # the reference's parameters: step <= 0.1, |dt|^2 <= 0.01
    assert r["rc"] == 0 and r["converged"] and r["iterations"] == 1   # a step of <= 0.1 always meets |dt|^2 <= 0.01
    s0 = o.pcl_eval(g["p"])["score"]
    assert r["score"] > s0                                  # More-Thuente: sufficient increase of the score
    assert np.linalg.norm(r["p"] - g["p"]) <= 0.1 + 1e-12
    # with a tight epsilon the Newton iteration goes on and ends near the true pose
    o.set_params(max_iterations=35, eps=1e-6)
    r2 = o.pcl_align(golden["pair"]["guess"])
    assert r2["rc"] == 0 and r2["iterations"] > 3 and r2["score"] > r["score"]
    err = np.linalg.norm(r2["T"][:3, 3] - golden["pair"]["T_true"][:3, 3])
    assert err < 0.05, err
    o.set_params(max_iterations=35, eps=0.01)
