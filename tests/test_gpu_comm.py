"""The sharded mode's RCCL path on one GPU: a one-rank communicator must leave eval/align results unchanged."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_is_identity():
    import lio_slam_amd as la
    from lio_slam_amd import synth
    c = synth.make_config(0)
    n = la.NDT()
    n.set_target(c["map"]); n.set_source(c["scan"])
    e0 = n.eval(c["guess"]); r0 = n.align(c["guess"])
    uid = la.NDT.comm_unique_id()
    assert len(uid) == 128
    n.comm_init(0, 1, uid)
    e1 = n.eval(c["guess"]); r1 = n.align(c["guess"])
    assert np.array_equal(e0["H"], e1["H"]) and np.array_equal(e0["b"], e1["b"]) and e0["n_effective"] == e1["n_effective"]
    assert np.array_equal(r0["T"], r1["T"]) and r0["iterations"] == r1["iterations"] and np.array_equal(r0["H"], r1["H"])
    keys0, m0 = n.source_cells(c["guess"])          # a local diagnostic: no collective
    n.comm_destroy()
    keys1, m1 = n.source_cells(c["guess"])
    assert np.array_equal(keys0, keys1) and np.array_equal(m0, m1)
    r2 = n.align(c["guess"])
    assert np.array_equal(r0["T"], r2["T"])


def test_chained_sharded_loop_matches_four_launch_loop(monkeypatch):
    """With a communicator the loop runs residual(+previous step) -> local sum -> all-reduce per iteration; NDT_NO_CHAIN=1
    keeps the step as a launch of its own.  Same bits, also for a fixed iteration count, a large source and a failure."""
    import lio_slam_amd as la
    from lio_slam_amd import synth
    c = synth.make_config(1, small=True)
    rng = np.random.default_rng(8)
    big = np.concatenate([c["scan"] + rng.normal(0, 0.01, c["scan"].shape).astype(np.float32) for _ in range(40)])   # 1.2 M points: 2048 rows
    far = c["guess"].copy(); far[:3, 3] += [300.0, 0, 0]
    res = []
    for no_chain in (False, True):
        if no_chain:
            monkeypatch.setenv("NDT_NO_CHAIN", "1")
        out = []
        for cfg in (dict(max_iterations=35, eps=1e-2), dict(max_iterations=6, eps=0.0)):
            n = la.NDT(**cfg)
            n.set_target(c["map"])
            n.comm_init(0, 1, la.NDT.comm_unique_id())
            for src in (c["scan"], big):
                n.set_source(src)
                out.append(n.align(c["guess"]))
            out.append(n.align(far, check=False))
            n.comm_destroy(); n.close()
        res.append(out)
    for a, b in zip(*res):
        assert a["rc"] == b["rc"] and a["iterations"] == b["iterations"] and a["converged"] == b["converged"]
        assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["H"], b["H"]) and np.array_equal(a["b"], b["b"])
    assert res[0][2]["rc"] < 0 and res[0][3]["iterations"] == 6
