"""The sharded mode on one GPU: a one-rank communicator (RCCL, peer exchange) must leave eval/align results unchanged; two
ranks sharing the GPU exchange their rows through each other's mailboxes (HIP IPC) and agree bit for bit."""
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


def test_p2p_single_rank_is_identity():
    import lio_slam_amd as la
    from lio_slam_amd import synth
    c = synth.make_config(0)
    n = la.NDT()
    n.set_target(c["map"]); n.set_source(c["scan"])
    e0 = n.eval(c["guess"]); r0 = n.align(c["guess"])
    h = n.comm_p2p_export()
    assert len(h) == 128
    n.comm_p2p_init(0, 1, [h])
    assert n.comm_size() == 1
    e1 = n.eval(c["guess"]); r1 = n.align(c["guess"])
    assert np.array_equal(e0["H"], e1["H"]) and np.array_equal(e0["b"], e1["b"]) and e0["n_effective"] == e1["n_effective"]
    assert np.array_equal(r0["T"], r1["T"]) and r0["iterations"] == r1["iterations"] and np.array_equal(r0["H"], r1["H"])
    n.comm_destroy()
    r2 = n.align(c["guess"])
    assert np.array_equal(r0["T"], r2["T"])


def test_p2p_two_ranks_share_one_gpu(tmp_path, monkeypatch):
    """The N > 1 collective path, executed: two processes on this GPU, each with half of the source, exchange their reduced
    rows through each other's mailboxes on every Gauss-Newton iteration.  Both must end with identical bits, and with the
    unsharded result up to the summation order.  Also with the step as a launch of its own (NDT_NO_CHAIN=1)."""
    import os, subprocess, sys
    import lio_slam_amd as la
    from lio_slam_amd import synth
    c = synth.make_config(1, small=True)
    ref = {}
    for tag, cfg in (("conv", dict(max_iterations=35, eps=1e-2)), ("fixed", dict(max_iterations=6, eps=0.0))):
        n = la.NDT(**cfg)
        n.set_target(c["map"]); n.set_source(c["scan"])
        ref[tag] = (n.eval(c["guess"]), n.align(c["guess"]))
        n.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "p2p_worker.py")
    # chain: the rows are waited for in every block's prologue (what ranks with a GPU each run; forced here, the grids are
    # small enough to leave the other rank room); oneblock: what ranks that share a GPU get by default; nochain: NDT_NO_CHAIN
    for variant in ("chain", "oneblock", "nochain"):
        d = tmp_path / variant
        d.mkdir()
        env = dict(os.environ)
        if variant == "nochain":
            env["NDT_NO_CHAIN"] = "1"
        if variant == "chain":
            env["NDT_P2P_CHAIN"] = "1"
        procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(d)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        outs = []
        try:
            for p in procs:
                o, _ = p.communicate(timeout=150)
                outs.append(o.decode(errors="replace"))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        assert all(p.returncode == 0 for p in procs), outs
        r0, r1 = np.load(d / "result0.npz"), np.load(d / "result1.npz")
        for k in r0.files:
            assert np.array_equal(r0[k], r1[k]), (variant, k)          # every rank summed the same rows in the same order
        for tag in ("conv", "fixed"):
            e, r = ref[tag]
            assert int(r0[f"{tag}_neff"]) == e["n_effective"]
            assert np.allclose(r0[f"{tag}_H"], e["H"], rtol=1e-9, atol=1e-9 * np.abs(e["H"]).max())
            assert np.allclose(r0[f"{tag}_b"], e["b"], rtol=1e-9, atol=1e-9 * np.abs(e["b"]).max())
            assert int(r0[f"{tag}_it"]) == r["iterations"] and bool(r0[f"{tag}_conv"]) == bool(r["converged"])
            assert np.abs(r0[f"{tag}_T"] - r["T"]).max() < 1e-9
