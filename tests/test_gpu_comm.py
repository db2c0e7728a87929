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
