"""ndt_create_group (the replay half of §8(e)): several member engines in one process, map replicas,
pairs dealt in contiguous shares.  The box has one GPU, so the members share device 0 — the fan-out, the per-member scan
gathering and the result order are what is under test, not the devices."""
import numpy as np
import pytest

import lio_slam_amd as la
from conftest import pose_error, rel_err
from lio_slam_amd import synth

pytestmark = pytest.mark.gpu


def test_group_replay_matches_single_handle():
    c = synth.make_config(1, small=True)
    rng = np.random.Generator(np.random.PCG64(17))
    scans = [c["scan"][i::5] for i in range(5)]                        # five scans of ~6 k points
    n_pairs = 23
    pair_scan = rng.integers(0, 5, n_pairs)                             # unsorted: members gather different scan subsets
    guesses = np.stack([c["T_true"] @ synth.perturbation(rng) for _ in range(n_pairs)])
    half = len(c["map"]) // 2
    one = la.NDT(max_iterations=8, eps=1e-3)
    one.set_target(c["map"][:half]); one.add_target(c["map"][half:])
    ref = one.align_batch(scans, guesses, pair_scan)
    for devices in ([0], [0, 0, 0], [0] * 4):
        g = la.NDTGroup(devices, max_iterations=8, eps=1e-3)
        assert len(g) == len(devices)
        g.set_target(c["map"][:half]); g.add_target(c["map"][half:])
        for m in range(len(g)):                                          # every member holds the same map
            vm, v1 = g.member(m).export_voxels(), one.export_voxels()
            assert np.array_equal(vm["keys"], v1["keys"]) and np.array_equal(vm["counts"], v1["counts"]) and np.array_equal(vm["mean"], v1["mean"])
        res = g.align_batch(scans, guesses, pair_scan)
        assert len(res) == n_pairs
        for a, b in zip(res, ref):
            assert a["status"] == b["status"] == 0 and a["iterations"] == b["iterations"]
            dt, dr = pose_error(a["T"], b["T"])
            assert dt < 1e-9 and dr < 1e-9 and rel_err(a["H"], b["H"]) < 1e-10        # batch size differs: summation order only
        again = g.align_batch(scans, guesses, pair_scan)
        assert all(np.array_equal(a["T"], b["T"]) for a, b in zip(res, again))       # deterministic
        # more members than pairs, one scan per pair
        few = g.align_batch(scans[:2], guesses[:2])
        assert len(few) == 2 and all(r["status"] == 0 for r in few)
        g.close()
    one.close()


def test_group_errors():
    with pytest.raises(la.NdtError):
        la.NDTGroup([0, 99])                                            # no such device
    g = la.NDTGroup([0, 0])
    c = synth.make_config(0, small=True)
    with pytest.raises(la.NdtError) as ei:
        g.align_batch([c["scan"]], [np.eye(4)])                         # no target yet
    assert ei.value.rc == -2 and "member 0" in str(ei.value)
    g.set_target(c["map"])
    with pytest.raises(la.NdtError):
        g.align_batch([c["scan"]], [np.eye(4), np.eye(4)], [0, 3])      # pair_scan out of range
    g.close()
