"""Offline replay (BASELINE.json configs[4]): ndt_align_batch against per-pair oracle aligns."""
import numpy as np
import pytest

from conftest import pose_error, rel_err, TIGHT_H, TIGHT_T, TIGHT_R

pytestmark = pytest.mark.gpu


def test_batch_matches_oracle_and_single():
    import lio_slam_amd as la
    from lio_slam_amd import synth
    from oracle.oracle import Oracle
    c = synth.make_config(1, small=True)
    rng = np.random.default_rng(9)
    poses = [c["poses"][i] for i in (2, 5, 8)]
    scans = [synth.scan(c["scene"], "vlp16", T, rng) for T in poses]
    scans.append(np.zeros((0, 3), np.float32))                     # an empty scan: its pairs must fail alone
    pair_scan, guesses = [], []
    for si, T in enumerate(poses):
        for _ in range(2):                                         # two hypotheses per scan
            pair_scan.append(si); guesses.append(T @ synth.perturbation(rng))
    pair_scan.append(3); guesses.append(np.eye(4))
    n = la.NDT(); o = Oracle()
    n.set_target(c["map"]); o.set_target(c["map"])
    res = n.align_batch(scans, guesses, pair_scan)
    assert len(res) == 7
    for p, r in enumerate(res[:6]):
        o.set_source(scans[pair_scan[p]])
        ro = o.align(guesses[p])
        assert r["status"] == 0 and ro["rc"] == 0
        assert r["iterations"] == ro["iterations"] and r["converged"] == ro["converged"]
        dt, dr = pose_error(r["T"], ro["T"])
        assert dt < TIGHT_T and dr < TIGHT_R
        assert rel_err(r["H"], ro["H"]) < TIGHT_H and r["n_effective"] == ro["n_effective"]
        et, er = pose_error(r["T"], poses[pair_scan[p]])
        assert et < 0.05 and er < 0.01
        n.set_source(scans[pair_scan[p]])                           # the single-scan path agrees with the batch
        rs = n.align(guesses[p])
        dt, dr = pose_error(r["T"], rs["T"])
        assert dt < 1e-9 and dr < 1e-9 and rel_err(r["H"], rs["H"]) < 1e-10
    assert res[6]["status"] == -5 and res[6]["iterations"] == 0
    # without pair_scan: one scan per pair
    res2 = n.align_batch(scans[:3], [guesses[0], guesses[2], guesses[4]])
    for a, b in zip(res2, (res[0], res[2], res[4])):
        # the grid (points per lane) is chosen per batch size, so the summation tree differs: last bits only
        assert pose_error(a["T"], b["T"])[0] < 1e-9 and rel_err(a["H"], b["H"]) < 1e-10
    with pytest.raises(la.NdtError):
        n.align_batch(scans[:3], guesses[:2])
