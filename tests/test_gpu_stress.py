"""Stress cases for the voxel hash and the upload path: dense random maps (long probe chains, table growth over
several merges), ragged inputs (non-finite points, 16-byte records, device-resident buffers)."""
import ctypes

import numpy as np
import pytest

from conftest import rel_err, TIGHT_H

pytestmark = pytest.mark.gpu


def _dense_cloud(rng, n_cells=(48, 48, 12), per_cell=8, origin=(-20.0, 7.0, -3.0)):
    cells = np.stack(np.meshgrid(*[np.arange(k) for k in n_cells], indexing="ij"), -1).reshape(-1, 3)
    pts = (cells[:, None, :] + rng.uniform(0.05, 0.95, (len(cells), per_cell, 3))).reshape(-1, 3) + np.array(origin)
    return rng.permutation(pts).astype(np.float32)


def test_dense_random_map_growth_and_chains():
    import lio_slam_amd as la
    from oracle.oracle import Oracle
    rng = np.random.default_rng(21)
    m = _dense_cloud(rng)                                   # 27 648 voxels x 8 points, every voxel valid
    n = la.NDT(); o = Oracle()
    parts = np.array_split(m, 5)
    n.set_target(parts[0]); o.set_target(parts[0])
    for p in parts[1:]:                                      # the table grows (rehash) more than once
        n.add_target(p); o.add_target(p)
    vg, vo = n.export_voxels(), o.export_voxels()
    assert np.array_equal(vg["keys"], vo["keys"]) and np.array_equal(vg["counts"], vo["counts"])
    assert np.array_equal(vg["valid"], vo["valid"]) and len(vg["keys"]) >= 27000
    np.testing.assert_allclose(vg["mean"], vo["mean"], rtol=0, atol=1e-12)
    assert np.max(np.abs(vg["cov"] - vo["cov"]) / np.maximum(np.abs(vo["cov"]).max(axis=(1, 2), keepdims=True), 1e-300)) < 1e-10
    s = (rng.uniform(0, 1, (60000, 3)) * np.array([50, 50, 14]) + np.array([-21.0, 6.0, -4.0])).astype(np.float32)
    n.set_source(s); o.set_source(s)
    T = np.eye(4); T[:3, 3] = [0.013, -0.021, 0.008]
    for nearby in (7, 1):
        n.configure(nearby=nearby); o.set_params(nearby=nearby)
        kg, mg = n.source_cells(T); ko, mo = o.source_cells(T)
        assert np.array_equal(kg, ko) and np.array_equal(mg, mo)
        eg, eo = n.eval(T), o.eval(T)
        assert eg["n_effective"] == eo["n_effective"] and eo["n_effective"] > 10000
        assert rel_err(eg["H"], eo["H"]) < TIGHT_H and rel_err(eg["b"], eo["b"]) < TIGHT_H


def test_ragged_inputs_device_pointers_and_strides():
    import lio_slam_amd as la
    from lio_slam_amd import synth
    from oracle.oracle import Oracle
    c = synth.make_config(0)
    rng = np.random.default_rng(4)
    scan = c["scan"].copy()
    bad = rng.choice(len(scan), 500, replace=False)
    scan[bad[:250], rng.integers(0, 3, 250)] = np.nan           # the simulator's organised clouds hold NaN points
    scan[bad[250:], rng.integers(0, 3, 250)] = np.inf
    scan[0] = [3e6, 0, 0]                                        # finite but outside the 21-bit key range
    m16 = np.zeros((len(c["map"]), 4), np.float32); m16[:, :3] = c["map"]; m16[:, 3] = 7.0    # 16-byte records
    o = Oracle(); o.set_target(c["map"]); o.set_source(scan)
    n = la.NDT(); n.set_target(m16)
    # source handed over as a device pointer (hipMalloc'd by the caller), 32-byte slam_czc::Point records
    hip = ctypes.CDLL("libamdhip64.so")
    rec = synth.to_point32(scan)
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(rec.nbytes)) == 0
    assert hip.hipMemcpy(dptr, ctypes.c_void_p(rec.ctypes.data), ctypes.c_size_t(rec.nbytes), 1) == 0
    n.set_source((dptr.value, len(rec), 32))
    assert hip.hipFree(dptr) == 0                                # the engine copied: the caller's buffer is free
    kg, mg = n.source_cells(c["guess"]); ko, mo = o.source_cells(c["guess"])
    fin = np.isfinite(scan).all(axis=1)
    assert np.all(kg[~fin] == np.iinfo(np.int64).min) and kg[0] == np.iinfo(np.int64).min
    assert np.array_equal(kg[fin], ko) and np.array_equal(mg[fin], mo)     # the oracle drops non-finite points
    eg, eo = n.eval(c["guess"]), o.eval(c["guess"])
    assert eg["n_effective"] == eo["n_effective"]
    assert rel_err(eg["H"], eo["H"]) < TIGHT_H and rel_err(eg["b"], eo["b"]) < TIGHT_H
    with pytest.raises(la.NdtError):
        n.set_source((0, 10, 12))                                # null buffer
    with pytest.raises(la.NdtError):
        n.set_target((1, 10, 10))                                # stride not a multiple of 4 / below 12


def test_linked_and_hashed_nearby7_agree_bit_for_bit(monkeypatch):
    """NEARBY7 through the slots' neighbour links (default) and through seven hashed probes give identical bits."""
    import lio_slam_amd as la
    from lio_slam_amd import synth
    c = synth.make_config(1, small=True)
    res = []
    for no_links in (False, True):
        if no_links:
            monkeypatch.setenv("NDT_NO_LINKS", "1")
        n = la.NDT()
        n.set_target(c["map"][: len(c["map"]) // 2]); n.add_target(c["map"][len(c["map"]) // 2:])   # links rebuilt after a merge
        n.set_source(c["scan"])
        res.append((n.source_cells(c["guess"]), n.eval(c["guess"]), n.align(c["guess"])))
    (ka, ma), ea, ra = res[0]; (kb, mb), eb, rb = res[1]
    assert np.array_equal(ka, kb) and np.array_equal(ma, mb)
    assert np.array_equal(ea["H"], eb["H"]) and np.array_equal(ea["b"], eb["b"]) and ea["n_effective"] == eb["n_effective"]
    assert np.array_equal(ra["T"], rb["T"]) and ra["iterations"] == rb["iterations"]


def test_two_handles_on_two_host_threads_are_independent():
    """SURVEY.md §8(b) threading: a handle is not thread-safe, distinct handles are independent (one HIP stream each).
    Two host threads each drive their own handle through map build, merges, aligns and batches at the same time; every
    result must equal, bit for bit, what the same calls give when the threads run one after the other."""
    import threading

    import lio_slam_amd as la
    from lio_slam_amd import synth
    cfgs = [synth.make_config(0), synth.make_config(1)]

    def job(c, out):
        n = la.NDT(max_iterations=12, eps=1e-4)
        res = []
        for rep in range(6):
            n.set_target(c["map"]); n.set_source(c["scan"])
            res.append(n.align(c["guess"]))
            n.add_target(c["scan"], res[-1]["T"])
            res.append(n.align(c["guess"]))
            res += n.align_batch([c["scan"][::2], c["scan"][1::2]], [c["guess"], c["T_true"], c["guess"]], [0, 1, 1])
        out.append(res); n.close()

    serial = [[], []]
    for c, out in zip(cfgs, serial):
        job(c, out)
    conc = [[], []]
    th = [threading.Thread(target=job, args=(c, out)) for c, out in zip(cfgs, conc)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for a, b in zip(serial, conc):
        assert len(a) == len(b) == 1 and len(a[0]) == len(b[0]) == 6 * 5
        for ra, rb in zip(a[0], b[0]):
            assert ra.get("status", 0) == rb.get("status", 0) == 0 and ra["iterations"] == rb["iterations"]
            assert np.array_equal(ra["T"], rb["T"]) and np.array_equal(ra["H"], rb["H"]) and np.array_equal(ra["b"], rb["b"])


def test_chained_loop_matches_two_launch_loop_bit_for_bit(monkeypatch):
    """The single-scan loop runs one launch per iteration (the previous iteration's reduction + step in every block's
    prologue); NDT_NO_CHAIN=1 runs residual and reduce/solve as two launches.  Same sums in the same order, same step:
    identical bits, for a loop that stops early (eps), one that runs to the cap, one that fails, a source that
    fills the chained form's limit (one block per CU: 250 rows) and one above it (which runs as two launches either way)."""
    import lio_slam_amd as la
    from lio_slam_amd import synth
    c = synth.make_config(1, small=True)
    rng = np.random.default_rng(3)
    far = c["guess"].copy(); far[:3, 3] += [300.0, 0, 0]                      # no voxel in reach: too few residuals
    big = np.concatenate([c["scan"] + rng.normal(0, 0.01, c["scan"].shape).astype(np.float32) for _ in range(40)])
    mid = big[: 250 * 512 - 77]
    res = []
    for no_chain in (False, True):
        if no_chain:
            monkeypatch.setenv("NDT_NO_CHAIN", "1")
        out = []
        for cfg in (dict(max_iterations=35, eps=1e-2), dict(max_iterations=7, eps=0.0), dict(max_iterations=30, eps=1e-6, nearby=1),
                    dict(max_iterations=9, eps=1e-3, final_eval=0)):
            n = la.NDT(**cfg)
            n.set_target(c["map"]); n.set_source(c["scan"])
            out.append(n.align(c["guess"]))
            out.append(n.align(c["T_true"]))
            bad = n.align(far, check=False)
            assert bad["rc"] < 0
            out.append(bad)
            for src in (mid, big):
                n.set_source(src)
                out.append(n.align(c["guess"]))
            n.close()
        res.append(out)
    assert 249 * 512 < len(mid) <= 256 * 512 < len(big)
    for a, b in zip(*res):
        assert a["iterations"] == b["iterations"] and a["converged"] == b["converged"]
        for k in ("T", "H", "b"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert a["chi2"] == b["chi2"] and a["n_effective"] == b["n_effective"] and a["last_dx_norm"] == b["last_dx_norm"]
    assert res[0][0]["iterations"] < 35 and res[0][5]["iterations"] == 7


def test_dropped_points_never_tie_with_the_max_corner_cell():
    """A bounding box of exactly 128 x 128 x 4 cells has 16 sort-key bits and the (max, max, max) cell's key is all ones:
    dropped points (NaN, inf, out of key range) must still sort behind it, in the map build and in the VoxelGrid path."""
    import lio_slam_amd as la
    from oracle.oracle import Oracle, voxel_downsample
    rng = np.random.default_rng(128)
    cells = np.array([[0, 0, 0], [127, 127, 3], [127, 127, 2], [126, 127, 3], [5, 9, 1], [64, 64, 2]])
    pts = (cells[:, None, :] + rng.uniform(0.05, 0.95, (len(cells), 9, 3))).reshape(-1, 3).astype(np.float32)
    bad = np.full((40, 3), np.nan, np.float32); bad[::3, 1] = np.inf; bad[1::3] = [3e6, 1.0, 1.0]
    cloud = np.concatenate([pts, bad])
    cloud = cloud[rng.permutation(len(cloud))]
    n = la.NDT(); o = Oracle()
    n.set_target(cloud); o.set_target(cloud)
    vg, vo = n.export_voxels(), o.export_voxels()
    assert len(vg["keys"]) == len(cells)
    assert np.array_equal(vg["keys"], vo["keys"]) and np.array_equal(vg["counts"], vo["counts"]) and np.all(vg["counts"] == 9)
    assert np.array_equal(vg["valid"], vo["valid"]) and np.isfinite(vg["mean"]).all() and np.isfinite(vg["cov"]).all()
    np.testing.assert_allclose(vg["mean"], vo["mean"], rtol=0, atol=1e-12)
    n.add_target(cloud); o.add_target(cloud)                           # the merge path sorts the same way
    vg, vo = n.export_voxels(), o.export_voxels()
    assert np.array_equal(vg["keys"], vo["keys"]) and np.all(vg["counts"] == 18) and np.array_equal(vg["counts"], vo["counts"])
    # VoxelGrid with leaf 1.0 on the same cloud: six centroids in PCL's index order, none of them NaN
    n.set_source_downsampled(cloud, leaf=1.0)
    got = n.get_source(); want = voxel_downsample(cloud, 1.0)
    assert got.shape == want.shape == (len(cells), 3) and np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_radix_pass_that_gives_up_is_sorted_again(tmp_path):
    """A one-sweep radix pass whose bounded wait runs out raises a flag; the insert stages then refuse the batch and the host
    sorts again with the three-kernel passes.  NDT_TEST_SORT_ERR=1 raises the flag after every one-sweep sort (the switch is
    read once per process, hence the child process): build + merge must give the same map, bit for bit."""
    import os, subprocess, sys
    import lio_slam_amd as la
    from lio_slam_amd import synth
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import lio_slam_amd as la\n"
        "from lio_slam_amd import synth\n"
        "c = synth.make_config(1, small=True)\n"
        "n = la.NDT(); n.set_target(c['map']); n.add_target(c['scan'], c['guess'])\n"
        "v = n.export_voxels()\n"
        "o = np.argsort(v['keys'])\n"
        "np.savez(sys.argv[1], **{k: np.asarray(x)[o] for k, x in v.items()})\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NDT_TEST_SORT_ERR="1")
    out = tmp_path / "resorted.npz"
    r = subprocess.run([sys.executable, "-c", code, str(out)], env=env, capture_output=True, timeout=150)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    got = np.load(out)
    c = synth.make_config(1, small=True)
    n = la.NDT(); n.set_target(c["map"]); n.add_target(c["scan"], c["guess"])
    v = n.export_voxels()
    o = np.argsort(v["keys"])
    for k, x in v.items():
        assert np.array_equal(np.asarray(x)[o], got[k]), k


def test_stream_ordered_source_gives_the_same_align():
    """ndt_set_source_async (device records, no host wait) followed by align = ndt_set_source + align, bit for bit; a host
    pointer is refused."""
    import lio_slam_amd as la
    from lio_slam_amd import synth
    import bench
    c = synth.make_config(1, small=True)
    n = la.NDT()
    n.set_target(c["map"])
    n.set_source(c["scan"])
    r0 = n.align(c["guess"])
    d = bench.DeviceCloud(c["scan"])
    try:
        other = bench.DeviceCloud(c["scan"][: len(c["scan"]) // 2])
        try:
            n.set_source_async(other.view()); n.align(c["guess"], check=False)
            n.set_source_async(d.view())
            r1 = n.align(c["guess"])
            assert n.source_size() == len(c["scan"])
        finally:
            other.free()
    finally:
        d.free()
    assert np.array_equal(r0["T"], r1["T"]) and np.array_equal(r0["H"], r1["H"]) and r0["iterations"] == r1["iterations"]
    with pytest.raises(la.NdtError) as ei:
        n.set_source_async(c["scan"])
    assert ei.value.rc == -1
    # the records stay pending until something reads the source: every reader must find them (eval, read-back, keyframe
    # merge, and the two-launch loop), not only the chained loop's first launch
    e0 = n.eval(c["guess"])
    d = bench.DeviceCloud(c["scan"])
    try:
        n.set_source_async(d.view()); e1 = n.eval(c["guess"])
        assert np.array_equal(e0["H"], e1["H"]) and e0["n_effective"] == e1["n_effective"]
        n.set_source_async(d.view())
        assert np.array_equal(n.get_source(), c["scan"][:, :3])
        m0 = la.NDT(); m0.set_target(c["map"]); m0.set_source(c["scan"]); m0.add_target_from_source(c["guess"])
        m1 = la.NDT(); m1.set_target(c["map"]); m1.set_source_async(d.view()); m1.add_target_from_source(c["guess"])
        v0, v1 = m0.export_voxels(), m1.export_voxels()
        o0, o1 = np.argsort(v0["keys"]), np.argsort(v1["keys"])
        for k in v0:
            assert np.array_equal(np.asarray(v0[k])[o0], np.asarray(v1[k])[o1]), k
        m1.set_source_async(d.view()); m1.profile(True)            # profiling uses the two-launch loop
        rp = m1.align(c["guess"]); m1.profile(False)
        m0.set_source(c["scan"]); rq = m0.align(c["guess"])
        assert np.array_equal(rp["T"], rq["T"])
        m0.close(); m1.close()
    finally:
        d.free()


def test_pending_async_records_never_overwrite_a_later_source():
    """ndt_set_source_async leaves records pending; a later call that REPLACES the source without reading it
    (ndt_set_source_downsampled, ndt_set_source) must drop them: the next align uses the new source, not an unpack of the
    old caller buffer — which by then may be freed, and is shorter or longer than the new source."""
    import lio_slam_amd as la
    from lio_slam_amd import synth
    import bench
    c = synth.make_config(1, small=True)
    ref = la.NDT(); ref.set_target(c["map"])
    ref.set_source_downsampled(c["scan"], leaf=0.5)
    want_src = ref.get_source(); want = ref.align(c["guess"])
    n = la.NDT(); n.set_target(c["map"])
    for pending in (c["scan"][: len(c["scan"]) // 8], np.concatenate([c["scan"], c["scan"]])):      # shorter / longer than the new source
        d = bench.DeviceCloud(pending)
        n.set_source_async(d.view())
        n.set_source_downsampled(c["scan"], leaf=0.5)
        d.free()                                               # the pending records are gone: nothing may read them any more
        got = n.align(c["guess"])
        assert n.source_size() == len(want_src) and np.array_equal(n.get_source(), want_src)
        assert np.array_equal(got["T"], want["T"]) and np.array_equal(got["H"], want["H"]) and got["iterations"] == want["iterations"]
        # the same with a plain ndt_set_source in between
        d = bench.DeviceCloud(pending)
        n.set_source_async(d.view()); n.set_source(want_src); d.free()
        got = n.align(c["guess"])
        assert np.array_equal(got["T"], want["T"]) and np.array_equal(got["H"], want["H"])
    # a failed replacement leaves NO source (not the half-replaced old one)
    with pytest.raises(la.NdtError):
        n.set_source_async(c["scan"])                          # host pointer: refused before anything changes
    assert n.source_size() == len(want_src)
    n.close(); ref.close()
