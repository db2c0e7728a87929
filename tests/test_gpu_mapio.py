"""Voxel-map serialisation: a loaded map must answer bit for bit like the saved one."""
import struct
import numpy as np
import pytest
import lio_slam_amd as la
from lio_slam_amd import synth

pytestmark = pytest.mark.gpu


def _case(seed=5):
    c = synth.make_config(1, small=True)
    return c


def test_save_load_roundtrip_bit_exact():
    c = _case()
    a = la.NDT(max_iterations=8, eps=1e-4)
    half = len(c["map"]) // 2
    a.set_target(c["map"][:half]); a.add_target(c["map"][half:])          # a map grown by an incremental merge
    a.set_source(c["scan"])
    blob = a.save_map()
    magic, ver, res, rule, _, nv = struct.unpack_from("<8sIfiiQ", blob, 0)
    assert magic == b"NDTMI355" and ver == 1 and res == 1.0 and nv == a.num_voxels()
    assert len(blob) == 64 + nv * 84
    b = la.NDT(max_iterations=8, eps=1e-4)
    b.load_map(blob); b.set_source(c["scan"])
    va, vb = a.export_voxels(), b.export_voxels()
    for k in ("keys", "counts", "mean", "cov", "info", "valid"):
        assert np.array_equal(va[k], vb[k]), k
    ea, eb = a.eval(c["guess"]), b.eval(c["guess"])
    assert np.array_equal(ea["H"], eb["H"]) and np.array_equal(ea["b"], eb["b"]) and ea["chi2"] == eb["chi2"]
    ra, rb = a.align(c["guess"]), b.align(c["guess"])
    assert np.array_equal(ra["T"], rb["T"]) and ra["iterations"] == rb["iterations"]
    assert b.save_map() == blob                                            # idempotent
    # merges continue from the same statistics
    extra = synth.transform(c["T_true"], c["scan"]).astype(np.float32)
    a.add_target(extra); b.add_target(extra)
    va, vb = a.export_voxels(), b.export_voxels()
    for k in ("keys", "counts", "mean", "cov", "info", "valid"):
        assert np.array_equal(va[k], vb[k]), k


def test_load_replaces_and_revalidates_under_current_config():
    c = _case()
    a = la.NDT(); a.set_target(c["map"])
    blob = a.save_map()
    b = la.NDT(min_pts_in_voxel=20)
    b.set_target(c["map"][:1000])                 # replaced by the load
    b.load_map(blob)
    vb = b.export_voxels(); va = a.export_voxels()
    assert np.array_equal(va["keys"], vb["keys"]) and np.array_equal(va["counts"], vb["counts"])
    assert np.array_equal(vb["valid"].astype(bool), vb["counts"] >= 20)


def test_bad_blobs_are_rejected():
    c = _case()
    a = la.NDT(); a.set_target(c["map"][:5000])
    blob = bytearray(a.save_map())
    n = la.NDT()
    for bad in (b"", bytes(blob[:40]), bytes(blob[:-4]), b"XXXXXXXX" + bytes(blob[8:])):
        with pytest.raises(la.NdtError):
            n.load_map(bad)
    other = la.NDT(resolution=0.5)
    with pytest.raises(la.NdtError):
        other.load_map(bytes(blob))
    dup = bytearray(blob); dup[64 + 8:64 + 16] = dup[64:64 + 8]            # second key := first key
    with pytest.raises(la.NdtError):
        n.load_map(bytes(dup))
    nan = bytearray(blob); nv = struct.unpack_from("<Q", blob, 24)[0]
    struct.pack_into("<d", nan, 64 + nv * 8, float("nan"))
    with pytest.raises(la.NdtError):
        n.load_map(bytes(nan))
    n.load_map(bytes(blob))                                                # the handle is still usable
    assert n.num_voxels() == a.num_voxels()


def test_loaded_map_under_a_voxel_capacity():
    """A map loaded into a handle with `capacity` counts as one cloud (file order); the next merge cuts it down to
    capacity - 1 voxels and keeps every voxel the new cloud touched."""
    import lio_slam_amd as la
    r = np.random.default_rng(3)
    first = r.uniform([0, 0, 0], [20, 10, 2], (6000, 3)).astype(np.float32)        # 400 cells
    a = la.NDT(min_pts_in_voxel=3)
    a.set_target(first)
    blob = a.save_map()
    n_first = a.num_voxels()
    b = la.NDT(min_pts_in_voxel=3, capacity=150)
    b.load_map(blob)
    assert b.num_voxels() == n_first > 150                                  # nothing is dropped by the load itself
    new = (r.uniform([0, 0, 0], [5, 5, 2], (2000, 3)) + [40, 0, 0]).astype(np.float32)   # 50 cells elsewhere
    b.add_target(new)
    assert b.num_voxels() == 149
    kept = set(map(tuple, la.unpack_keys(b.export_voxels()["keys"]).tolist()))
    assert {tuple(int(v) for v in np.floor(p)) for p in new} <= kept
    # of the loaded voxels the ones latest in file order (ascending key) stay
    old = sorted(k for k in kept if k[0] < 40)
    want = sorted(map(tuple, la.unpack_keys(np.sort(a.export_voxels()["keys"])[-len(old):]).tolist()))
    assert old == want
    a.close(); b.close()
