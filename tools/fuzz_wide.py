"""Randomised differential sweep over the options the suite's fuzz test leaves fixed: IncNdt3d's info rule / frozen voxels / LRU
capacity, merges through poses, the device-side downsample as the source, replay batches, map save / load, the static entry —
engine against oracle at the tight bars of tests/conftest.py.  python tools/fuzz_wide.py <lo> <hi>   (not part of the suite)"""
import ctypes, os, sys, time, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as f
from conftest import pose_error, rel_err, TIGHT_H, TIGHT_T, TIGHT_R
import lio_slam_amd as la
from oracle.oracle import Oracle, voxel_downsample


def cmp_maps(vg, vo, res):
    assert np.array_equal(vg["keys"], vo["keys"]) and np.array_equal(vg["counts"], vo["counts"]) and np.array_equal(vg["valid"], vo["valid"])
    assert np.abs(vg["mean"] - vo["mean"]).max() < 1e-12 * max(1.0, res)
    assert np.max(np.abs(vg["cov"] - vo["cov"]) / np.maximum(np.abs(vo["cov"]).max(axis=(1, 2), keepdims=True), 1e-300)) < 1e-10
    v = vo["valid"].astype(bool)
    if v.any():
        d = np.abs(vg["info"][v] - vo["info"][v]) / np.abs(vo["info"][v]).max(axis=(1, 2), keepdims=True)
        assert d.max() < 5e-7 and np.mean(d > 1e-10) < 1e-3, (d.max(), np.mean(d > 1e-10))


def one(seed):
    rng = np.random.default_rng(7000 + seed)
    res = float(rng.choice([0.5, 1.0, 2.0]))
    cfg = dict(resolution=res, key_rule=int(rng.integers(0, 2)), nearby=int(rng.choice([1, 7])), min_pts_in_voxel=int(rng.integers(3, 8)),
               info_rule=int(rng.integers(0, 2)), max_pts_in_voxel=int(rng.choice([0, 0, 30, 80])), res_outlier_th=float(rng.choice([20.0, 100.0])),
               max_iterations=int(rng.integers(4, 12)), eps=float(rng.choice([0.0, 1e-3])), min_effective_pts=50, final_eval=int(rng.integers(0, 2)))
    use_cap = rng.random() < 0.25
    pts, origin = f._scene(rng, int(rng.integers(20000, 60000)), extent=float(rng.uniform(20, 40)) * res)
    pts = rng.permutation(pts)
    if use_cap:
        cfg["capacity"] = int(rng.integers(200, 2000))
    n = la.NDT(**cfg); o = Oracle(**cfg)
    cuts = sorted(rng.choice(np.arange(1000, len(pts) - 1000), 3, replace=False))
    n.set_target(pts[: cuts[0]]); o.set_target(pts[: cuts[0]])
    for a, b in zip(cuts, cuts[1:] + [len(pts)]):
        Tm = f._se3(rng, 0.3, 0.02) if rng.random() < 0.5 else None
        if Tm is not None:
            Tm[:3, 3] += origin - Tm[:3, :3] @ origin            # a small motion about the scene, not about the world origin
        n.add_target(pts[a:b], Tm); o.add_target(pts[a:b], Tm)
    vg, vo = n.export_voxels(), o.export_voxels()
    cmp_maps(vg, vo, res)
    # source: raw scan through the device-side downsample, or a plain subset
    T_true = f._se3(rng, 2.0, 0.3); T_true[:3, 3] += origin
    sel = rng.choice(len(pts), int(rng.integers(3000, 12000)), replace=False)
    src = ((pts[sel].astype(np.float64) - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
    if rng.random() < 0.4:
        leaf = 0.5 * res
        n.set_source_downsampled(src, leaf, None)
        src = n.get_source()                                      # the same source on both sides from here on
        ref = voxel_downsample(src if False else ((pts[sel].astype(np.float64) - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32), leaf, None)
        assert src.shape == ref.shape and np.max(np.abs(src.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)) < 2.5e-7
        o.set_source(src)
    else:
        n.set_source(src); o.set_source(src)
    guess = T_true @ f._se3(rng, 0.1 * res, 0.01)
    kg, mg = n.source_cells(guess); ko, mo = o.source_cells(guess)
    assert np.array_equal(kg, ko) and np.array_equal(mg, mo)
    eg, eo = n.eval(guess), o.eval(guess)
    assert eg["n_effective"] == eo["n_effective"]
    if eo["n_effective"] < 200:
        n.close(); return "sparse"
    assert rel_err(eg["H"], eo["H"]) < TIGHT_H and rel_err(eg["b"], eo["b"]) < 1e-8 and abs(eg["chi2"] - eo["chi2"]) <= TIGHT_H * abs(eo["chi2"])
    rg, ro = n.align(guess, check=False), o.align(guess)
    assert rg["rc"] == ro["rc"]
    if ro["rc"] == 0:
        w = np.linalg.eigvalsh(ro["H"]); cond = w[-1] / max(w[0], 1e-300)
        dt, dr = pose_error(rg["T"], ro["T"])
        assert rg["iterations"] == ro["iterations"] and rg["converged"] == ro["converged"]
        bar = 1e-9 if cond < 1e6 else 1e-6
        assert dt < bar and dr < bar, (dt, dr, cond)
        assert rel_err(rg["H"], ro["H"]) < 1e-8
        # the static entry (device SoA arrays) and a replay batch give the same answer as the plain call
        hip = ctypes.CDLL("libamdhip64.so")
        soa = np.ascontiguousarray(n.get_source().T)
        d = ctypes.c_void_p(); nb = soa.nbytes
        assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(nb)) == 0
        assert hip.hipMemcpy(d, ctypes.c_void_p(soa.ctypes.data), ctypes.c_size_t(nb), 1) == 0
        m = soa.shape[1]
        rs = n.align_static(d.value, d.value + 4 * m, d.value + 8 * m, m, guess, check=False)
        assert hip.hipFree(d) == 0
        if rs["rc"] == 0:
            assert np.array_equal(rs["T"], rg["T"]) and rs["iterations"] == rg["iterations"], "static entry differs"
        rb = n.align_batch([n.get_source()], [guess, guess], [0, 0])
        dt, dr = pose_error(rb[0]["T"], ro["T"])
        assert rb[0]["status"] == 0 and dt < bar and dr < bar and np.array_equal(rb[0]["T"], rb[1]["T"])
    # save / load: the loaded map answers with the same bits
    blob = n.save_map()
    n2 = la.NDT(**cfg); n2.load_map(blob); n2.set_source(n.get_source())
    e2 = n2.eval(guess); e1 = n.eval(guess)
    assert np.array_equal(e1["H"], e2["H"]) and np.array_equal(e1["b"], e2["b"])
    n2.close(); n.close()
    return "ok"


lo, hi = int(sys.argv[1]), int(sys.argv[2])
fails = 0; t0 = time.time(); kinds = {}
for seed in range(lo, hi):
    try:
        k = one(seed); kinds[k] = kinds.get(k, 0) + 1
    except Exception:
        fails += 1
        print("FAIL", seed, flush=True); traceback.print_exc(limit=2)
    if seed % 20 == 0:
        print("seed", seed, "elapsed %.0f s" % (time.time() - t0), "fails", fails, kinds, flush=True)
print("done", lo, hi, "fails", fails, kinds)
