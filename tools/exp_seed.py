"""Diagnoses one seed of tests/test_gpu_fuzz.py::test_random_scene_and_config: per Gauss-Newton iteration, engine vs oracle."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as f
import lio_slam_amd as la
from oracle.oracle import Oracle
from conftest import pose_error, rel_err
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
res = float(rng.choice([0.5, 0.8, 1.0, 1.5, 3.0]))
cfg = dict(resolution=res, key_rule=int(rng.integers(0, 2)), nearby=int(rng.choice([1, 7])), min_pts_in_voxel=int(rng.integers(3, 9)),
           res_outlier_th=float(rng.choice([5.0, 20.0, 100.0])), eig_floor_ratio=float(rng.choice([1e-3, 1e-2, 5e-2])),
           max_iterations=int(rng.integers(3, 16)), eps=float(rng.choice([0.0, 1e-3, 1e-2])), min_effective_pts=int(rng.choice([10, 50])))
print(cfg)
pts, origin = f._scene(rng, int(rng.integers(4000, 30000)), extent=float(rng.uniform(10, 40)) * res)
pts = rng.permutation(pts)
bad = rng.choice(len(pts), 40, replace=False)
pts[bad[:15], rng.integers(0, 3, 15)] = np.nan
pts[bad[15:25], rng.integers(0, 3, 10)] = -np.inf
pts[bad[25:35]] = pts[(bad[25:35] + 7) % len(pts)]
pts[bad[35]] = [4e6 * res, 0, 0]
pts[bad[36]] = np.float32(np.round(origin / res) * res)
n = la.NDT(**cfg); o = Oracle(**cfg)
cut = sorted(rng.choice(np.arange(1, len(pts)), 2, replace=False))
n.set_target(pts[: cut[0]]); o.set_target(pts[: cut[0]])
Tm = f._se3(rng, 0.4, 0.02)
n.add_target(pts[cut[0]: cut[1]]); o.add_target(pts[cut[0]: cut[1]])
n.add_target(pts[cut[1]:], Tm); o.add_target(pts[cut[1]:], Tm)
T_true = f._se3(rng, 2.0, 0.3); T_true[:3, 3] += origin
sel = rng.choice(len(pts), min(len(pts), int(rng.integers(500, 6000))), replace=False)
world = pts[sel].astype(np.float64)
with np.errstate(invalid="ignore"):
    src = ((world - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
guess = T_true @ f._se3(rng, 0.15 * res, 0.01)
half = len(src) // 2
g3 = [guess, T_true, T_true @ f._se3(rng, 0.1 * res, 0.005)]
vg, vo = n.export_voxels(), o.export_voxels()
v = vo["valid"].astype(bool)
print("voxels", len(vo["keys"]), "valid", int(v.sum()), "src", len(src))
rb = n.align_batch([src[:half], src[half:]], g3, [0, 1, 0])
for p, (g, sc) in enumerate(zip(g3, (src[:half], src[half:], src[:half]))):
    n.set_source(sc); o.set_source(sc)
    ra, ro = n.align(g, check=False), o.align(g)
    print("pair", p, "single rc", ra["rc"], ro["rc"], "it", ra["iterations"], ro["iterations"], "batch it", rb[p]["iterations"], "status", rb[p]["status"],
          "single-vs-oracle", pose_error(ra["T"], ro["T"]), "batch-vs-oracle", pose_error(rb[p]["T"], ro["T"]))
    Tg, To = g.copy(), g.copy()
    n.configure(max_iterations=1, eps=0.0); o.set_params(max_iterations=1, eps=0.0)
    for it in range(cfg["max_iterations"]):
        eg, eo = n.eval(Tg), o.eval(To)
        lam = np.linalg.eigvalsh(eo["H"]) if eo["n_effective"] else np.ones(6)
        rg, r1 = n.align(Tg, check=False), o.align(To)
        Tg, To = rg["T"], r1["T"]
        dt, dr = pose_error(Tg, To)
        print("   it", it, "neff", eo["n_effective"], eg["n_effective"], "cond %.2e" % (lam[-1] / max(abs(lam[0]), 1e-300)), "relH %.1e relb %.1e" % (rel_err(eg["H"], eo["H"]), rel_err(eg["b"], eo["b"])),
              "dt %.2e dr %.2e" % (dt, dr), "dx %.2e" % r1["last_dx_norm"], "rc", rg["rc"], r1["rc"])
    n.configure(max_iterations=cfg["max_iterations"], eps=cfg["eps"]); o.set_params(max_iterations=cfg["max_iterations"], eps=cfg["eps"])
