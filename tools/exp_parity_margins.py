"""How far inside the parity bars the engine sits: engine against oracle on the configs and on fuzz scenes — H, b, chi2 of a pass,
the aligned pose, the PCL-mode pass and pose.  (Not part of the suite; its output is what the tests' tight bars were set from.)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from conftest import pose_error, rel_err
import lio_slam_amd as la
from lio_slam_amd import synth
from oracle.oracle import Oracle
import test_gpu_fuzz as f

worst = {}
def note(k, v):
    worst[k] = max(worst.get(k, 0.0), float(v))

for idx, small in ((0, False), (1, False), (2, True), (2, False)):
    c = synth.make_config(idx, small=small)
    for nearby in (7, 1):
        n = la.NDT(nearby=nearby, max_iterations=10, eps=0.0); o = Oracle(nearby=nearby, max_iterations=10, eps=0.0)
        n.set_target(c["map"]); o.set_target(c["map"]); n.set_source(c["scan"]); o.set_source(c["scan"])
        for T in (c["guess"], c["T_true"]):
            eg, eo = n.eval(T), o.eval(T)
            note("gn H", rel_err(eg["H"], eo["H"])); note("gn b", rel_err(eg["b"], eo["b"])); note("gn chi2", abs(eg["chi2"] - eo["chi2"]) / abs(eo["chi2"]))
        rg, ro = n.align(c["guess"]), o.align(c["guess"])
        dt, dr = pose_error(rg["T"], ro["T"]); note("gn pose m", dt); note("gn pose rad", dr); note("gn final H", rel_err(rg["H"], ro["H"]))
        vg, vo = n.export_voxels(), o.export_voxels()
        note("map mean", np.abs(vg["mean"] - vo["mean"]).max()); note("map cov rel", (np.abs(vg["cov"] - vo["cov"]) / (np.abs(vo["cov"]).max(axis=(1, 2), keepdims=True) + 1e-300)).max())
        sc = np.abs(vo["info"]).max(axis=(1, 2), keepdims=True) + 1e-300
        note("map info rel", (np.abs(vg["info"] - vo["info"]) / sc).max()); note("map info entries differing", int((vg["info"] != vo["info"]).sum()))
        n.close()
    print("config", idx, "small" if small else "full", {k: f"{v:.2e}" for k, v in worst.items()}, flush=True)

# PCL mode on fuzz scenes
for seed in range(0, 40):
    rng = np.random.default_rng(3000 + seed)
    res = float(rng.choice([0.5, 1.0, 2.0]))
    cfg = dict(resolution=res, max_iterations=int(rng.integers(2, 12)), eps=float(rng.choice([1e-2, 1e-4, 1e-6])))
    prm = dict(outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])), step_size=float(rng.choice([0.05, 0.1, 0.5])) * res, hessian_d1_as_pcl=int(rng.integers(0, 2)))
    pts, origin = f._scene(rng, int(rng.integers(6000, 20000)), extent=float(rng.uniform(10, 25)) * res)
    pts = rng.permutation(pts - origin.astype(np.float32))
    n = la.NDT(**cfg); o = Oracle(**cfg)
    h = len(pts) // 2
    n.set_target(pts[:h]); o.set_target(pts[:h]); n.add_target(pts[h:]); o.add_target(pts[h:])
    T_true = f._se3(rng, 1.0, 0.2)
    sel = rng.choice(len(pts), int(rng.integers(800, 4000)), replace=False)
    src = ((pts[sel].astype(np.float64) - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
    guess = T_true @ f._se3(rng, 0.1 * res, 0.01)
    n.set_source(src); o.set_source(src)
    p = la.pcl_vector_from_matrix(guess)
    a, b = n.pcl_eval(p, **prm), o.pcl_eval(p, **prm)
    note("pcl score", abs(a["score"] - b["score"]) / abs(b["score"])); note("pcl g", rel_err(a["g"], b["g"])); note("pcl H", rel_err(a["H"], b["H"]))
    ra, rb = n.pcl_align(guess, **prm), o.pcl_align(guess, **prm)
    if ra["iterations"] == rb["iterations"] and ra["evaluations"] == rb["evaluations"]:
        dt, dr = pose_error(ra["T"], rb["T"]); note("pcl pose m", dt); note("pcl pose rad", dr)
    else:
        note("pcl branch flips", worst.get("pcl branch flips", 0) + 1)
    n.close()
print("all", {k: f"{v:.2e}" for k, v in worst.items()})
