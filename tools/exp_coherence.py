"""Experiment: residual-kernel time against the order of the source points (as scanned / sorted by sensor-frame cell)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)

def morton(c):
    c = c.astype(np.uint64); out = np.zeros(len(c), np.uint64)
    for b in range(12):
        for a in range(3):
            out |= ((c[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return out

def order(p, mode, leaf=1.0):
    if mode == "asis": return p
    c = np.floor(p[:, :3] / leaf).astype(np.int64); c -= c.min(0)
    if mode == "lex": k = (c[:, 2] * 4096 + c[:, 1]) * 4096 + c[:, 0]
    elif mode == "morton": k = morton(c)
    elif mode == "random": k = np.random.default_rng(0).permutation(len(p))
    return p[np.argsort(k, kind="stable")]

for mode in ("asis", "lex", "morton", "morton2", "random"):
    leaf = 2.0 if mode == "morton2" else 1.0
    m = "morton" if mode == "morton2" else mode
    ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    ndt.set_target(w["map"]); ndt.set_source(order(w["scan"], m, leaf))
    r = ndt.align(w["guess"])
    us1 = ndt.time_residual(w["guess"], 200)
    ndt.set_batch_sources([order(w[f"scan{j}"], m, leaf) for j in range(8)])
    ndt.align_batch_resident([w[f"guess{p % 8}"] for p in range(16)], [p % 8 for p in range(16)])
    usb, pts, hp = ndt.time_residual_batch(100)
    print(f"{mode:8s} single {us1:7.2f} us   batch {usb:7.2f} us  ({pts:.0f} pts, {hp:.2f} hits/pt)  chi2 {r['chi2']:.6f}", flush=True)
