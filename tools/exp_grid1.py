"""Single-source align time vs source size (grid cap given by the environment)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
m, s = bench.load_dense_workload(w, 0, 1, None)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"])
out = []
for n_pts in (117_594, 250_000, 500_000, 1_000_000, len(s)):
    n.set_source(s[:n_pts])
    n.align(w["guess"]); n.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): n.align(w["guess"])
    n.synchronize()
    out.append("%d: %.3f ms" % (n_pts, 1e3 * (time.perf_counter() - t0) / 30))
print("cap", os.environ.get("NDT_DEBUG_CAP"), "chain", "off" if os.environ.get("NDT_NO_CHAIN") else "on", " | ".join(out), flush=True)
