"""Single-scan align timing, several repetitions (min / median of 7 blocks of 200 aligns)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"]); n.set_source(w["scan"])
for _ in range(50): n.align(w["guess"])
ts = []
for b in range(7):
    n.synchronize(); t0 = time.perf_counter()
    for _ in range(200): n.align(w["guess"])
    n.synchronize(); ts.append(1e3 * (time.perf_counter() - t0) / 200)
print("align 117k x 10 it: min %.4f median %.4f ms" % (min(ts), float(np.median(ts))), flush=True)
