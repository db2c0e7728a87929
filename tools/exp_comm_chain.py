"""One-rank RCCL path at the per-rank size of an 8-way shard (250 k points): chained vs four launches per iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
m, s = bench.load_dense_workload(w, 0, 1, None)
for n_pts in (250_000, 117_594):
    n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    n.set_target(w["map"]); n.set_source(s[:n_pts])
    def t(reps=50):
        n.align(w["guess"]); n.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): n.align(w["guess"])
        n.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps
    a = t()
    n.comm_init(0, 1, la.NDT.comm_unique_id())
    b = t()
    print("%d pts: no comm %.3f ms, comm(1 rank) %.3f ms  [NDT_NO_CHAIN=%s]" % (n_pts, a, b, os.environ.get("NDT_NO_CHAIN")), flush=True)
    n.comm_destroy(); n.close()
