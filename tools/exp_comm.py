"""Cost of the sharded loop's collective with ONE rank (what the machinery itself adds): configs[2] scan, 10 iterations.
none / RCCL one-rank communicator / peer exchange with one rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
def timed(ndt, reps=300):
    for _ in range(20): ndt.align(w["guess0"])
    ndt.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ndt.align(w["guess0"])
    ndt.synchronize(); return 1e6 * (time.perf_counter() - t0) / reps
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"]); ndt.set_source(w["scan0"])
print("no collective        us per align", timed(ndt))
ndt.comm_init(0, 1, la.NDT.comm_unique_id())
print("RCCL, one rank       us per align", timed(ndt))
ndt.comm_destroy()
h = ndt.comm_p2p_export(); ndt.comm_p2p_init(0, 1, [h])
print("peer exchange, one rank us per align", timed(ndt))
ndt.comm_destroy()
