"""Batched residual launch alone (one lane of bench.py's default batch: 32 pairs of configs[2]), timed with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NDT_BATCH_LANES"] = "1"
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"])
scans = [w[f"scan{j}"] for j in range(8)]
n.set_batch_sources(scans)
np_ = int(sys.argv[1]) if len(sys.argv) > 1 else 32
r = n.align_batch_resident([w[f"guess{p % 8}"] for p in range(np_)], [p % 8 for p in range(np_)])
us, pts, hits = n.time_residual_batch(100)
print("pairs", np_, "kernel us", us, "points", pts, "hits/pt", hits, "T00", r[0]["T"][0, 0], r[np_ - 1]["T"][1, 3])
