"""Short profiling driver: configs[2] workload (cached by bench.py under /tmp), N aligns.  Used under rocprofv3."""
import sys, os
os.environ["NDT_BATCH_LANES"] = "1"   # one launch lane: kernels run alone, so the trace's durations are those of bench.py's HIP-event timing
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, lio_slam_amd as la
n_align = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = bench.load_workload(2, 0, 1, None, False)
ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
ndt.set_target(w["map"]); ndt.set_source(w["scan"])
for _ in range(n_align):
    r = ndt.align(w["guess"])          # k_residual_chain x iterations + one k_reduce_solve
for _ in range(n_align):
    ndt.eval(w["guess"]); ndt.eval(r["T"])     # k_residual (the pass bench.py times alone) + k_reduce_solve
if len(sys.argv) > 2:      # also a few batched aligns: one lane of bench.py's default batch (= the launches bench.py times)
    scans = [w[f"scan{j}"] for j in range(8)]
    ndt.set_batch_sources(scans)
    for _ in range(int(sys.argv[2])):
        ndt.align_batch_resident([w[f"guess{p % 8}"] for p in range(bench.DEFAULT_BATCH // 2)], [p % 8 for p in range(bench.DEFAULT_BATCH // 2)])
print("done", r["iterations"], ndt.num_voxels())
