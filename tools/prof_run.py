"""Short profiling driver, run under rocprofv3 (tools/gpu_prof.sh).  Workloads are bench.py's own (cached under /tmp).
   prof_run.py online N B : N aligns of configs[2] scan0 (k_residual_chain x iterations + one k_reduce_solve each), N x 2
                            single passes (k_residual), B batched aligns of one launch lane of bench.py's default batch
   prof_run.py large N    : N x 2 single passes + 1 align of the 2M-pt source against the 20M-pt / >500k-voxel map
   prof_run.py dense N    : the same passes on the 2M-pt source against the configs[2] voxel map replicated 10 x 10 (2.2M voxels)
   prof_run.py buildlarge N : N rebuilds of the 20M-pt map from HBM-resident input
   prof_run.py build N    : N rebuilds of the 2M-pt map from HBM-resident input + N keyframe merges of the resident source
Writes gpurun_out/prof_meta_<mode>.json: points per launch of every kernel the summary reports on."""
import json, os, sys
os.environ["NDT_BATCH_LANES"] = "1"   # one launch lane: kernels run alone, so the trace's durations are those of bench.py's HIP-event timing
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, lio_slam_amd as la
mode = sys.argv[1]
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
meta = {}
if mode == "online":
    w = bench.load_workload(2, 0, 1, None, False)
    ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    ndt.set_target(w["map"]); ndt.set_source(w["scan0"])
    n1 = ndt.source_size()
    for _ in range(n_rep):
        r = ndt.align(w["guess0"])
    for _ in range(n_rep):
        ndt.eval(w["guess0"]); ndt.eval(r["T"])
    meta["k_residual_chain<7>"] = {"match": "k_residual_chain<7, true>", "points_per_launch": n1}
    meta["k_residual<7>"] = {"match": "k_residual<7, false, true>", "points_per_launch": n1}
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if nb:
        scans = [w[f"scan{j}"] for j in range(8)]
        ndt.set_batch_sources(scans)
        half = bench.DEFAULT_BATCH // 2
        for _ in range(nb):
            ndt.align_batch_resident([w[f"guess{p % 8}"] for p in range(half)], [p % 8 for p in range(half)])
        meta["k_residual_batch<7>"] = {"match": "k_residual_batch<7, true>", "points_per_launch": int(sum(len(scans[p % 8]) for p in range(half)))}
    print("done", r["iterations"], ndt.num_voxels())
elif mode == "large":
    lm = bench.large_map_workload()
    ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    ndt.set_target(lm["map"]); ndt.set_source(lm["scan"])
    r = ndt.align(lm["guess"])
    for _ in range(n_rep):
        ndt.eval(lm["guess"]); ndt.eval(r["T"])
    meta["k_residual<7>@large_map"] = {"match": "k_residual<7, false, true>", "points_per_launch": ndt.source_size()}
    print("done", r["iterations"], ndt.num_voxels())
elif mode == "dense":
    import numpy as np
    from lio_slam_amd import synth
    w = bench.load_workload(2, 0, 1, None, False)
    ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    ndt.set_target(w["map"]); ndt.load_map(bench.replicate_map_blob(ndt.save_map(), 10, 10, 300))
    T0 = w["T0"]; sub = w["scan0"][::6]
    parts = []
    for ix in range(10):
        for iy in range(10):
            wp = synth.transform(T0, sub) + np.array([300.0 * ix, 300.0 * iy, 0.0], np.float32)
            parts.append(((wp.astype(np.float64) - T0[:3, 3]) @ T0[:3, :3]).astype(np.float32))
    ndt.set_source(np.concatenate(parts))
    gd = T0 @ synth.se3([6e-6, -5e-6, 6e-6], [0.07, -0.06, 0.03])
    r = ndt.align(gd)
    for _ in range(n_rep):
        ndt.eval(gd); ndt.eval(r["T"])
    meta["k_residual<7>@large_dense"] = {"match": "k_residual<7, false, true>", "points_per_launch": ndt.source_size()}
    print("done", r["iterations"], ndt.num_voxels())
elif mode == "build":
    w = bench.load_workload(2, 0, 1, None, False)
    ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    d = bench.DeviceCloud(w["map"])
    for _ in range(n_rep):
        ndt.set_target(d.view()); ndt.synchronize()
    ndt.set_source(w["scan0"])
    for _ in range(n_rep):
        ndt.add_target_from_source(w["T0"]); ndt.synchronize()
    meta["_build"] = {"map_points": len(w["map"]), "keyframe_points": ndt.source_size(), "rebuilds": n_rep, "merges": n_rep}
    print("done", ndt.num_voxels())
elif mode == "buildlarge":
    lm = bench.large_map_workload()
    ndt = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    d = bench.DeviceCloud(lm["map"])
    for _ in range(n_rep):
        ndt.set_target(d.view()); ndt.synchronize()
    meta["_buildlarge"] = {"map_points": len(lm["map"]), "rebuilds": n_rep, "voxels": ndt.num_voxels(), "radix_passes": ndt.last_radix_passes()}
    print("done", ndt.num_voxels())
os.makedirs("gpurun_out", exist_ok=True)
json.dump(meta, open(f"gpurun_out/prof_meta_{mode}.json", "w"))
