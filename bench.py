#!/usr/bin/env python3
"""bench.py — NDT scan-matching throughput on MI355X (BASELINE.json metric).

Workload (configs[2], the configuration the metric is quoted on): HDL-64-shaped scans (~118k points each)
aligned against a GPU-resident voxel map built from a 2M-point keyframe map, NEARBY7, 1.0 m voxels, each
from a perturbed ("IMU-predicted") pose with a fixed number of Gauss-Newton iterations (eps = 0, so no
work can be skipped).  One step = one batched align of `--batch` independent scan-to-map pairs (offline
replay; all scans resident in HBM).  value = source points x Gauss-Newton iterations per second, whole job.
The online case (one scan at a time, what the SLAM pipeline does per frame) is timed too and reported in
"single_scan" with its own roofline.

N>1 (`torch.distributed.run`, one process per GPU): every rank holds a replica of the map and replays its
own batch; no data-path collective; weak scaling.  `--mode sharded` instead shards one source cloud over
the ranks with an RCCL all-reduce of the 29 H/b/chi2/n terms per iteration (configs[3]).

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_BATCH = 64      # pairs per step per GPU: configs[4] replays 4096 pairs over 8 GPUs (512 per GPU), a step is a slice of that
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


@contextlib.contextmanager
def stdout_to_stderr():
    """File descriptor 1 pointed at stderr for the duration: native libraries (gloo, RCCL) that announce themselves on
    stdout must not get in front of the one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def algorithmic_bytes_per_point(k, h):
    """B_res = 12 (xyz) + 8 k (key slot per probe) + 40 h (payload per valid hit)."""
    return 12.0 + 8.0 * k + 40.0 * h


def measured_traffic(kernel, points_per_launch, field="bytes_per_launch"):
    """HBM bytes per launch of `kernel` (or its L2 hit rate) from the committed rocprofv3 PMC passes
    (profiles/traffic_r01.json); None when the file holds no measurement of a launch of this size."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
        return d[kernel][field] if d[kernel]["points_per_launch"] == int(points_per_launch) else None
    except (OSError, KeyError, ValueError):
        return None


def load_workload(index, rank, world, dist, small):
    """configs[index] inputs; generated once (rank 0) and cached under /tmp for the other ranks / later runs."""
    from lio_slam_amd import synth
    path = f"/tmp/ndt_bench_c{index}{'_small' if small else ''}.npz"
    if rank == 0 and not os.path.exists(path):
        c = synth.make_config(index, small=small)
        extra = {}
        if index == 2:   # a few more scans along the trajectory: keyframes to merge, replay scans for other ranks
            rng = np.random.Generator(np.random.PCG64(synth.seed_for(index) + 1000))
            n = len(c["poses"])
            for j in range(8):
                T = c["poses"][(n // 2 + 1 + j) % n].copy()
                T[:3, 3] += rng.uniform(-0.2, 0.2, 3) * [1, 1, 0]
                extra[f"scan{j}"] = synth.scan(c["scene"], "hdl64", T, rng)
                extra[f"T{j}"] = T
                extra[f"guess{j}"] = T @ synth.perturbation(rng)
        tmp = path + f".{os.getpid()}.npz"
        np.savez(tmp, map=c["map"], scan=c["scan"], T_true=c["T_true"], guess=c["guess"], **extra)
        os.replace(tmp, path)
    if world > 1:
        dist.barrier()
    return dict(np.load(path))


def load_dense_workload(w, rank, world, dist):
    """configs[3] inputs from the configs[2] pieces: the same yard and sensor pose, a dense 128-ring scan of ~2 M returns
    as the source, and a 20 M-point map (the 2 M-point keyframe map resampled ten times with the sensor's 1 cm range
    noise: ten times the sampling density on the same surfaces).  Generated once (rank 0), cached under /tmp."""
    from lio_slam_amd import synth
    path = "/tmp/ndt_bench_c3.npz"
    if rank == 0 and not os.path.exists(path):
        rng = np.random.Generator(np.random.PCG64(synth.seed_for(2)))
        scene = synth.yard_scene(rng)                  # make_config(2) draws the scene first: this is the same yard
        rng = np.random.Generator(np.random.PCG64(synth.seed_for(3)))
        src = synth.scan(scene, "dense128", w["T_true"], rng)
        m = np.concatenate([w["map"]] + [(w["map"] + rng.normal(0, 0.01, w["map"].shape)).astype(np.float32) for _ in range(9)])
        tmp = path + f".{os.getpid()}.npz"
        np.savez(tmp, map=m, scan=src)
        os.replace(tmp, path)
    if world > 1:
        dist.barrier()
    d = np.load(path)
    return d["map"], d["scan"]


def time_resident_build(ndt, pts, reps=5):
    """Map build with the target cloud already in HBM (handed over as a device pointer): the kernels and their host
    synchronisations without the PCIe copy.  Returns (ms, radix passes)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    a = np.ascontiguousarray(pts, np.float32)
    d = ctypes.c_void_p()
    if hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(a.nbytes)) != 0:
        return None, 0
    try:
        if hip.hipMemcpy(d, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1) != 0:
            return None, 0
        best = None
        for _ in range(reps + 1):
            t0 = time.perf_counter(); ndt.set_target((d.value, len(a), 12)); ndt.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
            best = ms if best is None else min(best, ms)
        return best, ndt.last_radix_passes()
    finally:
        hip.hipFree(d)


def build_roofline(n_points, n_voxels, passes, ms):
    """B_build = 12 (read xyz) + 12 (write key + index) + 24 P (radix passes) + 12 (gather) per target
    point + 40 per output voxel, against the HBM peak, over the whole resident-input build (all its kernels and syncs)."""
    if not ms:
        return None
    algo = n_points * (36.0 + 24.0 * passes) + 40.0 * n_voxels
    ach = algo / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "ms": ms, "radix_passes": passes,
            "bytes_per_point": 36.0 + 24.0 * passes, "points": int(n_points), "what": "whole ndt_set_target call, target cloud resident in HBM"}


def time_cpu_baseline(w, iters, nearby, budget_s=12.0):
    """The CPU oracle (a port of the same algorithm, oracle/ndt_oracle.c) on this box's host cores."""
    from oracle.oracle import Oracle
    out = {}
    o = Oracle(nearby=nearby, max_iterations=iters, eps=0.0, final_eval=0)
    t0 = time.perf_counter(); o.set_target(w["map"]); build_s = time.perf_counter() - t0
    o.set_source(w["scan"])
    n = o.source_size()
    for name, threads in (("cpu_baseline", 1), ("cpu_baseline_mt", min(16, os.cpu_count() or 1))):
        o.set_threads(threads)
        o.align(w["guess"])                      # warm
        reps, t0 = 0, time.perf_counter()
        while True:
            o.align(w["guess"]); reps += 1
            el = time.perf_counter() - t0
            if el > budget_s / 2 or reps >= 200:
                break
        out[name] = {"value": n * iters * reps / el, "unit": "source-points*iters/s", "cores": threads, "kind": "port",
                     "sample": f"{reps} aligns x {iters} GN iterations of the same {n}-pt scan vs 2M-pt map "
                               f"({el:.1f} s of CPU work)", "scans_per_s": reps / el}
    out["cpu_baseline"]["map_build_s"] = build_s
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--iters", type=int, default=10, help="Gauss-Newton iterations per align (fixed)")
    ap.add_argument("--nearby", type=int, default=7)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="scan-to-map pairs aligned together per step (replay mode)")
    ap.add_argument("--mode", choices=["replay", "sharded"], default="replay")
    ap.add_argument("--small", action="store_true", help="200k-pt map (quick functional run, not a valid number)")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with one rank)
    if use_dist:
        # rendezvous / barriers / timing reductions on CPU tensors (gloo); the data-path collective, where the
        # mode has one, is RCCL inside libndt_mi355.so on the handle's own stream
        import torch
        import torch.distributed as dist
        # gloo's C++ layer announces its connections on stdout; rank 0's stdout is for the ONE JSON line, so the
        # rendezvous runs with file descriptor 1 pointed at stderr
        with stdout_to_stderr():
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        ndev = torch.cuda.device_count()          # counting devices does not initialise the GPU
        if ndev:
            local %= ndev                         # fewer visible devices than ranks (a rehearsal on a one-GPU box): share them
    import lio_slam_amd as la
    from lio_slam_amd import synth

    w = load_workload(2, rank, world if use_dist else 1, dist, a.small)
    ndt = la.NDT(device=local, nearby=a.nearby, max_iterations=a.iters, eps=0.0, final_eval=0)

    def barrier():
        ndt.synchronize()
        if use_dist:
            dist.barrier()

    def allmax(x):
        if not use_dist:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); return float(t[0])

    def allsum(x):
        if not use_dist:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t); return float(t[0])

    t0 = time.perf_counter(); ndt.set_target(w["map"]); ndt.synchronize(); build_ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter(); ndt.set_target(w["map"]); ndt.synchronize(); build_ms = min(build_ms, 1e3 * (time.perf_counter() - t0))
    n_vox = ndt.num_voxels(); n_map = len(w["map"])
    rb_ms, rb_passes = time_resident_build(ndt, w["map"])
    build_rl = build_roofline(len(w["map"]), n_vox, rb_passes, rb_ms)
    merge_ms = []
    for j in range(2):       # incremental keyframe merge (configs[2]): two keyframes into the resident map
        t0 = time.perf_counter(); ndt.add_target(w[f"scan{j}"], w[f"T{j}"]); ndt.synchronize()
        merge_ms.append(1e3 * (time.perf_counter() - t0))

    # ---- the online case: ONE scan aligned at a time (what lio-slam's pipeline does per LiDAR frame)
    single = None
    if a.mode == "replay":
        ndt.set_source(w["scan"])
        n1 = ndt.source_size()
        for _ in range(a.warmup):
            r1 = ndt.align(w["guess"])
        ndt.synchronize()
        reps1 = max(20, a.steps)
        t0 = time.perf_counter()
        for _ in range(reps1):
            r1 = ndt.align(w["guess"])
        ndt.synchronize()
        el1 = time.perf_counter() - t0
        k1 = 0.5 * (ndt.time_residual(w["guess"], 300) + ndt.time_residual(r1["T"], 300))
        ndt.profile(True); ndt.eval(w["guess"]); ndt.eval(r1["T"]); h1 = ndt.profile_read()["hits_per_point"]; ndt.profile(False)
        ach1 = n1 * algorithmic_bytes_per_point(a.nearby, h1) / (k1 * 1e-6) / 1e9
        single = {"value": n1 * a.iters * reps1 / el1, "unit": "source-points*iters/s", "ms_per_align": 1e3 * el1 / reps1,
                  "scans_per_s": reps1 / el1, "source_points": int(n1),
                  "final_pose_error_m": float(np.linalg.norm((np.linalg.inv(w["T_true"]) @ r1["T"])[:3, 3])),
                  "roofline": {"bound": "hbm", "achieved": ach1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach1 / HBM_PEAK_GBS,
                               "traffic": measured_traffic(f"k_residual<{a.nearby}>", n1),
                               "l2_hit_rate": measured_traffic(f"k_residual<{a.nearby}>", n1, "l2_hit_rate"),
                               "kernel": f"k_residual<{a.nearby}>", "kernel_avg_us": k1, "points_per_launch": int(n1),
                               "hits_per_point": h1, "bytes_per_point": algorithmic_bytes_per_point(a.nearby, h1)}}

    # ---- PCL compatibility mode: the objective / optimiser the reference's PCL build runs, on the
    # same scan and map, with the reference's own parameters (process.cc:373-376: <= 35 iterations, eps 0.01, step 0.1)
    pcl = None
    if a.mode == "replay" and world == 1:
        ndt_p = la.NDT(device=local, max_iterations=35, eps=0.01)
        ndt_p.load_map(ndt.save_map()); ndt_p.set_source(w["scan"])
        pvec = la.pcl_vector_from_matrix(w["guess"])
        for _ in range(3):
            rp = ndt_p.pcl_align(w["guess"])
        repsp = max(20, a.steps); t0 = time.perf_counter()
        for _ in range(repsp):
            rp = ndt_p.pcl_align(w["guess"])
        elp = time.perf_counter() - t0
        usp, vxp, nbp = ndt_p.pcl_time_pass(pvec, hessian=True, reps=100)
        bppp = 12 + 8 * 27 + 40 * vxp
        achp = len(w["scan"]) * bppp / (usp * 1e-6) / 1e9
        pcl = {"ms_per_align": 1e3 * elp / repsp, "iterations": rp["iterations"], "evaluations": rp["evaluations"],
               "value": len(w["scan"]) * rp["evaluations"] * repsp / elp, "unit": "source-points*evaluations/s",
               "roofline": {"bound": "hbm", "achieved": achp, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achp / HBM_PEAK_GBS,
                            "kernel": "k_pcl_pass<hessian, gradient>", "kernel_avg_us": usp, "bytes_per_point": bppp,
                            "voxels_met_per_point": vxp, "neighbours_per_point": nbp, "points_per_launch": int(len(w["scan"])),
                            "algorithmic_bytes": "12 + 8*27 + 40*(valid voxels met in the 27 cells)"}}
        if not a.no_cpu:
            from oracle.oracle import Oracle
            o = Oracle(max_iterations=35, eps=0.01); o.set_target(w["map"]); o.set_source(w["scan"])
            t0 = time.perf_counter(); ro = o.pcl_align(w["guess"]); elo = time.perf_counter() - t0
            pcl["cpu_baseline"] = {"value": len(w["scan"]) * ro["evaluations"] / elo, "unit": "source-points*evaluations/s", "cores": 1,
                                   "kind": "port", "ms_per_align": 1e3 * elo,
                                   "sample": f"1 align ({ro['evaluations']} evaluations) of the same scan, oracle's PCL-literal restatement"}
        ndt_p.close()

    # ---- the timed workload
    if a.mode == "sharded":
        from lio_slam_amd.shard import shard_range
        # configs[3]: ONE dense scan (128 rings x 15 625 azimuth steps, ~2 M returns, taken at the same pose in the same
        # yard) against a 20 M-point map of that yard; the source is cut into contiguous ranges, one per rank
        guess, T = w["guess"], w["T_true"]
        big_map, scan_all = load_dense_workload(w, rank, world if use_dist else 1, dist)
        ndt.set_target(big_map); ndt.synchronize()
        t0 = time.perf_counter(); ndt.set_target(big_map); ndt.synchronize(); build_ms = 1e3 * (time.perf_counter() - t0)
        n_vox = ndt.num_voxels(); n_map = len(big_map)
        rb_ms, rb_passes = time_resident_build(ndt, big_map)
        build_rl = build_roofline(n_map, n_vox, rb_passes, rb_ms); del big_map
        lo, hi = shard_range(len(scan_all), rank, world)
        if use_dist:
            with stdout_to_stderr():                             # RCCL prints its version banner on stdout
                ids = [la.NDT.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ndt.comm_init(rank, world, ids[0])
        ndt.set_source(scan_all[lo:hi])                      # resident in HBM before the timed region
        pts_rank = float(ndt.source_size())
        step = lambda: ndt.align(guess)
        check = lambda r: r["iterations"] == a.iters
    else:
        # offline replay of configs[2]-sized pairs: `batch` independent (scan, initial pose) pairs per step, each rank
        # its own scans (8 distinct scans along the trajectory x hypotheses), all resident in HBM
        rng = np.random.Generator(np.random.PCG64(synth.seed_for(2) + 77 + rank))
        scans = [w[f"scan{(j + rank) % 8}"] for j in range(min(8, a.batch))]
        pair_scan = [p % len(scans) for p in range(a.batch)]
        guesses = [w[f"guess{(p % len(scans) + rank) % 8}"] if p < len(scans) else
                   w[f"T{(p % len(scans) + rank) % 8}"] @ synth.perturbation(rng) for p in range(a.batch)]
        guesses = np.stack(guesses); pair_scan = np.asarray(pair_scan, np.int32)
        ndt.set_batch_sources(scans)                          # resident in HBM before the timed region
        pts_rank = float(sum(len(scans[s]) for s in pair_scan))
        step = lambda: ndt.align_batch_resident(guesses, pair_scan)
        check = lambda r: all(x["iterations"] == a.iters and x["status"] == 0 for x in r)

    for _ in range(a.warmup):
        r = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step()
    ndt.synchronize()
    el = allmax(time.perf_counter() - t0)
    assert check(r), "an align did not run its fixed iterations"
    pts_all = allsum(pts_rank)
    value = pts_all * a.iters * a.steps / el

    # roofline of the dominant kernel: HIP events on the engine's stream around back-to-back launches of that
    # kernel alone, right after the timed region (kept outside it so that `value` is undisturbed)
    if a.mode == "sharded":
        k_us = 0.5 * (ndt.time_residual(guess, 300) + ndt.time_residual(r["T"], 300))
        ndt.profile(True); ndt.eval(guess); ndt.eval(r["T"]); h = ndt.profile_read()["hits_per_point"]; ndt.profile(False)
        ppl, kname = pts_rank, f"k_residual<{a.nearby}>"
        err = [float(np.linalg.norm((np.linalg.inv(T) @ r["T"])[:3, 3]))]
    else:
        k_us, ppl, h = ndt.time_residual_batch(100)
        kname = f"k_residual_batch<{a.nearby}>"
        err = [float(np.linalg.norm((np.linalg.inv(w[f"T{(pair_scan[p] + rank) % 8}"]) @ r[p]["T"])[:3, 3])) for p in range(a.batch)]
    bpp = algorithmic_bytes_per_point(a.nearby, h)
    achieved = ppl * bpp / (k_us * 1e-6) / 1e9 if k_us > 0 else 0.0

    out = {
        "metric": "NDT source-points*iters/sec (120k-pt scans vs 2M-pt map)", "value": value, "unit": "source-points*iters/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps,
        "higher_is_better": True, "scaling": "weak" if a.mode == "replay" else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": (("configs[2]: HDL-64-shaped scans (~118k pts) vs the GPU-resident voxel map of a 2M-pt keyframe map"
                                 f"; offline replay, {a.batch} independent scan-to-map pairs per step per GPU") if a.mode == "replay"
                                else ("configs[3]: ONE dense ~2M-point scan sharded over the ranks against the GPU-resident voxel map of a "
                                      "20M-point map (replicated per rank), RCCL all-reduce of H,b per iteration")
                                ) + (" (SMALL 200k-pt map: not a valid number)" if a.small else ""),
                   "mode": a.mode, "pairs_per_step_per_gpu": a.batch if a.mode == "replay" else 1,
                   "source_points_per_step_per_gpu": int(pts_rank), "map_points": int(n_map),
                   "voxels": int(n_vox), "nearby": a.nearby, "resolution_m": 1.0, "gn_iterations_per_align": a.iters,
                   "parallelism": f"{a.mode}{world}"},
        "scans_per_s": a.steps * (a.batch * world if a.mode == "replay" else 1) / el,
        "map_build_ms": build_ms, "map_build_roofline": build_rl, "keyframe_merge_ms": merge_ms,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": measured_traffic(kname, ppl), "l2_hit_rate": measured_traffic(kname, ppl, "l2_hit_rate"),
                     "traffic_source": "profiles/traffic_r01.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE and TCC_HIT / TCC_MISS, separate passes, same workload)",
                     "kernel": kname, "kernel_avg_us": k_us, "bytes_per_point": bpp, "hits_per_point": h,
                     "points_per_launch": int(ppl),
                     "timing": "hipEvents on the engine stream around back-to-back launches of the kernel alone"},
        "final_pose_error_m_max": max(err) if err else None,
        "single_scan": single,
        "pcl_mode": pcl,
    }
    if rank == 0 and world == 1 and not a.no_cpu:
        out.update(time_cpu_baseline(w, a.iters, a.nearby))
        out["gpu_over_cpu1"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    ndt.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
