"""Side table for BASELINE.md §5: configs[0] and configs[1] (VLP-16-shaped 30 k-point scans), the FULL per-scan job the
configs name — set_target (map build) + set_source + align from the perturbed pose until convergence (eps 1e-2, <= 35
iterations) — on the engine and on the CPU oracle (the checker, timed here as the baseline leg).  One JSON line per config.
Not part of bench.py's contract."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lio_slam_amd as la
from lio_slam_amd import synth
from oracle.oracle import Oracle


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


def main():
    for index in (0, 1):
        c = synth.make_config(index)
        cfg = dict(nearby=7, max_iterations=35, eps=1e-2)
        ndt = la.NDT(**cfg)

        def gpu_job():
            ndt.set_target(c["map"]); ndt.set_source(c["scan"]); return ndt.align(c["guess"])

        def gpu_align():
            return ndt.align(c["guess"])
        med(gpu_job, 5)
        t_job, r = med(gpu_job, 30)
        t_align, _ = med(gpu_align, 30)
        out = {"config": index, "source_points": int(len(c["scan"])), "map_points": int(len(c["map"])), "voxels": int(ndt.num_voxels()),
               "gpu": {"job_ms": 1e3 * t_job, "align_ms": 1e3 * t_align, "iterations": int(r["iterations"]), "scans_per_s_incl_map_build": 1 / t_job,
                       "pts_iters_per_s_align": len(c["scan"]) * r["iterations"] / t_align,
                       "pose_error_m": float(np.linalg.norm((np.linalg.inv(c["T_true"]) @ r["T"])[:3, 3]))}}
        ndt.close()
        for threads in (1, min(16, os.cpu_count() or 1)):
            o = Oracle(threads=threads, **cfg)

            def cpu_job():
                o.set_target(c["map"]); o.set_source(c["scan"]); return o.align(c["guess"])
            t_cjob, ro = med(cpu_job, 3)
            t_calign, _ = med(lambda: o.align(c["guess"]), 3)
            out[f"cpu{threads}"] = {"job_ms": 1e3 * t_cjob, "align_ms": 1e3 * t_calign, "iterations": int(ro["iterations"]),
                                    "scans_per_s_incl_map_build": 1 / t_cjob,
                                    "pts_iters_per_s_align": len(c["scan"]) * ro["iterations"] / t_calign, "cores": threads, "kind": "port",
                                    "pose_vs_gpu_m": float(np.linalg.norm(ro["T"][:3, 3] - r["T"][:3, 3]))}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
