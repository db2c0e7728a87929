"""Replay step time round by round in one process (bench.py's replay leg): is the 1.52 / 1.67 ms spread per process or in time?
usage: python tools/exp_replay.py [rounds] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lio_slam_amd as la  # noqa: E402
from lio_slam_amd import synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = int(os.environ.get("REPLAY_BATCH", "64"))
sys.argv = sys.argv[:1]
import bench  # noqa: E402  (its workload loader: the same scans and guesses as the bench's replay leg)
w = bench.load_workload(2, 0, 1, None, False)
rng = np.random.Generator(np.random.PCG64(synth.seed_for(2) + 77))
scans = [w[f"scan{j}"] for j in range(min(8, batch))]
pair_scan = np.asarray([p % len(scans) for p in range(batch)], np.int32)
guesses = np.stack([w[f"guess{p % len(scans)}"] if p < len(scans) else w[f"T{p % len(scans)}"] @ synth.perturbation(rng) for p in range(batch)])
nr = la.NDT(device=0, nearby=7, max_iterations=10, eps=0.0, final_eval=0)
nr.set_target(w["map"])
nr.set_batch_sources(scans)
for _ in range(3):
    nr.align_batch_resident(guesses, pair_scan)
nr.synchronize()
out = []
for r in range(rounds):
    t0 = time.perf_counter()
    for _ in range(steps):
        nr.align_batch_resident(guesses, pair_scan)
    nr.synchronize()
    out.append(1e3 * (time.perf_counter() - t0) / steps)
print("ms per step by round:", " ".join(f"{x:.3f}" for x in out))
