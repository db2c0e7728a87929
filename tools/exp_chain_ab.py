"""A/B of library builds on the single-scan loop: python tools/exp_chain_ab.py <lib.so> [<lib.so> ...]
Each library runs in a process of its own (NDT_AB_LARGE=1: the 20 M-point build too); prints the chained launch, the residual pass alone, the 10-iteration align and the
2 M-point build / 118 k-point merge, three interleaved rounds so that clock drift shows."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import lio_slam_amd as la
la.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
n.set_target(w["map"]); n.set_source(w["scan0"])
for _ in range(30): n.align(w["guess0"])
ch = 0.5 * (n.time_chain(w["guess0"], 400) + n.time_chain(w["T0"], 400))
rs = 0.5 * (n.time_residual(w["guess0"], 400) + n.time_residual(w["T0"], 400))
ts = []
for b in range(5):
    n.synchronize(); t0 = time.perf_counter()
    for _ in range(200): n.align(w["guess0"])
    n.synchronize(); ts.append(1e3 * (time.perf_counter() - t0) / 200)
d = bench.DeviceCloud(w["map"])
bt = []
for _ in range(12):
    t0 = time.perf_counter(); n.set_target(d.view()); n.synchronize(); bt.append(1e3 * (time.perf_counter() - t0))
n.set_source(w["scan0"])
mt = []
for _ in range(15):
    t0 = time.perf_counter(); n.add_target_from_source(w["T0"]); n.synchronize(); mt.append(1e3 * (time.perf_counter() - t0))
big = ""
if os.environ.get("NDT_AB_LARGE"):
    lm = bench.large_map_workload()
    n2 = la.NDT(max_iterations=10, eps=0.0, final_eval=0)
    d2 = bench.DeviceCloud(lm["map"])
    lt = []
    for _ in range(6):
        t0 = time.perf_counter(); n2.set_target(d2.view()); n2.synchronize(); lt.append(1e3 * (time.perf_counter() - t0))
    big = "  build20M min %%.4f ms" %% min(lt[2:])
print("%%-28s chain %%.3f us  residual %%.3f us  align min %%.4f med %%.4f ms  build2M min %%.4f ms  merge min %%.4f ms" %% (
    os.path.basename(sys.argv[1]), ch, rs, min(ts), float(np.median(ts)), min(bt[3:]), min(mt[3:])) + big, flush=True)
''' % ROOT
for rnd in range(3):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, timeout=600)
        sys.stdout.write(r.stdout if r.returncode == 0 else f"{lib}: FAILED\n{r.stderr[-1500:]}\n")
        sys.stdout.flush()
