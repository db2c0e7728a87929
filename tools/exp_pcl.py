"""PCL-mode timing on configs[2]: derivative pass, full align under the reference's parameters, CPU oracle beside it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, lio_slam_amd as la
w = bench.load_workload(2, 0, 1, None, False)
n = la.NDT(max_iterations=35, eps=0.01)
n.set_target(w["map"]); n.set_source(w["scan"])
p = la.pcl_vector_from_matrix(w["guess"])
for hess in (True, False):
    us, vx, nb = n.pcl_time_pass(p, hessian=hess, reps=50)
    print(f"pass hessian={hess}: {us:.1f} us, {vx:.2f} voxels/pt, {nb:.2f} neighbours/pt, B={12 + 8 * 27 + 40 * vx:.0f} B/pt -> {len(w['scan']) * (12 + 216 + 40 * vx) / us / 1e6:.2f} TB/s", flush=True)
t0 = time.perf_counter()
for _ in range(20): r = n.pcl_align(w["guess"])
dt = (time.perf_counter() - t0) / 20
print(f"align (eps 0.01, step 0.1): {dt * 1e3:.3f} ms, iterations {r['iterations']}, evaluations {r['evaluations']}, score {r['score']:.3f}")
n2 = la.NDT(max_iterations=35, eps=1e-6); n2.set_target(w["map"]); n2.set_source(w["scan"])
t0 = time.perf_counter(); r2 = n2.pcl_align(w["guess"]); dt2 = time.perf_counter() - t0
print(f"align (eps 1e-6): {dt2 * 1e3:.3f} ms, iterations {r2['iterations']}, evaluations {r2['evaluations']}, score {r2['score']:.3f}, err {np.linalg.norm(r2['T'][:3,3]-w['T_true'][:3,3]) if 'T_true' in w else -1:.4f}")
if len(sys.argv) > 1:
    from oracle.oracle import Oracle
    o = Oracle(max_iterations=35, eps=0.01); o.set_target(w["map"]); o.set_source(w["scan"])
    t0 = time.perf_counter(); ro = o.pcl_eval(p); te = time.perf_counter() - t0
    print(f"oracle pass: {te * 1e3:.1f} ms; score rel diff {abs(ro['score'] - n.pcl_eval(p)['score']) / abs(ro['score']):.2e}")
