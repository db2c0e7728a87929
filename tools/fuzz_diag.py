"""Why a fuzz seed of tools/fuzz_more.py missed its bar: python tools/fuzz_diag.py scene|pcl <seed>.
Replays the test's own random draws, then prints what the engine and the oracle did side by side, the conditioning of
the normal equations at the end pose, and each side's evaluation at the OTHER side's end pose (which separates a
difference in the arithmetic of one pass from one that the iteration amplified)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as f
from conftest import pose_error, rel_err
import lio_slam_amd as la
from oracle.oracle import Oracle

kind, seed = sys.argv[1], int(sys.argv[2])
np.set_printoptions(precision=6, linewidth=200)


def scene(seed):
    rng = np.random.default_rng(1000 + seed)
    res = float(rng.choice([0.5, 0.8, 1.0, 1.5, 3.0]))
    cfg = dict(resolution=res, key_rule=int(rng.integers(0, 2)), nearby=int(rng.choice([1, 7])), min_pts_in_voxel=int(rng.integers(3, 9)),
               res_outlier_th=float(rng.choice([5.0, 20.0, 100.0])), eig_floor_ratio=float(rng.choice([1e-3, 1e-2, 5e-2])),
               max_iterations=int(rng.integers(3, 16)), eps=float(rng.choice([0.0, 1e-3, 1e-2])), min_effective_pts=int(rng.choice([10, 50])))
    pts, origin = f._scene(rng, int(rng.integers(4000, 30000)), extent=float(rng.uniform(10, 40)) * res)
    pts = rng.permutation(pts)
    bad = rng.choice(len(pts), 40, replace=False)
    pts[bad[:15], rng.integers(0, 3, 15)] = np.nan
    pts[bad[15:25], rng.integers(0, 3, 10)] = -np.inf
    pts[bad[25:35]] = pts[(bad[25:35] + 7) % len(pts)]
    pts[bad[35]] = [4e6 * res, 0, 0]
    pts[bad[36]] = np.float32(np.round(origin / res) * res)
    print("cfg", cfg, "points", len(pts))
    n = la.NDT(**cfg); o = Oracle(**cfg)
    cut = sorted(rng.choice(np.arange(1, len(pts)), 2, replace=False))
    n.set_target(pts[: cut[0]]); o.set_target(pts[: cut[0]])
    Tm = f._se3(rng, 0.4, 0.02)
    n.add_target(pts[cut[0]: cut[1]]); o.add_target(pts[cut[0]: cut[1]])
    n.add_target(pts[cut[1]:], Tm); o.add_target(pts[cut[1]:], Tm)
    T_true = f._se3(rng, 2.0, 0.3); T_true[:3, 3] += origin
    sel = rng.choice(len(pts), min(len(pts), int(rng.integers(500, 6000))), replace=False)
    world = pts[sel].astype(np.float64)
    with np.errstate(invalid="ignore"):
        src = ((world - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
    guess = T_true @ f._se3(rng, 0.15 * res, 0.01)
    half = len(src) // 2
    guesses = [guess, T_true, T_true @ f._se3(rng, 0.1 * res, 0.005)]
    scans = (src[:half], src[half:], src[:half])
    rb = n.align_batch([src[:half], src[half:]], guesses, [0, 1, 0])
    for p, (g, sc) in enumerate(zip(guesses, scans)):
        o.set_source(sc); n.set_source(sc)
        rp = o.align(g); rs = n.align(g, check=False)
        print(f"pair {p}: status batch {rb[p]['status']} single {rs['rc']} oracle {rp['rc']}; iterations {rb[p]['iterations']} {rs['iterations']} {rp['iterations']};"
              f" n_eff {rb[p]['n_effective']} {rs['n_effective']} {rp['n_effective']}")
        print("   batch vs oracle", pose_error(rb[p]["T"], rp["T"]), " single vs oracle", pose_error(rs["T"], rp["T"]), " batch vs single", pose_error(rb[p]["T"], rs["T"]))
        print("   last_dx", rb[p]["last_dx_norm"], rs["last_dx_norm"], rp.get("last_dx_norm"))
        w = np.linalg.eigvalsh(rp["H"]); print("   eig(H) oracle", w, "cond %.3g" % (w[-1] / max(w[0], 1e-300)))
        # one pass at the same pose on both sides: the per-pass arithmetic
        for nm, T in (("guess", g), ("oracle end", rp["T"]), ("engine end", rs["T"])):
            eg, eo = n.eval(T), o.eval(T)
            print(f"   pass at {nm}: n_eff {eg['n_effective']} {eo['n_effective']} relH {rel_err(eg['H'], eo['H']):.2e} relb {rel_err(eg['b'], eo['b']):.2e} |b| {np.linalg.norm(eo['b']):.3e}"
                  f" |H^-1 b| {np.linalg.norm(np.linalg.solve(eo['H'], eo['b'])):.3e}")
        # walk the oracle one iteration at a time beside the engine (max_iterations 1, restart from each side's pose)
        n1 = la.NDT(**{**cfg, "max_iterations": 1, "eps": 0.0}); o1 = Oracle(**{**cfg, "max_iterations": 1, "eps": 0.0})
        n1.load_map(n.save_map()); o1.set_target(pts[: cut[0]]); o1.add_target(pts[cut[0]: cut[1]]); o1.add_target(pts[cut[1]:], Tm)
        n1.set_source(sc); o1.set_source(sc)
        Tg = To = g
        for it in range(cfg["max_iterations"]):
            a = n1.align(Tg, check=False); b = o1.align(To)
            if a["rc"] or b["rc"]:
                print("   it", it, "rc", a["rc"], b["rc"]); break
            Tg, To = a["T"], b["T"]
            print(f"   it {it}: n_eff {a['n_effective']} {b['n_effective']} pose diff {pose_error(Tg, To)} dx {a['last_dx_norm']:.3e}")
        n1.close()
    n.close()


def pcl(seed):
    rng = np.random.default_rng(3000 + seed)
    res = float(rng.choice([0.5, 1.0, 2.0]))
    cfg = dict(resolution=res, max_iterations=int(rng.integers(2, 12)), eps=float(rng.choice([1e-2, 1e-4, 1e-6])))
    prm = dict(outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])), step_size=float(rng.choice([0.05, 0.1, 0.5])) * res,
               hessian_d1_as_pcl=int(rng.integers(0, 2)))
    pts, origin = f._scene(rng, int(rng.integers(6000, 20000)), extent=float(rng.uniform(10, 25)) * res)
    pts = rng.permutation(pts - origin.astype(np.float32))
    print("cfg", cfg, prm, "points", len(pts))
    n = la.NDT(**cfg); o = Oracle(**cfg)
    h = len(pts) // 2
    n.set_target(pts[:h]); o.set_target(pts[:h]); n.add_target(pts[h:]); o.add_target(pts[h:])
    T_true = f._se3(rng, 1.0, 0.2)
    sel = rng.choice(len(pts), int(rng.integers(800, 4000)), replace=False)
    src = ((pts[sel].astype(np.float64) - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
    src[3] = np.nan; src[11, 2] = np.inf
    guess = T_true @ f._se3(rng, 0.1 * res, 0.01)
    n.set_source(src); o.set_source(src)
    ra, rb = n.pcl_align(guess, **prm), o.pcl_align(guess, **prm)
    print("rc", ra["rc"], rb["rc"], "iterations", ra["iterations"], rb["iterations"], "evaluations", ra["evaluations"], rb["evaluations"],
          "converged", ra["converged"], rb["converged"])
    print("pose diff", pose_error(ra["T"], rb["T"]), "score", ra["score"], rb["score"], "neighbours", ra["n_neighbours"], rb["n_neighbours"])
    w = np.linalg.eigvalsh(rb["H"]); print("eig(H) oracle", w, "cond %.3g" % (abs(w).max() / max(abs(w).min(), 1e-300)))
    for nm, r in (("guess", None), ("oracle end", rb), ("engine end", ra)):
        p = la.pcl_vector_from_matrix(guess) if r is None else r["p"]
        eg, eo = n.pcl_eval(p, **prm), o.pcl_eval(p, **prm)
        print(f"pass at {nm}: neighbours {eg['n_neighbours']} {eo['n_neighbours']} score rel {abs(eg['score'] - eo['score']) / abs(eo['score']):.2e}"
              f" relg {rel_err(eg['g'], eo['g']):.2e} relH {rel_err(eg['H'], eo['H']):.2e} |g| {np.linalg.norm(eo['g']):.3e}")
    # the same walk one Newton iteration at a time, restarted from each side's own pose and from the oracle's pose
    n1 = la.NDT(**{**cfg, "max_iterations": 1}); o1 = Oracle(**{**cfg, "max_iterations": 1})
    n1.load_map(n.save_map()); o1.set_target(pts[:h]); o1.add_target(pts[h:])
    n1.set_source(src); o1.set_source(src)
    Tg = To = guess
    for it in range(cfg["max_iterations"]):
        a = n1.pcl_align(Tg, **prm); b = o1.pcl_align(To, **prm); c = n1.pcl_align(To, **prm)
        print(f"it {it}: evaluations {a['evaluations']} {b['evaluations']} (engine from the oracle's pose: {c['evaluations']}) own-walk diff {pose_error(a['T'], b['T'])}"
              f" same-start diff {pose_error(c['T'], b['T'])} neighbours {a['n_neighbours']} {b['n_neighbours']} {c['n_neighbours']}")
        Tg, To = a["T"], b["T"]
        if a["converged"] or b["converged"]:
            print("   converged", a["converged"], b["converged"]); break
    n1.close(); n.close()


{"scene": scene, "pcl": pcl}[kind](seed)
