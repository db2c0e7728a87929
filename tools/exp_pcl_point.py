"""Where a PCL-mode pass of the engine and of the oracle differ: the pass point by point (one-point sources) on a fuzz scene.
python tools/exp_pcl_point.py <seed> [n_points]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as f
from conftest import rel_err
import lio_slam_amd as la
if os.environ.get("PCL_DBG_LIB"):
    la.LIB_PATH = os.path.abspath(os.environ["PCL_DBG_LIB"])
from oracle.oracle import Oracle

seed = int(sys.argv[1]); npts = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(3000 + seed)
res = float(rng.choice([0.5, 1.0, 2.0]))
cfg = dict(resolution=res, max_iterations=int(rng.integers(2, 12)), eps=float(rng.choice([1e-2, 1e-4, 1e-6])))
prm = dict(outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])), step_size=float(rng.choice([0.05, 0.1, 0.5])) * res, hessian_d1_as_pcl=int(rng.integers(0, 2)))
pts, origin = f._scene(rng, int(rng.integers(6000, 20000)), extent=float(rng.uniform(10, 25)) * res)
pts = rng.permutation(pts - origin.astype(np.float32))
n = la.NDT(**cfg); o = Oracle(**cfg); o0 = Oracle(quantize=0, **cfg)
h = len(pts) // 2
for x in (n, o, o0):
    x.set_target(pts[:h]); x.add_target(pts[h:])
T_true = f._se3(rng, 1.0, 0.2)
sel = rng.choice(len(pts), int(rng.integers(800, 4000)), replace=False)
src = ((pts[sel].astype(np.float64) - T_true[:3, 3]) @ T_true[:3, :3]).astype(np.float32)
guess = T_true @ f._se3(rng, 0.1 * res, 0.01)
p = la.pcl_vector_from_matrix(guess)
vg, vo = n.export_voxels(), o.export_voxels()
assert np.array_equal(vg["keys"], vo["keys"])
di = np.abs(vg["info"] - vo["info"]); sc = np.abs(vo["info"]).max(axis=(1, 2), keepdims=True) + 1e-300
print("voxels", len(vo["keys"]), "valid", int(vo["valid"].sum()), "info entries that differ", int((di > 0).sum()), "max rel", float((di / sc).max()))
print("mean max abs diff", float(np.abs(vg["mean"] - vo["mean"]).max()), "cov max rel", float((np.abs(vg["cov"] - vo["cov"]) / (np.abs(vo["cov"]).max(axis=(1, 2), keepdims=True) + 1e-300)).max()))
worst = []
for x in (n, o, o0): x.set_source(src)
eg, eo, e0 = n.pcl_eval(p, **prm), o.pcl_eval(p, **prm), o0.pcl_eval(p, **prm)
print("whole source: engine vs oracle relH %.2e relg %.2e score %.2e | oracle q1 vs q0 relH %.2e relg %.2e" % (
    rel_err(eg["H"], eo["H"]), rel_err(eg["g"], eo["g"]), abs(eg["score"] - eo["score"]) / abs(eo["score"]), rel_err(eo["H"], e0["H"]), rel_err(eo["g"], e0["g"])))
Hs_g = np.zeros((6, 6)); Hs_o = np.zeros((6, 6))
for i in range(min(npts, len(src))):
    n.set_source(src[i:i + 1]); o.set_source(src[i:i + 1])
    a, b = n.pcl_eval(p, **prm), o.pcl_eval(p, **prm)
    if b["n_neighbours"] == 0 and a["n_neighbours"] == 0:
        continue
    Hs_g += a["H"]; Hs_o += b["H"]
    worst.append((rel_err(a["H"], b["H"]), i, a["n_neighbours"], b["n_neighbours"], np.abs(b["H"]).max()))
worst.sort(reverse=True)
print("per point, worst relH:", [(f"{w[0]:.2e}", w[1], w[2], w[3], f"{w[4]:.2e}") for w in worst[:8]])
print("median relH per point %.2e; sum over these points relH %.2e" % (np.median([w[0] for w in worst]), rel_err(Hs_g, Hs_o)))
i = worst[0][1]
n.set_source(src[i:i + 1]); o.set_source(src[i:i + 1])
a, b = n.pcl_eval(p, **prm), o.pcl_eval(p, **prm)
np.set_printoptions(precision=10, linewidth=220)
print("point", i, src[i], "score", a["score"], b["score"]); print(a["H"] - b["H"]); print(b["H"])

# the same point in NumPy: float32 transform in the oracle's order, float64 everything else, from the exported table
import math
def rt(p):
    cx, sx, cy, sy, cz, sz = math.cos(p[3]), math.sin(p[3]), math.cos(p[4]), math.sin(p[4]), math.cos(p[5]), math.sin(p[5])
    R = np.array([[cy * cz, -(cy * sz), sy], [cx * sz + (sx * sy) * cz, cx * cz - (sx * sy) * sz, -(sx * cy)], [sx * sz - (cx * sy) * cz, sx * cz + (cx * sy) * sz, cx * cy]])
    return R, np.array(p[:3])
R, t = rt(p)
M = np.concatenate([R, t[:, None]], axis=1).astype(np.float32)
x = src[i]
q = np.array([np.float32(np.float32(np.float32(M[r, 0] * x[0]) + np.float32(M[r, 1] * x[1])) + np.float32(M[r, 2] * x[2])) + M[r, 3] for r in range(3)], np.float32)
print("q float32", q, [float(v).hex() for v in q])
c1 = 10.0 * (1.0 - prm["outlier_ratio"]); c2 = prm["outlier_ratio"] / res ** 3; d3 = -math.log(c2)
d1 = -math.log(c1 + c2) - d3; d2 = -2.0 * math.log((-math.log(c1 * math.exp(-0.5) + c2) - d3) / d1)
keys = vo["keys"]; val = vo["valid"].astype(bool)
cen = vo["mean"][val].astype(np.float32); info = vo["info"][val]; mean = vo["mean"][val]
dist2 = ((q[None, :] - cen) ** 2).astype(np.float32)
dist2 = (dist2[:, 0] + dist2[:, 1]) + dist2[:, 2]
nb = np.nonzero(dist2 < np.float32(res * res))[0]
sc = 0.0
for k in nb:
    xt = q.astype(np.float64) - mean[k]
    qv = xt @ info[k] @ xt
    sc += -d1 * math.exp(-d2 * qv / 2.0)
    print("  neighbour", k, "dist2", dist2[k], "qv", qv, "term", -d1 * math.exp(-d2 * qv / 2.0))
print("numpy score", sc, "engine", a["score"], "oracle", b["score"])

def score_of(qv32):
    qq = np.asarray(qv32, np.float32)
    d2v = ((qq[None, :] - cen) ** 2).astype(np.float32); d2v = (d2v[:, 0] + d2v[:, 1]) + d2v[:, 2]
    tot = 0.0
    for k in np.nonzero(d2v < np.float32(res * res))[0]:
        xt = qq.astype(np.float64) - mean[k]
        tot += -d1 * math.exp(-d2 * (xt @ info[k] @ xt) / 2.0)
    return tot
M64 = M.astype(np.float64); x64 = x.astype(np.float64)
variants = {
  "double then one rounding": (M64[:, :3] @ x64 + M64[:, 3]).astype(np.float32),
  "fma chain": np.array([np.float32(np.float32(np.float32(M64[r, 2] * x64[2] + np.float64(np.float32(M64[r, 1] * x64[1] + np.float64(np.float32(M64[r, 0] * x64[0]))))) ) + M[r, 3]) for r in range(3)], np.float32),
  "R,t in float64 (no float32 matrix)": (R @ x64 + t).astype(np.float32),
}
for nm, qq in variants.items():
    print(f"{nm:40s} q {[float(v).hex() for v in qq]} score {score_of(qq)!r}")
for dx in (-1, 1):
    for ax in range(3):
        qq = q.copy(); qq[ax] = np.nextafter(qq[ax], np.float32(dx * 1e9))
        print("q with axis", ax, "moved", dx, "ulp: score", repr(score_of(qq)))

if os.environ.get("PCL_DBG_LIB"):
    sys.stdout.flush()
    n.set_source(src[i:i + 1]); r = n.pcl_eval(p, **prm); n.synchronize() if hasattr(n, "synchronize") else None
    print("engine (debug build) score", repr(r["score"]), "d1 d2 numpy", repr(d1), repr(d2))
    for k in nb:
        xt = q.astype(np.float64) - mean[k]
        print("numpy neighbour", k, "key", keys[val][k], "x'", [repr(float(v)) for v in xt], "off", [float(np.float32(mean[k][j] - round((mean[k][j] - 0) // res) * res)).hex() for j in range(3)])
