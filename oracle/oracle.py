"""ctypes wrapper of oracle/libndt_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  See ndt_oracle.c for what the oracle restates and why parity with
the reference itself is unpinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libndt_oracle.so")


class Config(C.Structure):
    _fields_ = [
        ("resolution", C.c_float), ("max_iterations", C.c_int32), ("eps", C.c_double),
        ("nearby", C.c_int32), ("min_pts_in_voxel", C.c_int32), ("res_outlier_th", C.c_double),
        ("min_effective_pts", C.c_int32), ("key_rule", C.c_int32), ("eig_floor_ratio", C.c_double),
        ("final_eval", C.c_int32), ("device", C.c_int32),
        ("max_pts_in_voxel", C.c_int32), ("info_rule", C.c_int32), ("capacity", C.c_int32), ("reserved0", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16), ("H", C.c_double * 36), ("b", C.c_double * 6), ("chi2", C.c_double),
        ("last_dx_norm", C.c_double), ("n_effective", C.c_int64), ("iterations", C.c_int32),
        ("converged", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32),
    ]


class PclParams(C.Structure):
    _fields_ = [("outlier_ratio", C.c_double), ("step_size", C.c_double), ("rotation_eps", C.c_double),
                ("hessian_d1_as_pcl", C.c_int32), ("reserved", C.c_int32)]


class PclResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("p", C.c_double * 6), ("score", C.c_double), ("trans_probability", C.c_double),
                ("gradient", C.c_double * 6), ("hessian", C.c_double * 36), ("n_neighbours", C.c_int64),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("status", C.c_int32), ("evaluations", C.c_int32)]


def pcl_params(**kw):
    prm = PclParams(0.55, 0.1, 0.0, 1, 0)
    for k, v in kw.items():
        if not hasattr(prm, k):
            raise AttributeError(k)
        setattr(prm, k, v)
    return prm


def pcl_result_dict(r):
    return dict(rc=r.status, status=r.status, T=np.array(r.T).reshape(4, 4).T.copy(), p=np.array(r.p), score=r.score,
                trans_probability=r.trans_probability, g=np.array(r.gradient), H=np.array(r.hessian).reshape(6, 6),
                n_neighbours=r.n_neighbours, iterations=r.iterations, converged=bool(r.converged), evaluations=r.evaluations)


def default_config(**kw):
    cfg = Config(1.0, 35, 1e-2, 7, 6, 20.0, 10, 0, 0.01, 1, 0, 0, 0, 0, 0)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    vp, sz, dp = C.c_void_p, C.c_size_t, C.POINTER(C.c_double)
    lib.ora_create.restype = vp
    lib.ora_create.argtypes = [C.POINTER(Config), C.c_int]
    lib.ora_destroy.argtypes = [vp]
    lib.ora_set_threads.argtypes = [vp, C.c_int]
    lib.ora_set_params.argtypes = [vp, C.POINTER(Config)]
    lib.ora_set_target.argtypes = [vp, vp, sz, sz]
    lib.ora_add_target.argtypes = [vp, vp, sz, sz, vp]
    lib.ora_set_source.argtypes = [vp, vp, sz, sz]
    lib.ora_eval.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.ora_source_cells.argtypes = [vp, vp, vp, vp]
    lib.ora_align.argtypes = [vp, vp, C.POINTER(Result)]
    lib.ora_export_voxels.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, C.POINTER(sz)]
    lib.ora_source_size.restype = C.c_int64
    lib.ora_source_size.argtypes = [vp]
    lib.ora_num_voxels.restype = C.c_int64
    lib.ora_num_voxels.argtypes = [vp]
    lib.ora_target_key.argtypes = [vp, vp, vp]
    lib.ora_source_key.argtypes = [vp, vp, vp]
    lib.ora_voxel_downsample.argtypes = [vp, sz, sz, C.c_float, vp, vp, sz, C.POINTER(sz)]
    lib.ora_pcl_eval.argtypes = [vp, C.POINTER(PclParams), vp, vp, vp, vp, vp]
    lib.ora_pcl_align.argtypes = [vp, C.POINTER(PclParams), vp, C.POINTER(PclResult)]
    lib.ora_pcl_vector_from_matrix.argtypes = [vp, vp]
    lib.ora_pcl_matrix_from_vector.argtypes = [vp, vp]
    lib.ora_pack_key.restype = C.c_int64
    lib.ora_pack_key.argtypes = [C.c_int32] * 3
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a, a.shape[0], a.shape[1] * 4


def _T(T):
    """4x4 (row-indexed numpy) -> column-major double[16]."""
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).reshape(16)


class Oracle:
    """Same call surface as lio_slam_amd.NDT, computed on the CPU in float64."""

    def __init__(self, quantize=True, threads=1, **cfg):
        self.cfg = default_config(**cfg)
        self._h = lib().ora_create(C.byref(self.cfg), int(quantize))
        lib().ora_set_threads(self._h, threads)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ora_destroy(self._h)
            self._h = None

    def set_params(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        lib().ora_set_params(self._h, C.byref(self.cfg))

    def set_threads(self, n):
        lib().ora_set_threads(self._h, n)

    def set_target(self, pts):
        a, n, s = _pts(pts)
        return lib().ora_set_target(self._h, a.ctypes.data, n, s)

    def add_target(self, pts, T=None):
        a, n, s = _pts(pts)
        t = None if T is None else _T(T)
        return lib().ora_add_target(self._h, a.ctypes.data, n, s, None if t is None else t.ctypes.data)

    def set_source(self, pts):
        a, n, s = _pts(pts)
        return lib().ora_set_source(self._h, a.ctypes.data, n, s)

    def source_size(self):
        return lib().ora_source_size(self._h)

    def num_voxels(self):
        return lib().ora_num_voxels(self._h)

    def eval(self, T):
        t = _T(T)
        H = np.zeros((6, 6)); b = np.zeros(6)
        chi2 = C.c_double(); ne = C.c_int64(); hits = C.c_double()
        rc = lib().ora_eval(self._h, t.ctypes.data, H.ctypes.data, b.ctypes.data, C.addressof(chi2),
                            C.addressof(ne), C.addressof(hits))
        if rc:
            raise RuntimeError(f"ora_eval rc={rc}")
        return dict(H=H, b=b, chi2=chi2.value, n_effective=ne.value, hits=hits.value)

    def source_cells(self, T):
        t = _T(T)
        n = self.source_size()
        keys = np.zeros(n, np.int64); mask = np.zeros(n, np.uint8)
        rc = lib().ora_source_cells(self._h, t.ctypes.data, keys.ctypes.data, mask.ctypes.data)
        if rc:
            raise RuntimeError(f"ora_source_cells rc={rc}")
        return keys, mask

    def align(self, guess):
        t = _T(guess)
        r = Result()
        rc = lib().ora_align(self._h, t.ctypes.data, C.byref(r))
        return dict(rc=rc, T=np.array(r.T).reshape(4, 4).T.copy(), H=np.array(r.H).reshape(6, 6),
                    b=np.array(r.b), chi2=r.chi2, n_effective=r.n_effective, iterations=r.iterations,
                    converged=bool(r.converged), last_dx_norm=r.last_dx_norm)

    def export_voxels(self):
        n = self.num_voxels()
        keys = np.zeros(n, np.int64); cnt = np.zeros(n, np.int32)
        mean = np.zeros((n, 3)); cov = np.zeros((n, 3, 3)); info = np.zeros((n, 3, 3))
        valid = np.zeros(n, np.uint8)
        no = C.c_size_t()
        rc = lib().ora_export_voxels(self._h, keys.ctypes.data, cnt.ctypes.data, mean.ctypes.data,
                                     cov.ctypes.data, info.ctypes.data, valid.ctypes.data, n, C.byref(no))
        if rc:
            raise RuntimeError(f"ora_export_voxels rc={rc}")
        return dict(keys=keys, counts=cnt, mean=mean, cov=cov, info=info, valid=valid)

    # --- PCL compatibility mode
    def pcl_eval(self, p, hessian=True, **prm):
        p = np.ascontiguousarray(p, np.float64); g = np.zeros(6); H = np.zeros((6, 6))
        sc = C.c_double(); nb = C.c_int64(); pr = pcl_params(**prm)
        rc = lib().ora_pcl_eval(self._h, C.byref(pr), p.ctypes.data, C.addressof(sc), g.ctypes.data,
                                H.ctypes.data if hessian else None, C.addressof(nb))
        if rc:
            raise RuntimeError(f"ora_pcl_eval rc={rc}")
        return dict(score=sc.value, g=g, H=H if hessian else None, n_neighbours=nb.value)

    def pcl_align(self, guess, **prm):
        t = _T(guess); r = PclResult(); pr = pcl_params(**prm)
        lib().ora_pcl_align(self._h, C.byref(pr), t.ctypes.data, C.byref(r))
        return pcl_result_dict(r)

    def target_key(self, p):
        p = np.asarray(p, np.float32); c = np.zeros(3, np.int32)
        ok = lib().ora_target_key(self._h, p.ctypes.data, c.ctypes.data)
        return c if ok else None

    def source_key(self, q):
        q = np.asarray(q, np.float64); c = np.zeros(3, np.int32)
        ok = lib().ora_source_key(self._h, q.ctypes.data, c.ctypes.data)
        return c if ok else None


def voxel_downsample(pts, leaf=0.5, T=None):
    """pcl::transformPointCloud + pcl::VoxelGrid(leaf) (process.cc:62-73), xyz only, on the CPU."""
    a, n, s = _pts(pts)
    t = None if T is None else _T(T)
    out = np.zeros((max(n, 1), 3), np.float32); no = C.c_size_t()
    rc = lib().ora_voxel_downsample(a.ctypes.data, n, s, float(leaf), None if t is None else t.ctypes.data,
                                    out.ctypes.data, n, C.byref(no))
    if rc:
        raise RuntimeError(f"ora_voxel_downsample rc={rc}")
    return out[:no.value].copy()


def pcl_vector_from_matrix(T):
    t = _T(T); p = np.zeros(6)
    lib().ora_pcl_vector_from_matrix(t.ctypes.data, p.ctypes.data)
    return p


def pcl_matrix_from_vector(p):
    p = np.ascontiguousarray(p, np.float64); t = np.zeros(16)
    lib().ora_pcl_matrix_from_vector(p.ctypes.data, t.ctypes.data)
    return t.reshape(4, 4).T.copy()
