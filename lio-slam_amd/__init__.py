"""lio_slam_amd — MI355X-native NDT scan matching for lio-slam's point-cloud-matching step.

Python is only a thin ctypes view of the C ABI in include/ndt_mi355.h (the product is
libndt_mi355.so, built from lio-slam_amd/csrc/*.hip for gfx950).  The class below mirrors
the call surface the reference uses on its `pcl::NormalDistributionsTransform ndt_` member
(src/slam/include/process.h:86; src/slam/src/process.cc:78,87,94,95,147,373-376).
There is no CPU fallback: if the HIP library is missing or no GPU is present, construction
raises.
"""
import collections.abc
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libndt_mi355.so")

NDT_KEY_FLOOR_F32, NDT_KEY_ROUND_F64 = 0, 1
NDT_INFO_EIG_CLAMP, NDT_INFO_ADD_1E3 = 0, 1
ERRORS = {0: "OK", -1: "INVALID_ARG", -2: "NO_TARGET", -3: "NO_SOURCE", -4: "HIP", -5: "TOO_FEW_POINTS",
          -6: "SINGULAR", -7: "KEY_RANGE", -8: "COMM", -9: "CAPACITY"}

ABI_SYMBOLS = [
    "ndt_default_config", "ndt_create", "ndt_destroy", "ndt_configure", "ndt_set_target", "ndt_add_target",
    "ndt_set_source", "ndt_set_source_downsampled", "ndt_get_source", "ndt_add_target_from_source", "ndt_align", "ndt_align_static", "ndt_eval", "ndt_source_cells", "ndt_source_size", "ndt_num_voxels",
    "ndt_export_voxels", "ndt_map_save", "ndt_map_load", "ndt_pcl_default_params", "ndt_pcl_eval", "ndt_pcl_align", "ndt_pcl_time_pass", "ndt_pcl_vector_from_matrix", "ndt_pcl_matrix_from_vector", "ndt_information_from_hessian", "ndt_pack_key", "ndt_unpack_key", "ndt_align_batch", "ndt_set_batch_sources", "ndt_align_batch_resident", "ndt_comm_unique_id",
    "ndt_comm_init", "ndt_comm_destroy", "ndt_comm_size", "ndt_comm_p2p_export", "ndt_comm_p2p_init", "ndt_set_source_async", "ndt_time_chain", "ndt_profile", "ndt_profile_read", "ndt_time_residual", "ndt_time_residual_batch", "ndt_synchronize",
    "ndt_last_error", "ndt_version", "ndt_last_radix_passes",
    "ndt_create_group", "ndt_destroy_group", "ndt_group_size", "ndt_group_handle", "ndt_group_set_target", "ndt_group_add_target",
    "ndt_group_align_batch", "ndt_group_last_error",
]


class NdtError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__(f"{ERRORS.get(rc, rc)}: {msg}")
        self.rc = rc


class Config(C.Structure):
    _fields_ = [
        ("resolution", C.c_float), ("max_iterations", C.c_int32), ("eps", C.c_double),
        ("nearby", C.c_int32), ("min_pts_in_voxel", C.c_int32), ("res_outlier_th", C.c_double),
        ("min_effective_pts", C.c_int32), ("key_rule", C.c_int32), ("eig_floor_ratio", C.c_double),
        ("final_eval", C.c_int32), ("device", C.c_int32),
        ("max_pts_in_voxel", C.c_int32), ("info_rule", C.c_int32), ("capacity", C.c_int32), ("reserved0", C.c_int32),
    ]


class PclParams(C.Structure):
    _fields_ = [("outlier_ratio", C.c_double), ("step_size", C.c_double), ("rotation_eps", C.c_double),
                ("hessian_d1_as_pcl", C.c_int32), ("reserved", C.c_int32)]


class PclResult(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("p", C.c_double * 6), ("score", C.c_double), ("trans_probability", C.c_double),
                ("gradient", C.c_double * 6), ("hessian", C.c_double * 36), ("n_neighbours", C.c_int64),
                ("iterations", C.c_int32), ("converged", C.c_int32), ("status", C.c_int32), ("evaluations", C.c_int32)]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16), ("H", C.c_double * 36), ("b", C.c_double * 6), ("chi2", C.c_double),
        ("last_dx_norm", C.c_double), ("n_effective", C.c_int64), ("iterations", C.c_int32),
        ("converged", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32),
    ]


def build(force=False):
    """Compile libndt_mi355.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")])


_lib = None


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the NDT engine has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.ndt_default_config.argtypes = [C.POINTER(Config)]
    L.ndt_create.restype = vp
    L.ndt_create.argtypes = [C.POINTER(Config)]
    L.ndt_destroy.argtypes = [vp]
    L.ndt_configure.argtypes = [vp, C.POINTER(Config)]
    L.ndt_set_target.argtypes = [vp, vp, sz, sz]
    L.ndt_add_target.argtypes = [vp, vp, sz, sz, vp]
    L.ndt_set_source.argtypes = [vp, vp, sz, sz]
    L.ndt_set_source_async.argtypes = [vp, vp, sz, sz]
    L.ndt_set_source_downsampled.argtypes = [vp, vp, sz, sz, C.c_float, vp]
    L.ndt_get_source.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.ndt_add_target_from_source.argtypes = [vp, vp]
    L.ndt_align.argtypes = [vp, vp, C.POINTER(Result)]
    L.ndt_align_static.argtypes = [vp, vp, vp, vp, sz, vp, C.POINTER(Result)]
    L.ndt_eval.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ndt_source_cells.argtypes = [vp, vp, vp, vp, sz]
    L.ndt_source_size.restype = C.c_int64
    L.ndt_source_size.argtypes = [vp]
    L.ndt_num_voxels.restype = C.c_int64
    L.ndt_num_voxels.argtypes = [vp]
    L.ndt_export_voxels.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, C.POINTER(sz)]
    L.ndt_pcl_default_params.argtypes = [C.POINTER(PclParams)]
    L.ndt_pcl_eval.argtypes = [vp, C.POINTER(PclParams), vp, vp, vp, vp, vp]
    L.ndt_pcl_align.argtypes = [vp, C.POINTER(PclParams), vp, C.POINTER(PclResult)]
    L.ndt_pcl_time_pass.argtypes = [vp, C.POINTER(PclParams), vp, C.c_int, C.c_int, vp, vp, vp]
    L.ndt_pcl_vector_from_matrix.argtypes = [vp, vp]
    L.ndt_pcl_matrix_from_vector.argtypes = [vp, vp]
    L.ndt_map_save.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.ndt_map_load.argtypes = [vp, vp, sz]
    L.ndt_information_from_hessian.argtypes = [vp, C.c_double, C.c_double, vp]
    L.ndt_pack_key.restype = C.c_int64
    L.ndt_pack_key.argtypes = [C.c_int32] * 3
    L.ndt_unpack_key.argtypes = [C.c_int64, vp, vp, vp]
    L.ndt_align_batch.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    L.ndt_set_batch_sources.argtypes = [vp, vp, sz, vp, sz]
    L.ndt_align_batch_resident.argtypes = [vp, vp, sz, vp, vp]
    L.ndt_last_radix_passes.argtypes = [vp]
    L.ndt_create_group.restype = vp
    L.ndt_create_group.argtypes = [vp, C.c_int, vp]
    L.ndt_destroy_group.argtypes = [vp]
    L.ndt_destroy_group.restype = None
    L.ndt_group_size.argtypes = [vp]
    L.ndt_group_handle.restype = vp
    L.ndt_group_handle.argtypes = [vp, C.c_int]
    L.ndt_group_set_target.argtypes = [vp, vp, sz, sz]
    L.ndt_group_add_target.argtypes = [vp, vp, sz, sz, vp]
    L.ndt_group_align_batch.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    L.ndt_group_last_error.restype = C.c_char_p
    L.ndt_group_last_error.argtypes = [vp]
    L.ndt_comm_unique_id.argtypes = [vp]
    L.ndt_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.ndt_comm_destroy.argtypes = [vp]
    L.ndt_comm_size.argtypes = [vp]
    L.ndt_comm_p2p_export.argtypes = [vp, vp]
    L.ndt_comm_p2p_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.ndt_time_chain.argtypes = [vp, vp, C.c_int, vp]
    L.ndt_profile.argtypes = [vp, C.c_int]
    L.ndt_profile_read.argtypes = [vp, vp, vp, vp]
    L.ndt_time_residual.argtypes = [vp, vp, C.c_int, vp]
    L.ndt_time_residual_batch.argtypes = [vp, C.c_int, vp, vp, vp]
    L.ndt_synchronize.argtypes = [vp]
    L.ndt_last_error.restype = C.c_char_p
    L.ndt_last_error.argtypes = [vp]
    L.ndt_version.restype = C.c_char_p
    _lib = L
    return L


RESULT_DTYPE = np.dtype([("T", "f8", 16), ("H", "f8", 36), ("b", "f8", 6), ("chi2", "f8"), ("last_dx_norm", "f8"),
                         ("n_effective", "i8"), ("iterations", "i4"), ("converged", "i4"), ("status", "i4"), ("reserved", "i4")])
assert RESULT_DTYPE.itemsize == C.sizeof(Result)


def default_config(**kw):
    cfg = Config()
    lib().ndt_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def _T(T):
    """4x4 numpy pose -> column-major double[16] (Eigen's layout)."""
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("pose must be 4x4")
    return np.ascontiguousarray(T.T).reshape(16)


def _points(pts):
    """(pointer, n, stride) of a point buffer: a float32 numpy array (n, >=3), or (device_ptr, n, stride)."""
    if isinstance(pts, tuple):
        ptr, n, stride = pts
        return None, int(ptr), int(n), int(stride)
    a = np.ascontiguousarray(pts, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("points must be (n, >=3) float32")
    return a, a.ctypes.data, a.shape[0], a.shape[1] * 4


class BatchResults(collections.abc.Sequence):
    """Results of a batched align: behaves like a list of result dicts (same keys as align(), plus 'status'), built per
    item on access — building a few thousand dicts per call would cost a replay loop in Python more time than the GPU
    spends aligning.  `.array` is the structured ndt_result array itself."""

    def __init__(self, res):
        self.array = res
        n = len(res)
        self._T = res["T"].reshape(n, 4, 4).transpose(0, 2, 1)
        self._H = res["H"].reshape(n, 6, 6)

    def __len__(self):
        return len(self.array)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        r = self.array[i]
        st = int(r["status"])
        return dict(rc=st, status=st, T=self._T[i], H=self._H[i], b=r["b"], chi2=float(r["chi2"]), n_effective=int(r["n_effective"]),
                    iterations=int(r["iterations"]), converged=bool(r["converged"]), last_dx_norm=float(r["last_dx_norm"]))


def _unpack_results(res):
    return BatchResults(res)


class NDTGroup:
    """Several GPUs in one process (ndt_create_group): one member engine per entry of `devices`, map replicas, a replay
    batch dealt to the members in contiguous shares.  member(i) is a borrowed NDT for per-device calls."""

    def __init__(self, devices, **cfg):
        self.cfg = default_config(**cfg)
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._g = lib().ndt_create_group(C.byref(self.cfg), len(devices), ids)
        if not self._g:
            raise NdtError(-4, "ndt_create_group failed (device not visible, or bad config)")

    def close(self):
        if getattr(self, "_g", None):
            lib().ndt_destroy_group(self._g)
            self._g = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise NdtError(rc, lib().ndt_group_last_error(self._g).decode())

    def __len__(self): return lib().ndt_group_size(self._g)

    def member(self, i):
        h = lib().ndt_group_handle(self._g, int(i))
        if not h:
            raise IndexError(i)
        n = NDT.__new__(NDT)
        n.cfg, n._h, n.final, n._borrowed = self.cfg, h, None, True
        return n

    def set_target(self, pts):
        keep, p, n, s = _points(pts)
        self._chk(lib().ndt_group_set_target(self._g, p, n, s))

    def add_target(self, pts, T=None):
        keep, p, n, s = _points(pts)
        t = None if T is None else _T(T)
        self._chk(lib().ndt_group_add_target(self._g, p, n, s, None if t is None else t.ctypes.data))

    def align_batch(self, scans, guesses, pair_scan=None):
        scans = [np.ascontiguousarray(np.asarray(a, dtype=np.float32)[:, :3]) for a in scans]
        pts = np.ascontiguousarray(np.concatenate(scans))
        off = np.zeros(len(scans) + 1, np.int64); off[1:] = np.cumsum([len(a) for a in scans])
        G = np.asarray(guesses, dtype=np.float64)
        g = np.ascontiguousarray(G.transpose(0, 2, 1)).reshape(len(G), 16)
        ps = None if pair_scan is None else np.ascontiguousarray(pair_scan, dtype=np.int32)
        res = np.zeros(len(g), RESULT_DTYPE)
        self._chk(lib().ndt_group_align_batch(self._g, pts.ctypes.data, 12, off.ctypes.data, len(scans), None if ps is None else ps.ctypes.data,
                                              len(g), g.ctypes.data, res.ctypes.data))
        return _unpack_results(res)


class NDT:
    """Mirror of the reference's NDT member: set target, set source, align from a guess, read pose + Hessian.

    Reference setter -> here: setResolution -> resolution, setMaximumIterations -> max_iterations,
    setTransformationEpsilon -> eps, setStepSize -> accepted and ignored (Gauss-Newton has no line search).
    """

    def __init__(self, **cfg):
        self.cfg = default_config(**cfg)
        self._h = lib().ndt_create(C.byref(self.cfg))
        if not self._h:
            raise NdtError(-4, "ndt_create failed (no MI355X visible, or bad config)")
        self.final = None

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):       # a group's member belongs to the group
                lib().ndt_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise NdtError(rc, lib().ndt_last_error(self._h).decode())

    # --- the reference's setters (process.cc:373-376)
    def configure(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            setattr(self.cfg, k, v)
        self._chk(lib().ndt_configure(self._h, C.byref(self.cfg)))

    def setResolution(self, r): self.configure(resolution=float(r))
    def setMaximumIterations(self, n): self.configure(max_iterations=int(n))
    def setTransformationEpsilon(self, e): self.configure(eps=float(e))
    def setStepSize(self, s): pass

    # --- clouds
    def set_target(self, pts):
        keep, p, n, s = _points(pts)
        self._chk(lib().ndt_set_target(self._h, p, n, s))

    def add_target(self, pts, T=None):
        keep, p, n, s = _points(pts)
        t = None if T is None else _T(T)
        self._chk(lib().ndt_add_target(self._h, p, n, s, None if t is None else t.ctypes.data))

    def set_source(self, pts):
        keep, p, n, s = _points(pts)
        self._chk(lib().ndt_set_source(self._h, p, n, s))

    def set_source_async(self, dev_view):
        """dev_view = (device pointer, n, stride): records in this device's memory, read on the engine's stream; they must stay
        valid until a later call that returns results has returned."""
        keep, p, n, s = _points(dev_view)
        self._chk(lib().ndt_set_source_async(self._h, p, n, s))

    def set_source_downsampled(self, pts, leaf=0.5, T=None):
        """process.cc:62-73 + 87 on the device: extrinsic transform, VoxelGrid(leaf) centroids, set as the source."""
        keep, p, n, s = _points(pts)
        t = None if T is None else _T(T)
        self._chk(lib().ndt_set_source_downsampled(self._h, p, n, s, float(leaf), None if t is None else t.ctypes.data))

    def get_source(self):
        n = self.source_size()
        out = np.zeros((n, 3), np.float32); no = C.c_size_t()
        self._chk(lib().ndt_get_source(self._h, out.ctypes.data, n, C.byref(no)))
        return out

    def add_target_from_source(self, T=None):
        t = None if T is None else _T(T)
        self._chk(lib().ndt_add_target_from_source(self._h, None if t is None else t.ctypes.data))

    setInputTarget, setInputSource = set_target, set_source

    def last_radix_passes(self): return lib().ndt_last_radix_passes(self._h)
    def source_size(self): return lib().ndt_source_size(self._h)
    def num_voxels(self): return lib().ndt_num_voxels(self._h)

    # --- the hot path
    def eval(self, T):
        t = _T(T)
        H = np.zeros((6, 6)); b = np.zeros(6)
        chi2 = C.c_double(); ne = C.c_int64()
        self._chk(lib().ndt_eval(self._h, t.ctypes.data, H.ctypes.data, b.ctypes.data, C.addressof(chi2), C.addressof(ne)))
        return dict(H=H, b=b, chi2=chi2.value, n_effective=ne.value)

    def align(self, guess, check=True):
        t = _T(guess)
        r = Result()
        rc = lib().ndt_align(self._h, t.ctypes.data, C.byref(r))
        if check:
            self._chk(rc)
        self.final = dict(rc=rc, T=np.array(r.T).reshape(4, 4).T.copy(), H=np.array(r.H).reshape(6, 6), b=np.array(r.b),
                          chi2=r.chi2, n_effective=r.n_effective, iterations=r.iterations, converged=bool(r.converged),
                          last_dx_norm=r.last_dx_norm)
        return self.final

    def align_static(self, d_x, d_y, d_z, n, guess, check=True):
        """ndt_align_static: the entry point that allocates nothing.  d_x / d_y / d_z are device addresses (ints) of the
        caller's float32 arrays of n source coordinates each; same result dict as align()."""
        t = _T(guess)
        r = Result()
        rc = lib().ndt_align_static(self._h, int(d_x), int(d_y), int(d_z), int(n), t.ctypes.data, C.byref(r))
        if check:
            self._chk(rc)
        return dict(rc=rc, T=np.array(r.T).reshape(4, 4).T.copy(), H=np.array(r.H).reshape(6, 6), b=np.array(r.b), chi2=r.chi2,
                    n_effective=r.n_effective, iterations=r.iterations, converged=bool(r.converged), last_dx_norm=r.last_dx_norm)

    def align_batch(self, scans, guesses, pair_scan=None):
        """Offline replay: align many (scan, guess) pairs against the resident map in one batched Gauss-Newton loop.

        scans: list of float32 (n_i, 3) arrays; guesses: (n_pairs, 4, 4); pair_scan: scan index per pair (None: pair p
        uses scan p).  Returns a list of result dicts (same keys as align(), plus 'status')."""
        self.set_batch_sources(scans)
        return self.align_batch_resident(guesses, pair_scan)

    def set_batch_sources(self, scans):
        """Upload the scans of an offline-replay batch once; they stay resident for align_batch_resident()."""
        scans = [np.ascontiguousarray(np.asarray(a, dtype=np.float32)[:, :3]) for a in scans]
        pts = np.ascontiguousarray(np.concatenate(scans)) if scans else np.zeros((0, 3), np.float32)
        off = np.zeros(len(scans) + 1, np.int64); off[1:] = np.cumsum([len(a) for a in scans])
        self._chk(lib().ndt_set_batch_sources(self._h, pts.ctypes.data, 12, off.ctypes.data, len(scans)))

    def align_batch_resident(self, guesses, pair_scan=None):
        """guesses: (n, 4, 4) array or a list of 4x4 poses.  Results are unpacked in one vectorised step (a replay loop
        in Python would otherwise spend more time building result dicts than the GPU spends aligning)."""
        G = np.asarray(guesses, dtype=np.float64)
        g = np.ascontiguousarray(G.transpose(0, 2, 1)).reshape(len(G), 16)       # column-major per pose
        n_pairs = len(g)
        ps = None if pair_scan is None else np.ascontiguousarray(pair_scan, dtype=np.int32)
        res = np.zeros(n_pairs, RESULT_DTYPE)
        self._chk(lib().ndt_align_batch_resident(self._h, None if ps is None else ps.ctypes.data, n_pairs, g.ctypes.data,
                                                 C.cast(res.ctypes.data, C.POINTER(Result))))
        return _unpack_results(res)

    def getFinalTransformation(self): return self.final["T"]
    def hasConverged(self): return self.final["converged"]
    def getFinalNumIteration(self): return self.final["iterations"]
    def getHessian(self): return self.final["H"]

    def source_cells(self, T):
        t = _T(T)
        n = self.source_size()
        keys = np.zeros(n, np.int64); mask = np.zeros(n, np.uint8)
        self._chk(lib().ndt_source_cells(self._h, t.ctypes.data, keys.ctypes.data, mask.ctypes.data, n))
        return keys, mask

    def export_voxels(self):
        n = self.num_voxels()
        keys = np.zeros(n, np.int64); cnt = np.zeros(n, np.int32)
        mean = np.zeros((n, 3)); cov = np.zeros((n, 3, 3)); info = np.zeros((n, 3, 3)); valid = np.zeros(n, np.uint8)
        no = C.c_size_t()
        self._chk(lib().ndt_export_voxels(self._h, keys.ctypes.data, cnt.ctypes.data, mean.ctypes.data, cov.ctypes.data,
                                          info.ctypes.data, valid.ctypes.data, n, C.byref(no)))
        return dict(keys=keys, counts=cnt, mean=mean, cov=cov, info=info, valid=valid)

    # --- PCL compatibility mode
    @staticmethod
    def pcl_params(**kw):
        prm = PclParams()
        lib().ndt_pcl_default_params(C.byref(prm))
        for k, v in kw.items():
            if not hasattr(prm, k):
                raise AttributeError(k)
            setattr(prm, k, v)
        return prm

    def pcl_eval(self, p, hessian=True, **prm):
        """One computeDerivatives pass at transform vector p = [t, euler xyz]: score, gradient, Hessian."""
        p = np.ascontiguousarray(p, np.float64); g = np.zeros(6); H = np.zeros((6, 6))
        sc = C.c_double(); nb = C.c_int64(); pr = self.pcl_params(**prm)
        self._chk(lib().ndt_pcl_eval(self._h, C.byref(pr), p.ctypes.data, C.addressof(sc), g.ctypes.data,
                                     H.ctypes.data if hessian else None, C.addressof(nb)))
        return dict(score=sc.value, g=g, H=H if hessian else None, n_neighbours=nb.value)

    def pcl_align(self, guess, **prm):
        """pcl::NormalDistributionsTransform::computeTransformation (Newton + More-Thuente) from `guess`."""
        t = _T(guess); r = PclResult(); pr = self.pcl_params(**prm)
        rc = lib().ndt_pcl_align(self._h, C.byref(pr), t.ctypes.data, C.byref(r))
        if rc in (-1, -2, -3, -4):
            self._chk(rc)
        return dict(rc=r.status, status=r.status, T=np.array(r.T).reshape(4, 4).T.copy(), p=np.array(r.p), score=r.score,
                    trans_probability=r.trans_probability, g=np.array(r.gradient), H=np.array(r.hessian).reshape(6, 6),
                    n_neighbours=r.n_neighbours, iterations=r.iterations, converged=bool(r.converged), evaluations=r.evaluations)

    def pcl_time_pass(self, p, hessian=True, reps=50, **prm):
        """(us per launch, valid voxels met per point, neighbours per point) of the PCL-mode derivative pass."""
        p = np.ascontiguousarray(p, np.float64); pr = self.pcl_params(**prm)
        us = C.c_double(); vx = C.c_double(); nb = C.c_double()
        self._chk(lib().ndt_pcl_time_pass(self._h, C.byref(pr), p.ctypes.data, int(hessian), int(reps), C.addressof(us), C.addressof(vx), C.addressof(nb)))
        return us.value, vx.value, nb.value

    def save_map(self):
        """The resident voxel map as bytes (layout: include/ndt_mi355.h, ndt_map_save)."""
        need = C.c_size_t()
        self._chk(lib().ndt_map_save(self._h, None, 0, C.byref(need)))
        buf = np.zeros(need.value, np.uint8)
        self._chk(lib().ndt_map_save(self._h, buf.ctypes.data, buf.nbytes, C.byref(need)))
        return buf.tobytes()

    def load_map(self, blob):
        buf = np.frombuffer(blob, np.uint8)
        self._chk(lib().ndt_map_load(self._h, buf.ctypes.data if len(buf) else None, len(buf)))

    # --- measurement helpers
    def profile(self, enable=True): self._chk(lib().ndt_profile(self._h, int(enable)))

    def profile_read(self):
        us = C.c_double(); n = C.c_int64(); hp = C.c_double()
        self._chk(lib().ndt_profile_read(self._h, C.addressof(us), C.addressof(n), C.addressof(hp)))
        return dict(residual_us=us.value, launches=n.value, hits_per_point=hp.value)

    def time_residual(self, T, reps=200):
        """Mean microseconds of one residual-kernel launch at pose T (HIP events around `reps` launches)."""
        t = _T(T); us = C.c_double()
        self._chk(lib().ndt_time_residual(self._h, t.ctypes.data, int(reps), C.addressof(us)))
        return us.value

    def time_chain(self, T, reps=200):
        """Mean microseconds of one chained launch of the single-scan loop (previous step + residual pass) from pose T."""
        t = _T(T); us = C.c_double()
        self._chk(lib().ndt_time_chain(self._h, t.ctypes.data, int(reps), C.addressof(us)))
        return us.value

    def time_residual_batch(self, reps=100):
        """(us per launch, points per launch, hits per point) of the batched residual kernel on the resident batch."""
        us = C.c_double(); pts = C.c_double(); hp = C.c_double()
        self._chk(lib().ndt_time_residual_batch(self._h, int(reps), C.addressof(us), C.addressof(pts), C.addressof(hp)))
        return us.value, pts.value, hp.value

    def synchronize(self): self._chk(lib().ndt_synchronize(self._h))

    # --- multi-GPU (source sharded across ranks, RCCL all-reduce per iteration)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        rc = lib().ndt_comm_unique_id(buf)
        if rc:
            raise NdtError(rc, "ndt_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, n_ranks, uid):
        buf = C.create_string_buffer(bytes(uid), 128)
        self._chk(lib().ndt_comm_init(self._h, rank, n_ranks, buf))

    # --- the same without a library: peer stores into the ranks' mailboxes (ranks of one node)
    def comm_p2p_export(self):
        """Allocates this rank's mailbox; returns its opaque 128-byte handle for the caller to distribute."""
        buf = C.create_string_buffer(128)
        self._chk(lib().ndt_comm_p2p_export(self._h, buf))
        return buf.raw

    def comm_p2p_init(self, rank, n_ranks, handles):
        """handles: the 128-byte handles of all ranks, in rank order."""
        blob = b"".join(bytes(x) for x in handles)
        if len(blob) != 128 * n_ranks:
            raise ValueError("need one 128-byte handle per rank")
        buf = C.create_string_buffer(blob, len(blob))
        self._chk(lib().ndt_comm_p2p_init(self._h, int(rank), int(n_ranks), buf))

    def comm_destroy(self): self._chk(lib().ndt_comm_destroy(self._h))

    def comm_size(self): return lib().ndt_comm_size(self._h)


def information_from_hessian(H, scale=1.0, rel_floor=1e-6):
    """The NDT Hessian as a positive-definite information matrix for PointCloudFactor (factor.h:220-269)."""
    Hc = np.ascontiguousarray(H, dtype=np.float64); out = np.zeros((6, 6))
    rc = lib().ndt_information_from_hessian(Hc.ctypes.data, float(scale), float(rel_floor), out.ctypes.data)
    if rc:
        raise NdtError(rc, "ndt_information_from_hessian")
    return out


def unpack_keys(keys):
    """Packed int64 keys -> (n,3) int32 cells."""
    k = np.asarray(keys, np.int64)
    b = 1 << 20
    return np.stack([((k >> 42) & 0x1FFFFF) - b, ((k >> 21) & 0x1FFFFF) - b, (k & 0x1FFFFF) - b], 1).astype(np.int32)


def pcl_vector_from_matrix(T):
    """[t, euler xyz] of a 4x4 pose in Eigen's eulerAngles(0,1,2) convention (host only)."""
    t = _T(T); p = np.zeros(6)
    lib().ndt_pcl_vector_from_matrix(t.ctypes.data, p.ctypes.data)
    return p


def pcl_matrix_from_vector(p):
    p = np.ascontiguousarray(p, np.float64); t = np.zeros(16)
    lib().ndt_pcl_matrix_from_vector(p.ctypes.data, t.ctypes.data)
    return t.reshape(4, 4).T.copy()
