"""ndt_align_static — the one entry point that allocates nothing (include/ndt_mi355.h) — against ndt_align.

Called through the C ABI on a source the caller holds in device memory as three float32 arrays; its partial rows and
Gauss-Newton states live in the library's static device array.  Checked against ndt_set_source + ndt_align on the same
inputs: on MI355X both run the same launches, so the bar here is bit for bit (tighter than the 1e-4 m / 1e-5 rad pose
and 1e-5 relative H, b bars of the other parity tests), on two calls in a row, the second with a smaller input, then a
third one after other work has used the GPU.  The file is named to be collected last."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class DeviceSoA:
    """x[], y[], z[] of a cloud in device memory the CALLER owns (hipMalloc through ctypes)."""

    def __init__(self, hip, pts):
        self.hip, self.n, self.ptr = hip, len(pts), []
        for j in range(3):
            a = np.ascontiguousarray(pts[:, j], np.float32)
            d = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(max(a.nbytes, 4))) == 0
            if a.nbytes:
                assert hip.hipMemcpy(d, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1) == 0
            self.ptr.append(d)

    def free(self):
        for d in self.ptr:
            assert self.hip.hipFree(d) == 0


def _free_bytes(hip):
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _same(a, b):
    assert a["rc"] == b["rc"] == 0 and a["iterations"] == b["iterations"] and a["converged"] == b["converged"]
    for k in ("T", "H", "b"):
        assert np.array_equal(a[k], b[k]), k
    assert a["chi2"] == b["chi2"] and a["n_effective"] == b["n_effective"] and a["last_dx_norm"] == b["last_dx_norm"]


def test_align_static_matches_align_on_two_calls_in_a_row():
    import lio_slam_amd as la
    from conftest import pose_error
    from lio_slam_amd import synth
    hip = ctypes.CDLL("libamdhip64.so")
    c = synth.make_config(1, small=True)
    scan = c["scan"].copy()
    scan[5] = [np.nan, 0, 0]                                   # the caller's arrays are used as they are: skipped in the kernels
    small = np.ascontiguousarray(scan[: len(scan) // 3])       # second call: a smaller input (fewer rows per launch)
    n = la.NDT(max_iterations=20, eps=1e-3)
    n.set_target(c["map"])
    big_d, small_d = DeviceSoA(hip, scan), DeviceSoA(hip, small)
    try:
        # call 1 (big), call 2 (small) back to back; the references are computed afterwards so nothing runs in between.
        # "Allocates nothing": the device's free memory (hipMemGetInfo) is the same before, between and after the calls —
        # a hipMalloc of any size on the path would lower it (the allocator hands out 2 MB-granular pages, nothing is
        # cached on a path that never allocates), and a warm-up call first makes sure lazily loaded code objects are in
        n.align_static(small_d.ptr[0].value, small_d.ptr[1].value, small_d.ptr[2].value, small_d.n, c["T_true"])
        f0 = _free_bytes(hip)
        s1 = n.align_static(big_d.ptr[0].value, big_d.ptr[1].value, big_d.ptr[2].value, big_d.n, c["guess"])
        f1 = _free_bytes(hip)
        s2 = n.align_static(small_d.ptr[0].value, small_d.ptr[1].value, small_d.ptr[2].value, small_d.n, c["T_true"])
        f2 = _free_bytes(hip)
        assert f0 == f1 == f2, ("ndt_align_static changed the device's free memory", f0, f1, f2)
        n.set_source(scan); r1 = n.align(c["guess"])
        n.set_source(small); r2 = n.align(c["T_true"])
        _same(s1, r1); _same(s2, r2)
        assert s1["iterations"] >= 2 and pose_error(s1["T"], c["T_true"])[0] < 0.05
        # a second handle on the same GPU shares the array (calls do not overlap: each one ends synchronised)
        m = la.NDT(max_iterations=7, eps=0.0, nearby=1)
        m.set_target(c["map"])
        s3 = m.align_static(big_d.ptr[0].value, big_d.ptr[1].value, big_d.ptr[2].value, big_d.n, c["guess"])
        m.set_source(scan); r3 = m.align(c["guess"])
        _same(s3, r3)
        assert s3["iterations"] == 7
        # and the first handle again, after the other one wrote the scratch
        s4 = n.align_static(small_d.ptr[0].value, small_d.ptr[1].value, small_d.ptr[2].value, small_d.n, c["T_true"])
        _same(s4, r2)
        # an empty source fails like ndt_align does (too few residuals), a source beyond the limit is rejected up front
        bad = n.align_static(0, 0, 0, 0, c["guess"], check=False)
        assert bad["rc"] == -5
        with pytest.raises(la.NdtError) as ei:
            n.align_static(big_d.ptr[0].value, big_d.ptr[1].value, big_d.ptr[2].value, (1 << 22) + 1, c["guess"])
        assert ei.value.rc == -9 and "NDT_STATIC_MAX_POINTS" in str(ei.value)
        m.close()
    finally:
        big_d.free(); small_d.free()
    n.close()


def test_align_static_from_two_host_threads_takes_turns():
    """Every handle on a GPU shares that GPU's scratch array; the library serialises ndt_align_static calls per device, so two
    host threads calling it on two handles at once get exactly what they get one after the other (before round 3 they raced
    on the rows and states and returned wrong poses with rc 0)."""
    import threading
    import lio_slam_amd as la
    from lio_slam_amd import synth
    hip = ctypes.CDLL("libamdhip64.so")
    c = synth.make_config(1, small=True)
    clouds = [c["scan"], np.ascontiguousarray(c["scan"][::2])]
    handles = [la.NDT(max_iterations=12, eps=0.0), la.NDT(max_iterations=9, eps=0.0, nearby=1)]
    dev = [DeviceSoA(hip, s) for s in clouds]
    try:
        for h in handles:
            h.set_target(c["map"])
        serial = [[handles[i].align_static(dev[i].ptr[0].value, dev[i].ptr[1].value, dev[i].ptr[2].value, dev[i].n, c["guess"]) for _ in range(3)]
                  for i in range(2)]
        out = [[], []]

        def work(i):
            for _ in range(40):
                out[i].append(handles[i].align_static(dev[i].ptr[0].value, dev[i].ptr[1].value, dev[i].ptr[2].value, dev[i].n, c["guess"]))

        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for i in range(2):
            assert len(out[i]) == 40
            for r in out[i]:
                _same(r, serial[i][0])
    finally:
        for d in dev:
            d.free()
        for h in handles:
            h.close()
