"""SURVEY.md §8(f) row N3, the whole per-frame loop (process.cc:6-153, 318-349): LioPipeline<ScanToMapOdometry> on the
MI355X engine over a synthetic IMU + LiDAR stream, against the same loop written in Python on the CPU oracles
(oracle/ndt_oracle.c for the NDT, oracle/lio_oracle.py for the IMU / optimiser half) and against the true motion."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "lio-slam_amd", os.environ.get("NDT_TEST_EXE_DIR", ""), "lio_pipeline_test")   # (NDT_TEST_EXE_DIR=asan: the host-ASan builds of tools/build_asan.sh)
LEAF, NEED, ACC_W, GYR_W = 0.2, 1.0, 1e-4, 1e-5


def _stream():
    from lio_slam_amd import synth
    rng = np.random.Generator(np.random.PCG64(41))
    scene = synth.room_scene(rng)
    m = synth.Motion(t_move=1.6)
    T_sb0 = synth.se3([0, 0, 0.4], [0.5, 0.3, 0.5])                     # body at rest, in the scene
    T_il = synth.se3([0.0, 0.0, 0.0], [-0.17, 0.0, 0.255])              # TIL_ of process.h:84
    imu = synth.imu_stream(m, 0.0, 4.4, 100, rng, bg=(0.002, -0.001, 0.0015), ba=(0.03, -0.02, 0.01))
    ts = np.arange(0.105, 4.3, 0.1)
    scans = [synth.scan(scene, "vlp16", T_sb0 @ m.pose(t) @ T_il, rng) for t in ts]
    return m, T_il, imu, ts, scans


def _python_pipeline(T_il, imu, ts, scans):
    """The loop of LioPipeline::addScan on the CPU oracles."""
    from oracle import lio_oracle as lo
    from oracle.oracle import Oracle, voxel_downsample
    o = Oracle()
    si = lo.StaticImuInit(init_need_time=NEED)
    aw, gw = ACC_W ** 2 * np.eye(3), GYR_W ** 2 * np.eye(3)
    dq, nxt, last_t_imu = [], 0, -1.0
    init_flag, first_scan, last_imu, pre = False, True, None, None
    last = lo.State(); cur = lo.State(); prior = None
    last_key, n_key = None, 0
    out = {}
    new_pre = lambda ba, bg: lo.ImuPreint(last_imu[4:7], last_imu[1:4], ba, bg, si.cov_acce_n, si.cov_gyro_n, aw, gw)
    ga2, gp2 = (2.0 * 3.14 / 180) ** 2, 0.1 ** 2
    info = np.diag([1 / ga2] * 3 + [1 / gp2] * 3)
    for k, (t, raw) in enumerate(zip(ts, scans)):
        while nxt < len(imu) and imu[nxt, 0] <= t + 0.02:
            dq.append(imu[nxt]); last_t_imu = imu[nxt, 0]; nxt += 1
        if not dq or last_t_imu <= t:
            continue
        imus, imu_time = [], dq[0][0]
        while dq and imu_time < t:
            imu_time = dq[0][0]; imus.append(dq.pop(0))
        if not init_flag:
            for s in imus:
                si.add(s[0], s[1:4], s[4:7])
            if si.success and imus:
                last_imu = imus[-1]; pre = new_pre(si.ba, si.bg); init_flag = True
                cur.t = imus[-1][0]; last = cur.copy()
            continue
        for s in imus:
            pre.propagate(s[0] - last_imu[0], s[4:7], s[1:4]); last_imu = s
        ds = voxel_downsample(raw, LEAF, T_il)
        if first_scan:
            o.add_target(ds, last.transform()); last_key = last.transform(); n_key = 1
            pre = new_pre(si.ba, si.bg); first_scan = False
            out[k] = (last.copy(), n_key, 0)
            continue
        cur = pre.predict(last, si.gravity)
        o.set_source(ds)
        r = o.align(cur.transform())
        assert r["rc"] == 0
        l2, cur, Jr, Er, _ = lo.optimize(pre, si.gravity, last, cur, r["T"], info, prior)
        prior = (cur.copy(), Jr, Er)
        pre = new_pre(cur.ba, cur.bg)
        last = cur.copy()
        pose = last.transform()
        dp = np.linalg.norm(pose[:3, 3] - last_key[:3, 3])
        dth = np.arccos(min(1.0, max(-1.0, 0.5 * (np.trace(last_key[:3, :3].T @ pose[:3, :3]) - 1.0))))
        if dp > 0.3 or dth > 5.0 * 3.14 / 180.0:
            o.add_target(ds, pose); last_key = pose; n_key += 1
        out[k] = (last.copy(), n_key, r["iterations"])
    return out, si


def test_replay_matches_python_loop_and_truth(tmp_path):
    from oracle import lio_oracle as lo
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    m, T_il, imu, ts, scans = _stream()
    path = tmp_path / "stream.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<5d", NEED, LEAF, ACC_W, GYR_W, 0.0)); f.write(np.ascontiguousarray(T_il.T).tobytes())
        f.write(struct.pack("<d", len(imu))); f.write(np.ascontiguousarray(imu).tobytes())
        f.write(struct.pack("<d", len(scans)))
        for t, s in zip(ts, scans):
            f.write(struct.pack("<2d", t, len(s))); f.write(np.ascontiguousarray(s, np.float32).tobytes())
    out = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = {}
    for line in out.stdout.strip().split("\n"):
        v = line.split()
        if v[1] != "skipped":
            rows[int(v[0])] = np.array(v[1:], float)
    ref, si = _python_pipeline(T_il, imu, ts, scans)
    assert si.success and sorted(rows) == sorted(ref) and len(rows) >= 25
    worst_t = worst_truth = 0.0
    for k in sorted(rows):
        st, n_key, iters = ref[k]
        g = rows[k]
        assert int(g[0]) == n_key, (k, g[0], n_key)
        q = g[3:7]; p = g[7:10]
        worst_t = max(worst_t, np.linalg.norm(p - st.p))
        assert np.linalg.norm(p - st.p) < 1e-4, (k, p, st.p)                       # engine loop == oracle loop (north-star pose tolerance)
        assert np.linalg.norm(lo.log_R(lo.q_to_R(q).T @ lo.q_to_R(st.q))) < 1e-5
        # truth at the scan's time.  (The state's own stamp lags by the interval between IMU initialisation and the first
        # scan: process.cc:79 restarts the pre-integration there without moving last_state_.timestamp_ — reproduced.)
        Tt = m.pose(ts[k])
        worst_truth = max(worst_truth, np.linalg.norm(p - Tt[:3, 3]))
        assert np.linalg.norm(p - Tt[:3, 3]) < 0.02, (k, p, Tt[:3, 3])
        assert np.linalg.norm(lo.log_R(lo.q_to_R(q).T @ Tt[:3, :3])) < 0.02
    last = rows[max(rows)]
    assert np.linalg.norm(last[7:10]) > 1.5 and int(last[0]) >= 4                 # it drove, and merged keyframes
    print("worst |engine - oracle loop| %.2e m, worst |engine - truth| %.3f m" % (worst_t, worst_truth))


def test_replay_with_the_ndt_hessian_as_information(tmp_path):
    """SURVEY.md §8(f) N2 inside the loop: PointCloudFactor weighted by the engine's Hessian (made positive definite by
    ndt_information_from_hessian) instead of the constant diagonal of process.cc:205-208.  No oracle loop here (the
    property checked is that the estimate stays on the true motion and the optimiser accepts the matrix every frame)."""
    from oracle import lio_oracle as lo
    m, T_il, imu, ts, scans = _stream()
    path = tmp_path / "stream.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<5d", NEED, LEAF, ACC_W, GYR_W, 1.0)); f.write(np.ascontiguousarray(T_il.T).tobytes())
        f.write(struct.pack("<d", len(imu))); f.write(np.ascontiguousarray(imu).tobytes())
        f.write(struct.pack("<d", len(scans)))
        for t, s in zip(ts, scans):
            f.write(struct.pack("<2d", t, len(s))); f.write(np.ascontiguousarray(s, np.float32).tobytes())
    out = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = {int(v[0]): np.array(v[1:], float) for v in (l.split() for l in out.stdout.strip().split("\n")) if v[1] != "skipped"}
    assert len(rows) >= 25
    for k, g in rows.items():
        Tt = m.pose(ts[k])
        assert np.linalg.norm(g[7:10] - Tt[:3, 3]) < 0.02, (k, g[7:10], Tt[:3, 3])
        assert np.linalg.norm(lo.log_R(lo.q_to_R(g[3:7]).T @ Tt[:3, :3])) < 0.02
