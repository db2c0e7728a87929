"""Seeded synthetic Velodyne-shaped clouds for tests and benchmarks.

The sensor model follows the simulator the reference is fed by
(src/velodyne_gazebo_plugins/src/GazeboRosVelodyneLaser.cpp:332-352:
x = r cos(phi) cos(theta), y = r cos(phi) sin(theta), z = r sin(phi); theta
uniform in [-pi, pi), phi uniform between the min and max elevation; range gate
0.9-130 m, Gaussian range noise) and the scenes follow the shapes of
src/scout_gazebo/worlds/lab.world (a 22.75 x 11.1 x 2.5 m room) for the small
configurations and an open 200 x 200 m yard with boxes and cylinders for the
large ones.  Everything is NumPy on the host; nothing here is on the hot path.
"""
import numpy as np

SENSORS = {
    # name: (rings, min_elev_deg, max_elev_deg, azimuth_steps)
    "vlp16": (16, -15.0, 15.0, 1875),       # 30 000 rays  (VLP-16.urdf.xacro:4)
    "hdl64": (64, -24.8, 2.0, 1875),        # 120 000 rays
    "dense128": (128, -24.8, 2.0, 15625),   # 2 000 000 rays, the HDL-64 shape's field of view (nearly every ray returns)
}


def seed_for(config_index):
    return 20241220 + int(config_index)


class Scene:
    """Ground plane z=0 plus axis-aligned boxes, vertical cylinders and an optional enclosing room."""

    def __init__(self, boxes=(), cylinders=(), room=None, ground=True):
        self.boxes = np.asarray(boxes, np.float64).reshape(-1, 6)       # xmin ymin zmin xmax ymax zmax
        self.cyl = np.asarray(cylinders, np.float64).reshape(-1, 4)     # cx cy radius height
        self.room = None if room is None else np.asarray(room, np.float64)  # same 6 numbers, hit from inside
        self.ground = ground

    def cast(self, o, d, rmax=130.0):
        """Range of the first hit along unit rays d (n,3) from origin o (3,); inf when nothing is hit."""
        n = d.shape[0]
        out = np.full(n, np.inf)
        for s in range(0, n, 1 << 18):
            out[s:s + (1 << 18)] = self._cast(o, d[s:s + (1 << 18)], rmax)
        return out

    def _cast(self, o, d, rmax):
        n = d.shape[0]
        best = np.full(n, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            if self.ground:
                t = -o[2] * inv[:, 2]
                t[~(t > 1e-9)] = np.inf
                best = np.minimum(best, t)
            if self.room is not None:  # inside-out slab test: the exit distance
                t1 = (self.room[None, :3] - o[None, :]) * inv
                t2 = (self.room[None, 3:] - o[None, :]) * inv
                tex = np.nanmin(np.maximum(t1, t2), axis=1)
                tex[~(tex > 1e-9)] = np.inf
                best = np.minimum(best, tex)
            for b in self.boxes:
                t1 = (b[None, :3] - o[None, :]) * inv
                t2 = (b[None, 3:] - o[None, :]) * inv
                tn = np.nanmax(np.minimum(t1, t2), axis=1)
                tf = np.nanmin(np.maximum(t1, t2), axis=1)
                hit = (tn <= tf) & (tn > 1e-9)
                best = np.where(hit & (tn < best), tn, best)
            for c in self.cyl:
                ox, oy = o[0] - c[0], o[1] - c[1]
                a = d[:, 0] ** 2 + d[:, 1] ** 2
                bq = 2.0 * (ox * d[:, 0] + oy * d[:, 1])
                cq = ox * ox + oy * oy - c[2] ** 2
                disc = bq * bq - 4 * a * cq
                t = (-bq - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2 * a)
                z = o[2] + t * d[:, 2]
                hit = (disc > 0) & (t > 1e-9) & (z >= 0) & (z <= c[3])
                best = np.where(hit & (t < best), t, best)
        best[best > rmax] = np.inf
        return best


def room_scene(rng):
    """Office-like box room (walls 22.75 x 11.1 m, 2.5 m high) with 0.15 m partitions and some furniture."""
    L, W, Hh = 22.75, 11.1, 2.5
    boxes = [
        [6.0, -W / 2, 0, 6.15, 1.5, Hh], [-5.0, -1.0, 0, -4.85, W / 2, Hh], [13.0, -2.0, 0, 13.15, W / 2, Hh],
    ]
    for _ in range(10):
        cx, cy = rng.uniform(-3, L - 6), rng.uniform(-W / 2 + 1, W / 2 - 1)
        sx, sy, sz = rng.uniform(0.4, 1.6, 3)
        boxes.append([cx - sx / 2, cy - sy / 2, 0, cx + sx / 2, cy + sy / 2, sz])
    room = [-L / 2 + 5.0, -W / 2, -0.01, L / 2 + 5.0, W / 2, Hh]
    return Scene(boxes=boxes, room=room, ground=True)


def yard_scene(rng):
    """Outdoor 200 x 200 m ground with 40 boxes (2-10 m) and 30 cylinders."""
    boxes, cyl = [], []
    for _ in range(40):
        cx, cy = rng.uniform(-95, 95, 2)
        if abs(cy) < 4.0:      # keep the driving corridor free
            cy += 8.0 * np.sign(cy if cy else 1.0)
        sx, sy, sz = rng.uniform(2, 10, 3)
        boxes.append([cx - sx / 2, cy - sy / 2, 0, cx + sx / 2, cy + sy / 2, sz])
    for _ in range(30):
        cx, cy = rng.uniform(-95, 95, 2)
        if abs(cy) < 3.0:
            cy += 6.0 * np.sign(cy if cy else 1.0)
        cyl.append([cx, cy, rng.uniform(0.15, 0.6), rng.uniform(3, 12)])
    fence = [[-100, -100, 0, 100, -99.7, 3], [-100, 99.7, 0, 100, 100, 3],
             [-100, -100, 0, -99.7, 100, 3], [99.7, -100, 0, 100, 100, 3]]
    return Scene(boxes=boxes + fence, cylinders=cyl, ground=True)


def ray_dirs(sensor):
    rings, emin, emax, naz = SENSORS[sensor]
    phi = np.deg2rad(np.linspace(emin, emax, rings))
    theta = -np.pi + 2 * np.pi * np.arange(naz) / naz
    cp, sp = np.cos(phi)[:, None], np.sin(phi)[:, None]
    d = np.stack([cp * np.cos(theta)[None, :], cp * np.sin(theta)[None, :], np.broadcast_to(sp, (rings, naz))], -1)
    return d.reshape(-1, 3)   # ring-major, azimuth fastest


def scan(scene, sensor, T_ws, rng, sigma=0.01, rmin=0.9, rmax=130.0):
    """One scan from sensor pose T_ws (4x4, sensor->world).  Returns float32 (n,3) in the sensor frame."""
    d_s = ray_dirs(sensor)
    R, t = T_ws[:3, :3], T_ws[:3, 3]
    r = scene.cast(t, d_s @ R.T, rmax)
    ok = np.isfinite(r) & (r >= rmin)
    r = r[ok] + rng.normal(0.0, sigma, int(ok.sum()))
    return (d_s[ok] * r[:, None]).astype(np.float32)


def se3(rotvec, trans):
    """4x4 from a rotation vector and a translation (Rodrigues)."""
    w = np.asarray(rotvec, np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = trans
    return T


def perturbation(rng, t_lo=0.05, t_hi=0.30, a_lo_deg=0.5, a_hi_deg=3.0):
    """'IMU-predicted' error: random direction, |dt| ~ U(t_lo,t_hi) m, |dtheta| ~ U(a_lo,a_hi) deg."""
    dt = rng.normal(size=3); dt *= rng.uniform(t_lo, t_hi) / np.linalg.norm(dt)
    dw = rng.normal(size=3); dw *= np.deg2rad(rng.uniform(a_lo_deg, a_hi_deg)) / np.linalg.norm(dw)
    return se3(dw, dt)


def trajectory(n, step=0.5, z=1.0, start=(-40.0, 0.0), arc_after=60, arc_radius=40.0):
    """Straight then arc, one pose every `step` metres; sensor `z` above ground."""
    poses, x, y, yaw = [], start[0], start[1], 0.0
    for i in range(n):
        poses.append(se3([0, 0, yaw], [x, y, z]))
        if i >= arc_after:
            yaw += step / arc_radius
        x += step * np.cos(yaw); y += step * np.sin(yaw)
    return poses


def transform(T, pts):
    """float32 world points the way the engine's keyframe merge makes them (float64 math, one rounding)."""
    p = pts.astype(np.float64)
    return (p @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


def build_map(scene, sensor, poses, n_points, rng, sigma=0.01):
    """Accumulate scans at `poses` in the world frame and decimate uniformly to exactly n_points."""
    chunks = [transform(T, scan(scene, sensor, T, rng, sigma)) for T in poses]
    allp = np.concatenate(chunks)
    if allp.shape[0] < n_points:
        raise ValueError(f"only {allp.shape[0]} points from {len(poses)} scans, need {n_points}")
    keep = np.sort(rng.choice(allp.shape[0], n_points, replace=False))
    return allp[keep]


def to_point32(xyz):
    """Pack xyz into the reference's 32-byte slam_czc::Point records (utility.h:54-61)."""
    n = xyz.shape[0]
    rec = np.zeros((n, 8), np.float32)
    rec[:, :3] = xyz
    rec[:, 3] = 1.0
    return rec


def make_config(index, small=False):
    """Inputs for BASELINE.json configs[index]: dict(map=..., scan=..., T_true=..., guess=...)."""
    rng = np.random.Generator(np.random.PCG64(seed_for(index)))
    if index in (0, 1):
        scene = room_scene(rng)
        poses = [se3([0, 0, 0.02 * i], [0.25 * i, 0.05 * i, 0.6]) for i in range(12)]
        T_true = poses[1] if index == 0 else poses[6]
        if index == 0:
            m = transform(poses[0], scan(scene, "vlp16", poses[0], rng))
        else:
            m = build_map(scene, "vlp16", poses, 200_000 if not small else 50_000, rng)
        s = scan(scene, "vlp16", T_true, rng)
    elif index == 2:
        scene = yard_scene(rng)
        n_pose, n_map = (40, 2_000_000) if not small else (6, 200_000)
        poses = trajectory(n_pose, step=0.5)
        m = build_map(scene, "hdl64", poses, n_map, rng)
        T_true = poses[n_pose // 2].copy(); T_true[:3, 3] += [0.21, 0.13, 0.0]
        s = scan(scene, "hdl64", T_true, rng)
    else:
        raise ValueError("configs 3 and 4 are assembled by bench.py from these pieces")
    guess = T_true @ perturbation(rng)
    return dict(map=m, scan=s, T_true=T_true, guess=guess, scene=scene, poses=poses, rng=rng)


# ------------------------------------------------------------------ IMU + LiDAR streams
GRAVITY = np.array([0.0, 0.0, -9.81])      # world gravity; a level IMU at rest measures +9.81 on z


class Motion:
    """Body (IMU) pose over time: at rest until t_move, then a smooth drive with gentle yaw / roll / pitch.
    The world frame is the body frame at rest (the pipeline's map frame, process.cc:75-85)."""

    def __init__(self, t_move=1.5, accel=0.6, v_max=1.2, yaw_amp=0.25, yaw_rate=0.35, tilt=0.02):
        self.t_move, self.accel, self.v_max, self.yaw_amp, self.yaw_rate, self.tilt = t_move, accel, v_max, yaw_amp, yaw_rate, tilt

    def _s(self, tau):     # distance driven: constant acceleration up to v_max, then constant speed
        t1 = self.v_max / self.accel
        return np.where(tau < t1, 0.5 * self.accel * tau ** 2, 0.5 * self.accel * t1 ** 2 + self.v_max * (tau - t1))

    def pose(self, t):
        tau = max(0.0, t - self.t_move)
        s = float(self._s(np.float64(tau)))
        yaw = self.yaw_amp * (1 - np.cos(self.yaw_rate * tau)) * 0.5
        ramp = tau * tau / (1.0 + tau * tau)          # smooth start: no velocity step anywhere
        roll = self.tilt * np.sin(0.9 * tau) * ramp; pitch = self.tilt * np.sin(0.7 * tau) * ramp
        p = np.array([s * np.cos(0.15 * np.sin(0.2 * tau)), 0.6 * (1 - np.cos(0.25 * tau)), 0.01 * np.sin(0.5 * tau) * ramp])
        Rz = se3([0, 0, yaw], [0, 0, 0])[:3, :3]; Ry = se3([0, pitch, 0], [0, 0, 0])[:3, :3]; Rx = se3([roll, 0, 0], [0, 0, 0])[:3, :3]
        T = np.eye(4); T[:3, :3] = Rz @ Ry @ Rx; T[:3, 3] = p
        return T


def _log_so3(R):
    c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2)); th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * w if th < 1e-9 else th / (2 * np.sin(th)) * w


def imu_stream(motion, t0, t1, rate, rng, bg=(0, 0, 0), ba=(0, 0, 0), gyro_sigma=1e-3, acce_sigma=1e-2):
    """(t, gyro[3], acce[3]) rows at `rate` Hz: body angular rate and specific force R^T (a - g) by central differences
    of the analytic motion, plus constant biases and white noise."""
    h = 1e-4
    ts = t0 + np.arange(int(round((t1 - t0) * rate)) + 1) / rate
    out = np.zeros((len(ts), 7))
    for i, t in enumerate(ts):
        Tm, T0, Tp = motion.pose(t - h), motion.pose(t), motion.pose(t + h)
        w = _log_so3(Tm[:3, :3].T @ Tp[:3, :3]) / (2 * h)          # body frame (to first order around t)
        a = (Tp[:3, 3] - 2 * T0[:3, 3] + Tm[:3, 3]) / (h * h)
        out[i, 0] = t
        out[i, 1:4] = w + np.asarray(bg) + rng.normal(0, gyro_sigma, 3)
        out[i, 4:7] = T0[:3, :3].T @ (a - GRAVITY) + np.asarray(ba) + rng.normal(0, acce_sigma, 3)
    return out
