"""Deterministic synthetic depth streams for the KFusion path.

No dataset ships with the reference (ICL-NUIM / TUM are downloaded by
reference framework/makefiles/dataset-utils.makefile), so tests and the benchmark use an analytic
scene: a box room centred in the 8 m volume, two spheres and one tilted plane, seen by a pinhole
camera that follows a smooth Lissajous trajectory.  Depth is the planar z of the first hit in
millimetres (uint16), the unit the ICL-NUIM generator writes (reference
framework/tools/dataset-tools/ICLNUIM.cpp:96-116), with optional Kinect-like noise and holes.

Poses are camera-to-world 4x4, row-major, x right, y down, z forward; frame 0 is the identity
rotation at `centre`, which is KinectFusion's own initial pose (volume-direction).
"""
import numpy as np


def lissajous_trajectory(n_frames, centre=(4.0, 4.0, 4.0), amp=0.3, max_angle_deg=10.0, fps=30.0,
                         period_s=10.0):
    """Smooth translation (+-amp m) and yaw/pitch (+-max_angle) that start at rest at `centre`."""
    t = np.arange(n_frames, dtype=np.float64) / fps
    w = 2.0 * np.pi / period_s
    ease = 1.0 - np.cos(np.minimum(t / 1.0, 1.0) * np.pi / 2.0) ** 2  # 0 -> 1 during the first second
    tx = amp * np.sin(w * t) * ease
    ty = 0.5 * amp * np.sin(2.0 * w * t) * ease
    tz = 0.5 * amp * (1.0 - np.cos(w * t)) * ease
    yaw = np.deg2rad(max_angle_deg) * np.sin(w * t + 0.3) * ease
    pitch = 0.5 * np.deg2rad(max_angle_deg) * np.sin(1.5 * w * t) * ease
    poses = np.zeros((n_frames, 4, 4), np.float64)
    for i in range(n_frames):
        cy, sy = np.cos(yaw[i]), np.sin(yaw[i])
        cp, sp = np.cos(pitch[i]), np.sin(pitch[i])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        poses[i, :3, :3] = Ry @ Rx
        poses[i, :3, 3] = np.asarray(centre) + np.array([tx[i], ty[i], tz[i]])
        poses[i, 3, 3] = 1.0
    return poses


class SynthRoom:
    """Analytic scene; `render_depth` returns uint16 millimetres at the requested resolution."""

    def __init__(self, seed=1, centre=(4.0, 4.0, 4.0), half=2.4):
        self.seed = int(seed)
        c = np.asarray(centre, np.float64)
        self.lo, self.hi = c - half, c + half
        rng = np.random.default_rng(1000 + self.seed)
        j = rng.uniform(-0.15, 0.15, size=(2, 3))
        self.spheres = [
            (c + np.array([0.6, 0.5, 1.6]) + j[0], 0.5),
            (c + np.array([-0.8, 0.9, 1.9]) + j[1], 0.4),
        ]
        n = np.array([0.0, -0.94, -0.34]) + rng.uniform(-0.03, 0.03, size=3)
        self.plane_n = n / np.linalg.norm(n)
        self.plane_c = float(self.plane_n @ (c + np.array([0.0, 1.6, 1.0])))

    def render_depth(self, pose, intrinsics, size=(640, 480), noise=False, frame_index=0):
        w, h = size
        fx, fy, cx, cy = intrinsics
        xs = (np.arange(w, dtype=np.float64) - cx) / fx
        ys = (np.arange(h, dtype=np.float64) - cy) / fy
        dc = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None], np.ones((h, w))), axis=-1)
        R, o = pose[:3, :3], pose[:3, 3]
        d = dc @ R.T                                     # world direction, camera-frame z == 1
        with np.errstate(divide="ignore", invalid="ignore"):
            # room interior: first wall along the ray
            t_wall = np.where(d > 0, (self.hi - o) / d, np.where(d < 0, (self.lo - o) / d, np.inf))
            t = t_wall.min(axis=-1)
            # tilted plane (both faces)
            denom = d @ self.plane_n
            tp = (self.plane_c - o @ self.plane_n) / denom
            t = np.where((tp > 1e-6) & (tp < t), tp, t)
            # spheres
            dd = (d * d).sum(-1)
            for sc, sr in self.spheres:
                oc = o - sc
                b = d @ oc
                disc = b * b - dd * (oc @ oc - sr * sr)
                ts = (-b - np.sqrt(np.maximum(disc, 0.0))) / dd
                t = np.where((disc > 0) & (ts > 1e-6) & (ts < t), ts, t)
        z = t                                            # planar depth because d_cam.z == 1
        if noise:
            rng = np.random.default_rng((self.seed << 20) + frame_index)
            z = z + rng.normal(0.0, 1.0, z.shape) * 0.0015 * z * z
            holes = rng.random(z.shape) < 0.02
            z = np.where(holes, 0.0, z)
        mm = np.rint(z * 1000.0)
        mm = np.where((mm > 0) & (mm < 65535), mm, 0)
        return mm.astype(np.uint16)


def make_sequence(n_frames, seed=1, intrinsics=(481.2, 480.0, 319.5, 239.5), size=(640, 480), noise=False,
                  amp=0.3, max_angle_deg=10.0, period_s=10.0):
    """Returns (depth[n, h, w] uint16 mm, poses[n, 4, 4] float64 ground truth, timestamps[n] seconds)."""
    room = SynthRoom(seed)
    poses = lissajous_trajectory(n_frames, amp=amp, max_angle_deg=max_angle_deg, period_s=period_s)
    depth = np.empty((n_frames, size[1], size[0]), np.uint16)
    for i in range(n_frames):
        depth[i] = room.render_depth(poses[i], intrinsics, size, noise=noise, frame_index=i)
    ts = np.arange(n_frames, dtype=np.float64) / 30.0
    return depth, poses, ts
