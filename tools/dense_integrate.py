"""Integrate kernel on a volume that lies entirely inside the frustum and in front of the surface: every voxel
is read and written, so the launch moves exactly 8 B per voxel.  Gives the HBM-bound rate of the kernel and a
known byte count to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters against.

usage: dense_integrate.py [V] [repeats]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import native  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = native.lib()
w, h = 640, 480
fx, fy, cx, cy = 481.2, 480.0, 319.5, 239.5
dim = 0.5                                               # a 0.5 m cube, 1 m in front of the camera
depth = torch.full((h, w), 3.0, dtype=torch.float32, device="cuda")   # surface far behind the cube
vol = torch.empty((V, V, V, 2), dtype=torch.int16, device="cuda")
vol[..., 0] = 32766
vol[..., 1] = 0
inv_pose = np.eye(4, dtype=np.float32)
inv_pose[:3, 3] = [-dim / 2, -dim / 2, 1.0]             # volume frame -> camera frame
K = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
res = (C.c_int32 * 3)(V, V, V)
dm = (C.c_float * 3)(dim, dim, dim)
args = (C.c_void_p(vol.data_ptr()), res, dm, 0, V, C.c_void_p(depth.data_ptr()), w, h,
        inv_pose.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p), C.c_float(0.1), C.c_float(100.0), None)
native.check(L.kfh_k_integrate(*args))
torch.cuda.synchronize()
assert int(vol[..., 1].min()) == 1 and int(vol[..., 1].max()) == 1, "every voxel must have been updated once"
t0 = time.perf_counter()
for _ in range(rep):
    native.check(L.kfh_k_integrate(*args))              # each call synchronises
dt = (time.perf_counter() - t0) / rep
print("dense integrate V=%d: %.3f ms per launch (host-timed, includes a 64-byte upload and a sync), "
      "%.1f GVox/s, %.0f GB/s of 8 B/voxel" % (V, dt * 1e3, V ** 3 / dt / 1e9, 8.0 * V ** 3 / dt / 1e9))
assert int(vol[..., 1].min()) == rep + 1
