"""What a caller that needs the pose after every frame pays for it: frames/s of the default stream with kfh_sync,
with kfh_get_pose, and with nothing between frames (the benchmark's way).  usage: pose_wait_cost.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from slambench_amd import synth  # noqa: E402
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS  # noqa: E402
from slambench_amd.kfusion import KFusion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
depth, gt, ts = synth.make_sequence(40, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
frames = torch.from_numpy(depth.astype(np.int16)).cuda()
order = (list(range(40)) + list(range(38, 0, -1))) * 20
for mode in ("queue ahead", "kfh_sync per frame", "kfh_get_pose per frame", "kfh_get_pose per frame", "kfh_sync per frame"):
    kf = KFusion(KFusionConfig(input_size=(640, 480), intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(512, 512, 512)))
    for i in range(10):
        kf.process_frame(int(frames[order[i]].data_ptr()))
    kf.sync()
    t0 = time.perf_counter()
    for i in range(10, 10 + n):
        kf.process_frame(int(frames[order[i]].data_ptr()))
        if mode.startswith("kfh_sync"):
            kf.sync()
        elif mode.startswith("kfh_get_pose"):
            kf.pose
    kf.sync()
    dt = time.perf_counter() - t0
    print("%-26s %.1f us per frame" % (mode, dt / n * 1e6))
    kf.close()
