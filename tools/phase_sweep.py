"""Diagnostic: mean phase times (HIP events) on the bench stream; environment knobs are read by the library."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
V = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V,) * 3)
depth, gt, ts = synth.make_sequence(min(n, 100), seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
period = list(range(len(depth))) + list(range(len(depth) - 2, 0, -1))
import torch
d = torch.from_numpy(depth.view(np.int16)).cuda()
kf = KFusion(cfg)
kf.set_timing(1)
for i in range(n):
    kf.process_frame(int(d[period[i % len(period)]].data_ptr()))
kf.sync()
ph = np.array([kf.phase_ms(f) for f in range(20, n)])
integ = np.array([kf.status(f)["integrated"] for f in range(20, n)], bool)
print("env", {k: v for k, v in os.environ.items() if k.startswith("KFH_")},
      "pre %.1f track %.1f integrate(when run) %.1f raycast %.1f us; frame %.1f us" % (
          ph[:, 0].mean() * 1e3, ph[:, 1].mean() * 1e3, ph[integ, 2].mean() * 1e3, ph[:, 3].mean() * 1e3, ph.sum(1).mean() * 1e3))
