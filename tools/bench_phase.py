"""Micro-benchmark of single phases on a populated 640x480 / V^3 context (used under rocprofv3 --pmc).

usage: bench_phase.py [integrate|raycast|track|all] [V] [repeats]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion

what = sys.argv[1] if len(sys.argv) > 1 else "all"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V,) * 3)
n = 24
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
kf = KFusion(cfg)
for i in range(n):
    kf.process_frame(depth[i])
kf.sync()
kf.set_timing(1)
# one more frame, phase by phase, repeated (frame index n is even -> integration runs)
kf.set_depth(depth[n - 1])
acc = np.zeros(4)
cnt = 0
for r in range(rep):
    kf.preprocessing()
    kf.tracking()
    kf.integration()
    kf.raycasting()
    kf.end_frame(); kf.end_frame()      # keep the frame index even so that integration runs every time
    ms = kf.phase_ms(kf.frame - 2)
    if r >= 2:
        acc += ms
        cnt += 1
print("V=%d mean phase ms over %d repeats: pre %.4f track %.4f integrate %.4f raycast %.4f" % ((V, cnt) + tuple(acc / cnt)))
st = kf.status(kf.frame - 2)
print("iters", st["icp_iterations"], "tracked", st["tracked"], "integrated", st["integrated"])
