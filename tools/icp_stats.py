"""Diagnostic: ICP iterations / slow solves per frame and event-timed phases on the bench workload."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion
V = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = 40
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V, V, V))
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
kf = KFusion(cfg)
kf.set_timing(1)
for i in range(n):
    kf.process_frame(depth[i])
for i in range(n):
    st = kf.status(i)
    print(i, "iters", st["icp_iterations"], "slow", st["slow_solves"], "tracked", st["tracked"], "phases_ms", np.round(kf.phase_ms(i), 3))
