"""Diagnostic: conditioning of the ICP normal equations on the bench stream (last iteration of each frame)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(512,) * 3)
depth, gt, ts = synth.make_sequence(min(n, 100), seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
period = list(range(len(depth))) + list(range(len(depth) - 2, 0, -1))
kf = KFusion(cfg)
for i in range(n):
    kf.process_frame(depth[period[i % len(period)]])
conds, fro, slow, its, exam = [], [], [], [], []
for f in range(4, n):
    st = kf.status(f)
    s = np.asarray(st["sums"], np.float64)
    C = np.zeros((6, 6))
    q = 7
    for i in range(6):
        for j in range(i, 6):
            C[i, j] = C[j, i] = s[q]
            q += 1
    w = np.linalg.eigvalsh(C)
    conds.append(w[-1] / max(w[0], 1e-300))
    fro.append(np.linalg.norm(C) * np.linalg.norm(np.linalg.inv(C)))
    slow.append(st["slow_solves"])
    its.append(st["icp_iterations"])
    exam.append(st["examined_solves"])
conds, fro = np.array(conds), np.array(fro)
print("cond_2 percentiles 5/25/50/75/95/max: ", np.percentile(conds, [5, 25, 50, 75, 95, 100]))
print("fro product percentiles 5/25/50/75/95/max:", np.percentile(fro, [5, 25, 50, 75, 95, 100]))
print("frames with cond_2 > 1e6: %d of %d; with fro product > 0.98e6: %d" % ((conds > 1e6).sum(), len(conds), (fro > 0.98e6).sum()))
print("slow solves per frame: mean %.2f max %d; iterations mean %.1f" % (np.mean(slow), np.max(slow), np.mean(its)))
print("solves per frame that had to examine the smallest singular value: mean %.2f max %d" % (np.mean(exam), np.max(exam)))
print("eigenvalues of the last frame:", w)
