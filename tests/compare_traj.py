"""Diagnostic (test infrastructure: it uses the oracle): per-frame pose difference between the HIP pipeline and the oracle on a synthetic stream."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kf_ref  # noqa: E402
from slambench_amd import synth  # noqa: E402
from slambench_amd.config import KFusionConfig  # noqa: E402
from slambench_amd.kfusion import KFusion  # noqa: E402
from slambench_amd.metrics import ate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--csr", type=int, default=4)
ap.add_argument("--res", type=int, default=128)
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--noise", type=int, default=1)
a = ap.parse_args()
cfg = KFusionConfig(compute_size_ratio=a.csr, volume_resolution=(a.res,) * 3)
depth, gt, ts = synth.make_sequence(a.frames, seed=a.seed, noise=bool(a.noise))
ref = kf_ref.KFusionRef(cfg.to_c(kf_ref.Config))
kf = KFusion(cfg)
pr, pg = [], []
for i in range(a.frames):
    fl = ref.process_frame(depth[i])
    kf.process_frame(depth[i])
    st = kf.status(i)
    p, q = ref.pose.copy(), st["pose"]
    pr.append(p)
    pg.append(q)
    dR = np.abs(p[:3, :3] - q[:3, :3]).max()
    print("%3d flags %d/%d%d%d dpos %.3e dR %.3e  N %.0f/%.0f  gt_err ref %.4f hip %.4f |x| %.2e" % (
        i, fl, st["tracked"], st["integrated"], st["raycasted"], np.linalg.norm(p[:3, 3] - q[:3, 3]), dR,
        ref.reduction()[28], st["sums"][28], np.linalg.norm(p[:3, 3] - gt[i][:3, 3]),
        np.linalg.norm(q[:3, 3] - gt[i][:3, 3]), 0))
print("ATE ref", ate(np.stack(pr), gt, ts, ts)["rmse"], "hip", ate(np.stack(pg), gt, ts, ts)["rmse"])
