"""One seed of tests/test_fuzz_gpu.py frame by frame: HIP against the oracle, the oracle against itself from
several starting positions moved by a micrometre (how sensitive the stream is), flags, iteration counts and the
ICP sums of the last iteration.

usage: fuzz_case.py SEED [frames]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kf_ref  # noqa: E402
from slambench_amd import native, synth  # noqa: E402
from slambench_amd.config import KFusionConfig  # noqa: E402
from slambench_amd.kfusion import KFusion  # noqa: E402
from tests.test_fuzz_gpu import _draw  # noqa: E402
from tests.util import oracle_config  # noqa: E402

seed = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
kw, pose, stream_seed, size, intr = _draw(seed)
print(kw)
cfg = KFusionConfig(**kw)
depth, gt, ts = synth.make_sequence(n, seed=stream_seed, intrinsics=intr, size=size, noise=True)
shifts = [np.array(s, np.float32) * 1e-6 for s in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [1, 1, 1], [-1, 2, -1])]
ref = kf_ref.KFusionRef(oracle_config(cfg))
ref.set_pose(pose)
nudged = []
for s in shifts:
    r = kf_ref.KFusionRef(oracle_config(cfg))
    p = pose.copy()
    p[:3, 3] += s
    r.set_pose(p)
    nudged.append((r, p))
kf = KFusion(cfg)
kf.set_pose(pose)
for i in range(n):
    want = ref.process_frame(depth[i])
    kf.process_frame(depth[i])
    st = kf.status(i)
    yards = []
    flags = []
    for r, p in nudged:
        flags.append(r.process_frame(depth[i]))
        yards.append(float(np.linalg.norm(r.pose[:3, 3] - p[:3, 3] + pose[:3, 3] - ref.pose[:3, 3])))
    d = float(np.linalg.norm(st["pose"][:3, 3] - ref.pose[:3, 3]))
    rs = ref.reduction()
    print("frame %2d  hip-ref %.3e  flags hip %d%d%d ref %d nudged %s  iters hip %d ref %d slow %d  yards %s" % (
        i, d, st["tracked"], st["integrated"], st["raycasted"], want, flags, st["icp_iterations"],
        ref.icp_iterations(), st["slow_solves"], " ".join("%.1e" % y for y in yards)))
    print("          sums hip: err %.6e valid %.0f | ref: err %.6e valid %.0f" % (
        st["sums"][0], st["sums"][28], rs[0], rs[28]))
    if i >= 3:
        vol_equal = np.array_equal(kf.download(native.BUF_VOLUME), ref.volume())
        mv = kf.download(native.BUF_MODEL_VERTEX)[..., :3]
        dv = np.abs(mv - ref.model_vertex()).max()
        print("          volume identical %s, model vertex max diff %.3e" % (vol_equal, dv))
