"""The in-process z-slab group (kfh_group_*) against the unsharded pipeline at full size on ONE device: the first
frames of the synthetic 640x480 stream through one context and through groups of 2 members (launches queued by the
calling thread) and 4 members (one host thread per member), all on device 0.  Prints the largest difference of the
final pose, whether the same frames tracked, and the wall time per frame.  usage: group_check.py [V=2048]
"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from slambench_amd import synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion, GroupKFusion
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = 14
cfg = KFusionConfig(input_size=(640, 480), intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V, V, V))
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
one = KFusion(cfg)
for i in range(n):
    one.process_frame(depth[i])
one.sync()
p1 = one.pose.copy()
tr1 = [one.status(f)["tracked"] for f in range(n)]
one.close()
for members, threads in ((2, "0"), (4, "1")):
    import os
    os.environ["KFH_GROUP_THREADS"] = threads
    g = GroupKFusion(cfg, [0] * members)
    t0 = time.perf_counter()
    for i in range(n):
        g.process_frame(depth[i])
    g.sync()
    dt = time.perf_counter() - t0
    pg = g.pose
    trg = [g.members[0].status(f)["tracked"] for f in range(n)]
    print("V", V, "members", members, "threads", threads, "max |pose diff|", float(np.abs(pg - p1).max()),
          "tracked equal", trg == tr1, "ms/frame", dt / n * 1e3, flush=True)
    g.close()
