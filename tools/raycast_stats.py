"""Diagnostic: per-wavefront duration / batch / tail statistics of one raycast launch on the bench stream.

Needs a diagnostic build: make -C slambench_amd/csrc clean && make -C slambench_amd/csrc STAMPS=1 (rebuild plainly afterwards).
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
V = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V,) * 3)
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
kf = KFusion(cfg)
L = native.lib()
L.kfh_debug_raycast_stats.argtypes = [C.c_void_p, C.c_void_p]
for i in range(n):
    kf.process_frame(depth[i])
kf.sync()
w, h = cfg.compute_size
nb = ((w + 15) // 16) * ((h + 15) // 16)
buf = np.zeros(nb * 16 + 1, np.uint64)
for rep in range(2):
    native.check(L.kfh_debug_raycast_stats(kf._ctx, buf.ctypes.data_as(C.c_void_p)))
st = buf[:nb * 16].reshape(nb * 4, 4).astype(np.int64)
samples, skipped = (st[:, 2] >> 32).sum(), (st[:, 3] >> 32).sum()
heavy_rounds, heavy_us = (st[:, 2] >> 8) & 0xFF, ((st[:, 2] >> 16) & 0xFFFF) / 100.0
pre_us, post_us = ((st[:, 3] >> 8) & 0xFFF) / 100.0, ((st[:, 3] >> 20) & 0xFFF) / 100.0
st[:, 2] &= 0xFF
st[:, 3] &= 0xFF
print('samples in the first pass: %d (%.1f per ray), skipped by brick marks: %d (%.1f%%)' % (samples, samples / (w * h), skipped, 100.0 * skipped / max(samples, 1)))
print('rays handed to the second pass:', int(buf[-1]) & 0xFFFFFFFF)
t0 = st[:, 0].min()
start = (st[:, 0] - t0) / 100.0
end = (st[:, 1] - t0) / 100.0
dur = end - start
print("waves %d; kernel span %.1f us; wave start: median %.1f max %.1f us" % (len(st), end.max(), np.median(start), start.max()))
print("wave duration us: percentiles 10/50/90/99/max", np.round(np.percentile(dur, [10, 50, 90, 99, 100]), 1))
print("batches per wave: percentiles 10/50/90/99/max", np.percentile(st[:, 2], [10, 50, 90, 99, 100]))
print("rays handed over per wave: percentiles 10/50/90/99/max", np.percentile(st[:, 3], [10, 50, 90, 99, 100]))
print("sum of wave durations %.0f us; per SIMD (1024) %.1f us" % (dur.sum(), dur.sum() / 1024))
edges = [0, 5, 10, 15, 20, 30, 40, 50, 60, 80, 120]
print("waves finishing by (us):", {e: int((end <= e).sum()) for e in edges})
long_ = dur > np.percentile(dur, 95)
print("slowest 5%%: batches mean %.1f, tail mean %.1f, us per round %.2f" % (
    st[long_, 2].mean(), st[long_, 3].mean(), (dur[long_] / (st[long_, 2] + st[long_, 3]).clip(1)).mean()))
print("all: us per round %.2f" % (dur / (st[:, 2] + st[:, 3]).clip(1)).mean())
print("rounds with voxel fetches per wave 10/50/90:", np.percentile(heavy_rounds, [10, 50, 90]), "their time us 10/50/90:", np.round(np.percentile(heavy_us, [10, 50, 90]), 1))
light = st[:, 2] - heavy_rounds
print("mark-only rounds per wave 10/50/90:", np.percentile(light, [10, 50, 90]))
print("time before the loop us 10/50/90:", np.round(np.percentile(pre_us, [10, 50, 90]), 2), "after it:", np.round(np.percentile(post_us, [10, 50, 90]), 2))
print("sums over waves (us): total %.0f, heavy rounds %.0f, before %.0f, after %.0f, rest (mark-only rounds) %.0f" % (
    dur.sum(), heavy_us.sum(), pre_us.sum(), post_us.sum(), dur.sum() - heavy_us.sum() - pre_us.sum() - post_us.sum()))
