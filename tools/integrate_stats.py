"""Diagnostic: per-wavefront duration / work statistics of one integrate launch on the bench stream.

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
zchunk = 16 if V <= 512 else 32
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V,) * 3)
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
kf = KFusion(cfg)
L = native.lib()
L.kfh_debug_integrate_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for i in range(n):
    kf.process_frame(depth[i])
kf.sync()
nw = ((V + 63) // 64) * ((V + 3) // 4) * ((V + zchunk - 1) // zchunk) * 4
buf = np.zeros(nw * 4 + 1, np.uint64)
for rep in range(2):
    native.check(L.kfh_debug_integrate_stats(kf._ctx, buf.ctypes.data_as(C.c_void_p), buf.size))
print('boxes kept by the plan launch:', int(buf[-1]) & 0xFFFFFFFF)
st = buf[:-1].reshape(nw, 4).astype(np.int64)
ran = st[:, 0] > 0
print("waves launched %d, of which past the frustum test %d (%.1f%%)" % (nw, ran.sum(), 100.0 * ran.mean()))
st = st[ran]
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0
dur = end - start
print("span of surviving waves %.1f us; start percentiles 10/50/90/max" % end.max(), np.round(np.percentile(start, [10, 50, 90, 100]), 1))
print("wave duration us 10/50/90/99/max", np.round(np.percentile(dur, [10, 50, 90, 99, 100]), 1), "sum %.0f us" % dur.sum())
print("candidates per wave 10/50/90/max", np.percentile(st[:, 2], [10, 50, 90, 100]), "total %.3g" % st[:, 2].sum())
print("written per wave 10/50/90/max", np.percentile(st[:, 3], [10, 50, 90, 100]), "total %.3g (%.1f%% of the volume)" % (st[:, 3].sum(), 100.0 * st[:, 3].sum() / V ** 3))
idle = st[:, 2] == 0
print("surviving waves without a candidate: %d (mean duration %.1f us); with: mean duration %.1f us" % (idle.sum(), dur[idle].mean() if idle.any() else 0, dur[~idle].mean()))
edges = [5, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]
print("surviving waves finished by (us):", {e: int((end <= e).sum()) for e in edges})
print("surviving waves started by (us):", {e: int((start <= e).sum()) for e in edges})
