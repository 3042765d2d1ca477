"""Diagnostic: where one ICP iteration launch spends its time (block 0 stamps, 10 ns ticks)."""
import ctypes as C, os, sys
os.environ["KFH_ICP_STAMPS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(512,) * 3)
n = 30
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
kf = KFusion(cfg)
L = native.lib()
L.kfh_debug_icp_stamps.argtypes = [C.c_void_p, C.c_void_p]
for i in range(n):
    kf.process_frame(depth[i])
    kf.sync()
st = np.zeros((64, 8), np.uint64)
native.check(L.kfh_debug_icp_stamps(kf._ctx, st.ctypes.data_as(C.c_void_p)))
st = st.astype(np.int64)
t0 = st[0, 0]
print("seq  entry(us)  gap_prev  finalize (sum / solve / sync)  project+gather  residual  reduce  store   total")
for q in range(19):
    r = st[q]
    if r[0] == 0:
        continue
    prev_end = st[q - 1, 5] if q > 0 and st[q - 1, 5] else (st[q - 1, 1] if q > 0 else r[0])
    d = lambda a, b: (r[b] - r[a]) / 100.0 if r[a] and r[b] else float("nan")
    print("%3d %9.2f %9.2f %9.2f (%5.2f /%5.2f /%5.2f) %14.2f %9.2f %7.2f %6.2f %7.2f" % (
        q, (r[0] - t0) / 100.0, (r[0] - prev_end) / 100.0, d(0, 1), d(0, 6), d(6, 7), d(7, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5),
        ((r[5] if r[5] else r[1]) - r[0]) / 100.0))
