"""Diagnostic: where the ICP iteration launches of one frame spend their time.

Every block stamps the 100 MHz real-time counter at: 0 entry, 1 rows staged (first barrier), 2 solver
wavefront done (rows folded, 6x6 solved, pose updated), 7 pixel wavefront 1 done with its preparation
(speculative projection + gather, input normals), 3 second barrier passed, 4 exact projection done and
model map gathered, 5 residuals accumulated, 6 partial row stored; 8 / 9 / 10: wavefront 4 prepared, residuals
done, reduced.  Printed per launch: median and
maximum over the blocks, in microseconds relative to the first block's entry.

Needs a library built with the stamps: make -C slambench_amd/csrc clean && make -C slambench_amd/csrc STAMPS=1
(and a plain rebuild afterwards).
"""
import ctypes as C, os, sys
os.environ["KFH_ICP_STAMPS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import native, synth
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS
from slambench_amd.kfusion import KFusion

LAUNCHES, BLOCKS, SLOTS = 64, 320, 14
cfg = KFusionConfig(intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(512,) * 3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
depth, gt, ts = synth.make_sequence(n, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
kf = KFusion(cfg)
L = native.lib()
L.kfh_debug_icp_stamps.argtypes = [C.c_void_p, C.c_void_p]
for i in range(n):
    kf.process_frame(depth[i])
    kf.sync()
st = np.zeros((LAUNCHES, BLOCKS, SLOTS), np.uint64)
native.check(L.kfh_debug_icp_stamps(kf._ctx, st.ctypes.data_as(C.c_void_p)))
st = st.astype(np.int64)
t_frame = st[0, :, 0][st[0, :, 0] > 0].min()
order = [0, 1, 2, 7, 8, 3, 4, 5, 9, 10, 6]
names = ["entry", "staged", "solved", "prep w1", "prep w4", "barrier2", "project", "resid w1", "resid w4", "reduc w4", "end"]
print("launch blocks start(us)   gap |" + " |".join("%11s" % n for n in names))
prev_end = None
for q in range(19):
    r = st[q]
    act = r[:, 0] > 0
    if not act.any():
        continue
    r = r[act]
    t0 = r[:, 0].min()
    cols = []
    for s_ in order:
        v = r[:, s_]
        v = v[v >= t0]
        cols.append(((np.median(v) - t0) / 100.0, (v.max() - t0) / 100.0) if v.size else (float("nan"),) * 2)
    end = r[:, :12][r[:, :12] >= t0].max()
    print("%4d %6d %9.2f %6.2f |" % (q, act.sum(), (t0 - t_frame) / 100.0,
                                    (t0 - prev_end) / 100.0 if prev_end else 0.0)
          + " |".join(" %5.2f %5.2f" % c for c in cols))
    prev_end = end
    ok = (r[:, 13] > r[:, 12]) & (r[:, 6] > r[:, 0])
    if ok.any():
        clk = (r[ok, 13] - r[ok, 12]) / ((r[ok, 6] - r[ok, 0]) / 100.0)   # shader cycles per microsecond
        print("        shader clock inside this launch: median %.0f MHz (min %.0f, max %.0f)" % (np.median(clk), clk.min(), clk.max()))
st_ = kf.status(n - 1)
print("iterations", st_["icp_iterations"], "tracked", st_["tracked"])
