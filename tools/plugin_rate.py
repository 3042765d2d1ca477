"""Frame rate through the plugin boundary: a synthetic .slam file (Grey, RGB, Depth, ground truth per time step,
as the reference's dataset files are laid out) is replayed by the mini-harness in the reference's call order
(sb_update_frame per sensor frame, sb_process_once, sb_update_outputs), frames coming from the file in host
memory and the pose read back after every frame.  This is the rate a SLAMBench user sees per frame; bench.py's
`value` is the rate of the device pipeline with frames resident in HBM.

  python tools/plugin_rate.py [--frames 300] [--volume 512] [--dir /tmp] [--devices N]
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slambench_amd import harness, slamfile, synth  # noqa: E402
from slambench_amd.config import TUM_FR1_DEPTH_INTRINSICS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--dir", default="/tmp")
ap.add_argument("--depth-only", action="store_true", help="leave the Grey and RGB frames out of the file")
ap.add_argument("--devices", type=int, default=1,
                help="the plugin's --devices: z-slabs inside the plugin (more slabs than visible GPUs share them)")
a = ap.parse_args()

depth, gt, ts = synth.make_sequence(a.frames, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
path = os.path.join(a.dir, "plugin_rate_%d.slam" % os.getpid())
slamfile.write_depth_sequence(path, depth, ts, TUM_FR1_DEPTH_INTRINSICS, gt_poses=gt,
                              with_rgb_and_grey=not a.depth_only)
log = path + ".log"
v = "%d,%d,%d" % (a.volume, a.volume, a.volume)
t0 = time.perf_counter()
r = subprocess.run([harness.MINI_HARNESS_PATH, "-i", path, "-load", harness.PLUGIN_PATH + "=kf", "-o", log, "-kf-v", v]
                   + (["-kf-n", str(a.devices)] if a.devices > 1 else []),
                   capture_output=True, text=True, timeout=600)
wall = time.perf_counter() - t0
if r.returncode != 0:
    sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
    sys.exit(r.returncode)
props, cols, rows = harness.parse_log(open(log).read())
frame_s = rows[10:, cols.index("Duration_Frame")]
out = {
    "frames": int(rows.shape[0]), "volume": a.volume, "devices": a.devices, "file_mb": os.path.getsize(path) / 1e6,
    "mean_duration_frame_ms": float(frame_s.mean() * 1e3), "median_duration_frame_ms": float(np.median(frame_s) * 1e3),
    "frames_per_s_by_duration_frame": float(1.0 / frame_s.mean()),
    "process_wall_s": wall,
    "phase_ms": {c[len("Duration_"):]: float(rows[10:, i].mean() * 1e3)
                 for i, c in enumerate(cols) if c.startswith("Duration_") and c != "Duration_Frame"},
}
os.remove(path)
os.remove(log)
print(json.dumps(out))
