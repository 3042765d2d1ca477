"""Run a dataset through the plugin boundary and print a log in the reference harness' layout, accuracy columns
included when the file holds ground truth:

  python tools/slambench_run.py living_room_traj2_loop.slam -o lr_kt2.log -- -v 512,512,512 -m 0.1

Everything after `--` goes to the plugin (its own option names).  The frames run through
kfh_mini_slambench -> c_sb_* -> libkfusion-hip-library.so; ATE / RPE are computed from the positions it logs.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import harness, slamfile  # noqa: E402
from slambench_amd.metrics import frame_metrics  # noqa: E402

argv = sys.argv[1:]
plugin_args = []
if "--" in argv:
    i = argv.index("--")
    argv, plugin_args = argv[:i], argv[i + 1:]
ap = argparse.ArgumentParser()
ap.add_argument("input")
ap.add_argument("-o", "--log-file", default=None)
ap.add_argument("-fl", "--frame-limit", type=int, default=0)
a = ap.parse_args(argv)

r = harness.run_mini_harness(a.input, plugin_args, frame_limit=a.frame_limit, timeout=24 * 3600)
if r.returncode != 0:
    sys.stderr.write(r.stderr)
    sys.exit(r.returncode)
props, cols, rows = harness.parse_log(r.stdout)
acc = None
if a.input.endswith(".slam"):
    _, _, _, gt, gt_ts = slamfile.read_depth_sequence(a.input)
    if len(gt):
        est = np.tile(np.eye(4), (rows.shape[0], 1, 1))
        est[:, :3, 3] = rows[:, -3:]
        acc = frame_metrics(est, gt, rows[:, 1], gt_ts)
n_phase = len(cols) - 6                      # frame, timestamp, Duration_Frame, phases..., X, Y, Z
text = harness.format_log(props, rows[:, 1], rows[:, -3:], rows[:, 2:3 + n_phase], [c[9:] for c in cols[3:3 + n_phase]], acc)
if a.log_file:
    with open(a.log_file, "w") as f:
        f.write(text)
else:
    sys.stdout.write(text)
