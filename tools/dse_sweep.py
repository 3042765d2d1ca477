"""Sweep plugin parameters over a dataset, one process per GPU.

  python tools/dse_sweep.py stream.slam --gpus 0,1,2,3 --grid volume-resolution=128,256,512 mu=0.05,0.1 [-fl 100]

Prints one tab-separated row per point: the parameters, frames, mean frame time, ATE (when the file holds ground
truth) and the GPU that ran it.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import dse, slamfile  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("input")
ap.add_argument("--gpus", default="0")
ap.add_argument("--grid", nargs="+", required=True, metavar="option=v1,v2,...")
ap.add_argument("--per-gpu", type=int, default=1,
                help="processes sharing one GPU at a time (the pipeline is latency-bound: 3-4 sequences fit one MI355X)")
ap.add_argument("-fl", "--frame-limit", type=int, default=0)
a = ap.parse_args()
grid = {}
for item in a.grid:
    k, v = item.split("=", 1)
    grid[k] = v.split(";") if ";" in v else v.split(",")     # ';' separates values that contain commas
gt = gt_ts = None
if a.input.endswith(".slam"):
    _, _, _, gt, gt_ts = slamfile.read_depth_sequence(a.input)
points = dse.grid_points(grid)
results = dse.run_sweep(points, dse.mini_harness_cmd(a.input, frame_limit=a.frame_limit),
                        [int(g) for g in a.gpus.split(",")] * max(1, a.per_gpu), summarise_fn=lambda t: dse.summarise(t, gt, gt_ts))
keys = list(grid)
print("\t".join(keys + ["frames", "mean_frame_s", "ate_rmse_m", "gpu", "status"]))
for r in results:
    s = r.summary
    print("\t".join([str(r.point[k]) for k in keys] +
                    [str(s.get("frames", "")), "%.6f" % s["mean_frame_s"] if "mean_frame_s" in s else "",
                     "%.6f" % s["ate_rmse_m"] if "ate_rmse_m" in s else "", str(r.gpu),
                     "ok" if r.returncode == 0 else "failed(%d)" % r.returncode]))
