"""Raw TUM RGB-D or ICL-NUIM folder -> .slam file (ground truth + depth), the job of the reference's
dataset-generator for a depth-only plugin.

  python tools/make_slam.py tum  /data/rgbd_dataset_freiburg1_xyz  fr1_xyz.slam
  python tools/make_slam.py icl  /data/living_room_traj2_loop      lr_kt2.slam [--positive-focal] [--frames N]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slambench_amd import datasets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("layout", choices=["tum", "icl"])
ap.add_argument("folder")
ap.add_argument("output")
ap.add_argument("--frames", type=int, default=0, help="stop after this many depth frames (0 = all)")
ap.add_argument("--positive-focal", action="store_true", help="ICL-NUIM: the reference's -pf switch")
ap.add_argument("--disparity-scale", type=float, default=None,
                help="metres per raw unit to write into the depth sensor block (default: what the data means)")
a = ap.parse_args()
seq = datasets.load_tum(a.folder, a.frames) if a.layout == "tum" else \
    datasets.load_icl_nuim(a.folder, a.positive_focal, a.frames)
datasets.to_slam(seq, a.output, a.disparity_scale)
print("%s: %d depth frames, %d ground-truth poses, intrinsics %s, %g m per unit" %
      (a.output, seq.depth.shape[0], len(seq.gt_timestamps), seq.intrinsics_px,
       seq.disparity_scale if a.disparity_scale is None else a.disparity_scale))
