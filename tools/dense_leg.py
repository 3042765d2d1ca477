"""The dense integration leg of bench.py on its own (a volume whose every voxel is updated: 8 B x V^3 really moved),
for rocprofv3 passes: kernel trace, and FETCH_SIZE / WRITE_SIZE / SQ counters of k_integrate on a known byte count.

usage: dense_leg.py [V] [band]     prints the leg's roofline record as one JSON line ("band": the volume inside the
truncation band of a wall instead of in free space)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS  # noqa: E402
from slambench_amd.kfusion import KFusion  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 512
BAND = len(sys.argv) > 2 and sys.argv[2] == "band"
print(json.dumps(bench.dense_integrate_leg(KFusion, KFusionConfig, TUM_FR1_DEPTH_INTRINSICS, V, torch, band=BAND)))
