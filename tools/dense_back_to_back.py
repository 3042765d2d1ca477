"""Dense integrations back to back: the dense volume of bench.py's roofline_dense leg, but instead of sequences of four
frames with a reset between them (each reset rewrites the volume and empties the memory-side cache of it), ONE frame is
preprocessed and the integration stage is queued 24 times in a row (a sequence's first frames integrate
unconditionally, and the frame is not ended in between).  Timed with HIP events on the pipeline's stream around each
stage call (plan + work launch).  usage: dense_back_to_back.py [V=512]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slambench_amd import native  # noqa: E402
from slambench_amd.config import KFusionConfig, TUM_FR1_DEPTH_INTRINSICS  # noqa: E402
from slambench_amd.kfusion import KFusion  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = KFusionConfig(input_size=(640, 480), intrinsics=TUM_FR1_DEPTH_INTRINSICS, volume_resolution=(V, V, V),
                    volume_size=(0.5, 0.5, 0.5), volume_direction=(0.25, 0.25, -1.0))
kf = KFusion(cfg)
wall = torch.full((480, 640), 3000, dtype=torch.int16, device="cuda")
stream = torch.cuda.ExternalStream(kf.stream())
kf.set_depth(int(wall.data_ptr()))
kf.preprocessing()
kf.tracking()
n = 24
events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for e0, e1 in events:
    e0.record(stream)
    kf.integration()
    e1.record(stream)
kf.sync()
ms = np.array([e0.elapsed_time(e1) for e0, e1 in events])
weights = kf.download(native.BUF_VOLUME)[..., 1]
assert int(weights.min()) == n and int(weights.max()) == n, "every voxel must have been updated by every launch"
steady = ms[4:]
print(json.dumps({"volume": V, "launches": n, "us_per_launch": [round(float(x) * 1e3, 1) for x in ms],
                  "median_us_after_the_first_four": float(np.median(steady) * 1e3),
                  "frac_of_8TBs": 8.0 * V ** 3 / (float(np.median(steady)) * 1e-3) / 8e12}))
kf.close()
