#!/bin/bash
# Eight parameter points of a design-space sweep on ONE GPU (four mini-harness processes at a time), the shape of the
# 8-way sweep that runs one point per GPU on a full node: a 120-frame synthetic .slam file, volume resolution x mu x
# ICP threshold (the three axes the 8-way configuration names).  usage (GPU box, repository root): bash tools/rehearsal_sweep8.sh OUTDIR
set -e
OUT=${1:-gpurun_out/sweep8}
mkdir -p $OUT
python3 - <<PY
import sys
sys.path.insert(0, ".")
from slambench_amd import slamfile, synth
from slambench_amd.config import TUM_FR1_DEPTH_INTRINSICS
depth, gt, ts = synth.make_sequence(120, seed=1, intrinsics=TUM_FR1_DEPTH_INTRINSICS, noise=True)
slamfile.write_depth_sequence("/tmp/sweep8.slam", depth, ts, TUM_FR1_DEPTH_INTRINSICS, gt_poses=gt, with_rgb_and_grey=False)
PY
timeout -k 10 600 python3 tools/dse_sweep.py /tmp/sweep8.slam --gpus 0 --per-gpu 4 \
    --grid "volume-resolution=256,256,256;512,512,512" mu=0.05,0.1 icp-threshold=1e-5,1e-4 > $OUT/sweep8.tsv 2> $OUT/sweep8.err
cat $OUT/sweep8.tsv
