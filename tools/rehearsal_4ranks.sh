#!/bin/bash
# Four ranks on ONE GPU over gloo: the launcher path of `bench.py --gpus 4` in both modes, the z-slab mode on the
# 2048^3 volume of the 8-GPU configuration (four slabs of 512 planes).  A rehearsal of the path, not of its speed:
# the ranks share one device and the collectives go through host memory.
# usage (GPU box, repository root): bash tools/rehearsal_4ranks.sh OUTDIR
set -e
OUT=${1:-gpurun_out/rehearsal4}
mkdir -p $OUT
timeout -k 10 280 python3 bench.py --gpus 4 --backend gloo --steps 60 --warmup 10 --no-cpu-baseline > $OUT/replicas_512.json 2> $OUT/replicas_512.err
timeout -k 10 280 python3 bench.py --gpus 4 --backend gloo --mode zslab --steps 30 --warmup 6 --no-cpu-baseline > $OUT/zslab_512.json 2> $OUT/zslab_512.err
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --mode zslab --volume 2048 --steps 20 --warmup 6 --no-cpu-baseline > $OUT/zslab_2048.json 2> $OUT/zslab_2048.err
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --mode zslab --icp allreduce --volume 2048 --steps 20 --warmup 6 --no-cpu-baseline > $OUT/zslab_2048_allreduce.json 2> $OUT/zslab_2048_allreduce.err
python3 - <<PY
import json
for n in ("replicas_512", "zslab_512", "zslab_2048", "zslab_2048_allreduce"):
    j = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    print(n, "n_gpus", j["n_gpus"], "frames/s", round(j["value"], 1), "ate", j.get("ate_rmse_m"), "tracked", j.get("tracked_fraction"))
PY
