#!/bin/bash
# The benchmark stream at 1024^3 and 2048^3 (same scene, same camera path), each once plainly and the larger once
# under a kernel trace; results under gpurun_out/big_TAG/.
# usage (GPU box, repository root): bash tools/big_volume_profile.sh TAG
set -e
TAG=${1:-round}
R=$PWD
OUT=$R/gpurun_out/big_$TAG
mkdir -p $OUT
for V in 1024 2048; do
  timeout -k 10 280 python3 bench.py --volume $V --steps 100 --warmup 10 --no-cpu-baseline --no-dense-leg --no-host-frames > $OUT/bench_$V.json 2> $OUT/bench_$V.err
done
cd /tmp
export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt2048 -o kt -- python3 $R/bench.py --volume 2048 --steps 100 --warmup 10 --no-cpu-baseline --no-dense-leg --no-host-frames > $OUT/bench_2048_under_trace.json 2> $OUT/kt2048.err
find $OUT -name "*kernel_trace.csv" -delete
cd $R
python3 - <<PY
import json
for v in (1024, 2048):
    j = json.loads(open("$OUT/bench_%d.json" % v).read().strip().splitlines()[-1])
    print(v, round(j["value"], 1), "frames/s; integrate", round(j["integrate_ms"] * 1e3, 1), "us;",
          int(j["integrate_voxels_per_launch"]), "voxels per launch; frac", round(j["roofline"]["frac"], 3))
PY
