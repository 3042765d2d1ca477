#!/bin/bash
# How long an ICP iteration launch of the coarsest pyramid level (160 x 120) takes when its pixels are spread over
# fewer, larger workgroups: the default (120 blocks of 3 pixel wavefronts) against 30, 12 and 6 blocks.
# usage (GPU box, repository root): bash tools/icp_shape_experiment.sh OUTDIR
set -e
OUT=$1
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in default 8,1 8,3 8,5 8,8; do
    if [ "$shape" = default ]; then unset KFH_TRACK_SHAPE_160; else export KFH_TRACK_SHAPE_160=$shape; fi
    tag=$(echo $shape | tr , x)
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/shape_$tag -o kt -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-host-frames --no-dense-leg > $R/$OUT/shape_$tag.json 2> $R/$OUT/shape_$tag.err
    find $R/$OUT/shape_$tag -name "*kernel_trace.csv" -delete
    python3 - "$R/$OUT/shape_$tag" "$shape" "$R/$OUT/shape_$tag.json" <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
fps = json.load(open(sys.argv[3]))["value"]
for r in csv.DictReader(open(f)):
    if "k_icp_iteration" in r["Name"]:
        print("%-8s %-34s calls %5s  avg %.2f us" % (sys.argv[2], r["Name"].split("(")[0][-34:], r["Calls"], float(r["AverageNs"]) / 1e3))
print("%-8s frames/s under the profiler: %.0f" % (sys.argv[2], fps))
PY
done
