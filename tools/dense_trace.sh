#!/bin/bash
# Kernel trace of the dense integration leg (tools/dense_leg.py): per-launch durations of the integrate kernels.
# usage (GPU box, repository root): bash tools/dense_trace.sh OUTDIR [V]
set -e
OUT=$1
V=${2:-512}
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -o kt -- python3 $R/tools/dense_leg.py $V > $R/$OUT/dense_kt.json 2> $R/$OUT/dense_kt.err
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][-32:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    if "integrate" in k:
        print(k, len(v), "launches; us:", " ".join("%.1f" % x for x in v))
PY
find $OUT -name "*kernel_trace.csv" -delete
