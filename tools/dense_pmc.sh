#!/bin/bash
# Counter passes of the dense integration leg (tools/dense_leg.py), each group of counters in a run of its own.
# usage (GPU box, repository root): bash tools/dense_pmc.sh OUTDIR "CTR_A CTR_B" "CTR_C ..." ...
OUT=$1; shift
R=$PWD
mkdir -p $OUT
i=0
for group in "$@"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 90 rocprofv3 --pmc $group --output-format csv -d $R/$OUT/p$i -o p -- python3 $R/tools/dense_leg.py > $R/$OUT/p$i.json 2> $R/$OUT/p$i.err ) || { echo "pass $i failed"; tail -3 $OUT/p$i.err; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "integrate" in k:
        print(k, {c: round(sum(v[-6:]) / len(v[-6:])) for c, v in sorted(cs.items())})
PY
