#!/bin/bash
# SQ instruction counters of the dense integration leg (tools/dense_leg.py) under rocprofv3; prints the mean of
# the last launches of k_integrate_plan / k_integrate.  usage (GPU box, repository root): bash tools/dense_counters.sh OUTDIR [V]
set -e
OUT=$1
V=${2:-512}
R=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/$OUT/sq -o sq -- python3 $R/tools/dense_leg.py $V > $R/$OUT/dense_sq.json 2> $R/$OUT/dense_sq.err
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "integrate" in k:
        print(k, {c: round(sum(v[-6:]) / len(v[-6:])) for c, v in cs.items()}, len(list(cs.values())[0]))
PY
cat $OUT/dense_sq.json | cut -c1-400
