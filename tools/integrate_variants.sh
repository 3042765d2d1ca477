#!/bin/bash
# Dense leg and headline integrate time of several builds of the core library (make VARIANT=...).
# usage (GPU box, repository root): bash tools/integrate_variants.sh name1 name2 ...   ("" = the real build)
for v in "$@"; do
    if [ "$v" = "main" ]; then unset KFH_CORE_LIB; else export KFH_CORE_LIB=$PWD/slambench_amd/lib/libkfusion_hip_core_$v.so; fi
    timeout -k 10 100 python3 tools/dense_leg.py > /tmp/dense_$v.json || exit 1
    timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-host-frames --no-dense-leg > /tmp/bench_$v.json || exit 1
    python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("/tmp/dense_%s.json" % v)); b = json.load(open("/tmp/bench_%s.json" % v))
print("%-8s dense %.1f us (frac %.3f) | headline %.0f frames/s, integrate %.1f us, frac %.3f" % (
    v, d["ms_per_launch"] * 1e3, d["frac"], b["value"], b["integrate_ms"] * 1e3, b["roofline"]["frac"]))
PY
done
