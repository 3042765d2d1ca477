#!/bin/bash
# Kernel-trace durations of the dense leg for several builds of the core library.
# usage (GPU box, repository root): bash tools/dense_variants.sh OUTDIR name1 name2 ...   ("" = the default build)
OUT=$1; shift
for v in "$@"; do
  if [ "$v" = "default" ]; then unset KFH_CORE_LIB; else export KFH_CORE_LIB=$PWD/slambench_amd/lib/libkfusion_hip_core_$v.so; fi
  echo "== $v"
  bash tools/dense_trace.sh $OUT/$v | cut -c1-200
done
