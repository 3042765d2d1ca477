#!/bin/bash
# Cache behaviour of the raycast (and the other gathering kernels) on the benchmark stream: two counter passes
# (vector instructions; L1 accesses and L1-to-L2 requests) and the sample statistics of a march launch from the
# diagnostic build.  usage (GPU box, repository root): bash tools/gather_efficiency.sh TAG
# then: python tools/make_profiles.py --gather TAG gpurun_out/gather_TAG
set -e
TAG=${1:-round}
R=$PWD
OUT=$R/gpurun_out/gather_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dense-leg --no-host-frames > /dev/null 2> $OUT/sq.err
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcp -o tcp -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dense-leg --no-host-frames > /dev/null 2> $OUT/tcp.err
cd $R
KFH_CORE_LIB=$R/slambench_amd/lib/libkfusion_hip_core_stamps.so timeout -k 10 120 python3 tools/raycast_stats.py > $OUT/raycast_stats.txt 2> $OUT/raycast_stats.err
cat $OUT/raycast_stats.txt | tail -20
