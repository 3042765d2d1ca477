#!/bin/bash
# rocprofv3 passes behind the files of profiles/: one kernel trace of the default benchmark and two counter
# passes (FETCH_SIZE and WRITE_SIZE each in a run of their own), the same three passes for the dense integration
# leg (tools/dense_leg.py: a volume whose every voxel is updated, so the byte count is known), written under
# gpurun_out/prof_TAG/.
# usage (on the GPU box, from the repository root): bash tools/profile_round.sh TAG
# then, where gpurun_out/ has been merged back:
#   python tools/make_profiles.py TAG 0 gpurun_out/prof_TAG/kt gpurun_out/prof_TAG/fetch gpurun_out/prof_TAG/write
#   python tools/make_profiles.py --dense TAG gpurun_out/prof_TAG
set -e
TAG=${1:-round}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $R/bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-dense-leg > $OUT/bench_under_trace.json 2> $OUT/kt.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dense-leg --no-host-frames > /dev/null 2> $OUT/fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dense-leg --no-host-frames > /dev/null 2> $OUT/write.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dense_kt -o kt -- python3 $R/tools/dense_leg.py > $OUT/dense_under_trace.json 2> $OUT/dense_kt.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/dense_fetch -o fetch -- python3 $R/tools/dense_leg.py > /dev/null 2> $OUT/dense_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/dense_write -o write -- python3 $R/tools/dense_leg.py > /dev/null 2> $OUT/dense_write.err
cd $R
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python3 tools/launch_gaps.py $OUT/kt > $OUT/integrate_launch_gaps.txt
# the raw traces are large: keep the statistics and the counter tables
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -60
