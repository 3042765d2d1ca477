cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cfg in "16 1" "32 1" "64 1" "16 4" "64 4" "128 4"; do
  set -- $cfg
  echo "== zchunk $1 group $2" >> gpurun_out/dense_sweep.txt
  KFH_INTEGRATE_ZCHUNK=$1 KFH_INTEGRATE_GROUP=$2 timeout -k 10 90 rocprofv3 --kernel-trace --stats -d gpurun_out/dense_$1_$2 -o d -- python3 tools/dense_integrate.py 512 10 >> gpurun_out/dense_sweep.txt 2>&1 || exit 1
  python3 tools/prof_summary.py gpurun_out/dense_$1_$2 >> gpurun_out/dense_sweep.txt 2>&1 || true
done
