#!/bin/bash
# frames/s of the default benchmark with other workgroup shapes (pixel wavefronts, rows per thread) for the ICP
# iteration launches of one pyramid level.  usage: bash tools/icp_shape_sweep.sh WIDTH "pw,ppt pw,ppt ..."
W=$1
for shape in default $2; do
    if [ "$shape" = default ]; then unset KFH_TRACK_SHAPE_$W; else export KFH_TRACK_SHAPE_$W=$shape; fi
    for rep in 1 2; do
        timeout -k 10 100 python3 bench.py --steps 600 --no-cpu-baseline --no-host-frames --no-dense-leg 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$W $shape', round(r['value']), round(r['ms_per_step']*1e3,2))" || exit 1
    done
done
