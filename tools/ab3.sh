#!/bin/bash
# usage (GPU box): tools/ab3.sh <variant> ...   -- serial-mode shade / trace kernel time of A/B builds, dragon and closed cave
cd ${GRAFT_REPO_ROOT:?run through gpurun}
for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  for sc in dragon cave; do
  python3 bench.py --scene $sc --no-cpu-baseline --serial-only --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$v $sc', 'ms/step', d['ms_per_step'], 'trace ms', round(r['avg_launch_ms']*r['launches'],2), 'shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2))"
  done
done
