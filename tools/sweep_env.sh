#!/bin/bash
# usage (GPU box): tools/sweep_env.sh VAR "v1 v2 ..." [bench args]  -- serial-mode kernel times for each value of an environment knob
cd ${GRAFT_REPO_ROOT:?run through gpurun}
var=$1; vals=$2; shift 2
for v in $vals; do
  export $var=$v
  python3 bench.py --no-cpu-baseline --serial-only --warmup 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$var=$v', 'serial ms/step', d['ms_per_step'], 'trace ms', round(r['avg_launch_ms']*r['launches'],2), 'shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'sum', r['sum_kernel_ms'])"
done
