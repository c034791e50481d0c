#!/bin/bash
# usage (GPU box): tools/ab_hints.sh [bench args]  -- serial-mode kernel times with / without start-below-the-root, then the MVRT_UTIL_STATS build's visit tallies
cd ${GRAFT_REPO_ROOT:?run through gpurun}
for h in "" "--no-hints"; do
  python3 bench.py --no-cpu-baseline --serial-only --warmup 1 $h "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('hints' if '$h'=='' else 'root ', 'ms/step', d['ms_per_step'], 'trace ms', round(r['avg_launch_ms']*r['launches'],2), 'shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'trace Mrays/s', r['trace_kernel_mrays_per_s'])"
done
if [ -f build/ab/libmvrt_util.so ]; then
for h in "" "--no-hints"; do
  echo "util build $h"; MVRT_LIB=$PWD/build/ab/libmvrt_util.so MVRT_PRINT_UTIL=1 python3 bench.py --no-cpu-baseline --serial-only --warmup 0 --steps 1 $h "$@" 2>&1 >/dev/null | grep "\[util\]" | head -12
done
fi
