#!/bin/bash
# usage (GPU box): tools/ab_serial.sh <reps> <variant> ...  -- serial-mode traversal / shade kernel ms (dragon) + the default pipelined frame, interleaved repetitions
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$v', d['value'], d['ms_per_step'], 'serial: trace ms', round(r['avg_launch_ms']*r['launches'],2), 'shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'trace Mrays/s', r['trace_kernel_mrays_per_s'])"
done; done
