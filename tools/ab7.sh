#!/bin/bash
# usage (GPU box): tools/ab7.sh <reps> <variant> ...  -- serial-mode shade / traversal kernel ms and the default pipelined frame, dragon + cave, interleaved repetitions
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  for sc in dragon cave; do
    python3 bench.py --scene $sc --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$v $sc', d['value'], d['ms_per_step'], 'serial: trace share', r['trace_share_of_kernel_time'], 'shade share', r['shade_share_of_kernel_time'], 'sum kernel ms', r['sum_kernel_ms'])"
  done
done; done
