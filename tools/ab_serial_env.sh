#!/bin/bash
# usage (GPU box): tools/ab_serial_env.sh <reps> <variant>[:ENV=V[,ENV=V...]] ...   -- like ab_serial.sh, each variant with its own environment knobs
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for spec in "$@"; do
  v=${spec%%:*}; e=""; [[ $spec == *:* ]] && e=$(echo ${spec#*:} | tr ',' ' ')
  env MVRT_LIB=$PWD/build/ab/libmvrt_$v.so $e python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$spec', d['value'], d['ms_per_step'], 'serial: trace ms', round(r['avg_launch_ms']*r['launches'],2), 'shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'trace Mrays/s', r['trace_kernel_mrays_per_s'])"
done; done
