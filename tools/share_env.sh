#!/bin/bash
# usage (GPU box): tools/share_env.sh <reps> <variant>[:ENV=V[,ENV=V...]] ...  -- like share_ab.sh (dragon / rtcamp, full frame and 1/8 tile share), each variant with its own environment knobs
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for spec in "$@"; do for sc in dragon rtcamp; do for t in 0 8; do
  v=${spec%%:*}; e=""; [[ $spec == *:* ]] && e=$(echo ${spec#*:} | tr ',' ' ')
  env MVRT_LIB=$PWD/build/ab/libmvrt_$v.so $e python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --steps 8 --warmup 4 $([ $t != 0 ] && echo --emulate-tiles $t) $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1])
print('$spec $sc tiles=$t', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"
done; done; done; done
