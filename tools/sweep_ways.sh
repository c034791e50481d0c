#!/bin/bash
# usage (GPU box): tools/sweep_ways.sh  -- 1/8 tile share (dragon, rtcamp): sibling passes x traversal-grid divisor x pipeline depth x slot-stream priorities (MVRT_EXPERIMENT build "exp")
cd ${GRAFT_REPO_ROOT:?run through gpurun}
export MVRT_LIB=$PWD/build/ab/libmvrt_exp.so
for cfg in "2 1 3 0" "2 1 3 1" "2 1 3 2" "3 1 3 1" "3 2 3 1" "4 1 4 1" "4 2 4 1" "4 2 4 2"; do set -- $cfg
  for sc in dragon rtcamp; do
  MVRT_SPLIT_WAYS=$1 MVRT_TRACE_GRID_DIV=$2 MVRT_PIPELINE_DEPTH=$3 MVRT_SLOT_PRIO=$4 python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --steps 8 --warmup 4 --emulate-tiles 8 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1])
print('ways=$1 div=$2 depth=$3 prio=$4 $sc', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"
done; done
