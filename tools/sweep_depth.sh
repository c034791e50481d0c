#!/bin/bash
# usage (GPU box): tools/sweep_depth.sh -- full-frame headline: pipeline depth x merged steps x slot-stream priorities (MVRT_EXPERIMENT build "exp")
cd ${GRAFT_REPO_ROOT:?run through gpurun}
export MVRT_LIB=$PWD/build/ab/libmvrt_exp.so
for cfg in "3 0 0" "3 0 1" "4 0 1" "4 1 1" "3 1 0" "2 0 0" "4 2 1"; do set -- $cfg
  MVRT_PIPELINE_DEPTH=$1 MVRT_BATCH_STEPS=$2 MVRT_SLOT_PRIO=$3 python3 bench.py --no-cpu-baseline --no-serial-pass --steps 20 --warmup 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1])
print('depth=$1 batch=$2 prio=$3', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"
done
