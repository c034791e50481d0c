#!/bin/bash
# usage (GPU box): tools/ab_stress.sh <reps> <variant> ...  -- config 5 (bench.py --mode stress) of A/B builds
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  MVRT_LIB=$PWD/build/ab/libmvrt_$v.so python3 bench.py --mode stress --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'descents/ray', d['config']['descents_per_ray'], 'hits', d['config']['hits'])"
done; done
