#!/bin/bash
# usage (GPU box): tools/ab8.sh <reps> <variant> ...  -- config 5 (bench.py --mode stress), interleaved repetitions
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  echo "$v $(python3 bench.py --mode stress --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done
