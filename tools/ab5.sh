#!/bin/bash
# usage (GPU box): tools/ab5.sh <reps> <variant> ...  -- default (pipelined) frame, 20 steps, interleaved repetitions
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  echo "$v $(python3 bench.py --no-cpu-baseline --no-serial-pass --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done
