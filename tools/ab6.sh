#!/bin/bash
# usage (GPU box): tools/ab6.sh <reps> <variant> ...  -- one rank's share of an 8-way tile split (64-spp frames), interleaved repetitions
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  echo "$v $(python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 --steps 16 --warmup 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])")"
done; done
