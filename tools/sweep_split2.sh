#!/bin/bash
# usage (GPU box): tools/sweep_split2.sh [bench args]  -- sibling-pass ways x pipeline depth on one rank's share of an 8-way tile split (and the full frame)
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for depth in 3 4; do for ways in 2 3 4; do
  export MVRT_PIPELINE_DEPTH=$depth MVRT_SPLIT_WAYS=$ways
  a=$(python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 "$@" 2>/dev/null | get)
  b=$(python3 bench.py --no-cpu-baseline --no-serial-pass "$@" 2>/dev/null | get)
  echo "depth=$depth ways=$ways | tile 1/8: $a | full: $b"
done; done
