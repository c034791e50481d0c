#!/bin/bash
# usage (GPU box): tools/sweep_div.sh  -- sibling passes with full vs divided traversal grids, on one rank's share of an 8-way split
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for sc in dragon rtcamp; do
for cfg in "2 0 3" "2 1 3" "4 1 4" "4 2 4" "3 1 3" "2 1 99:0"; do set -- $cfg
  export MVRT_SPLIT_WAYS=$1 MVRT_TRACE_GRID_DIV=$2 MVRT_PIPELINE_DEPTH=$3
  [ "$2" = 0 ] && unset MVRT_TRACE_GRID_DIV
  echo "$sc ways=$1 div=$2 depth=$3: $(python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get)"
done; done
