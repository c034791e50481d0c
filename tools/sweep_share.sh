#!/bin/bash
# usage (GPU box): tools/sweep_share.sh  -- scheduling knobs on one rank's share of an 8-way tile split (dragon, rtcamp)
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
run() { echo "$1 | dragon $(env $2 python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get) | rtcamp $(env $2 python3 bench.py --scene rtcamp --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get)"; }
run "default" "X=1"
run "no split (one pass of 4 steps)" "MVRT_SPLIT_SMALL=0"
run "3 sibling passes" "MVRT_SPLIT_WAYS=3"
run "4 sibling passes, depth 4" "MVRT_SPLIT_WAYS=4 MVRT_PIPELINE_DEPTH=4"
run "2 siblings, full grids" "MVRT_TRACE_GRID_DIV=1"
run "minw 1536" "MVRT_SMALL_MINW=1536"
run "minw 3072" "MVRT_SMALL_MINW=3072"
run "rpl 32" "MVRT_SMALL_RPL=32"
