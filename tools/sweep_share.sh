#!/bin/bash
# usage (GPU box): tools/sweep_share.sh  -- scheduling knobs on one rank's share of an 8-way tile split (dragon, rtcamp)
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
run() { echo "$1 | dragon $(env $2 python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get) | rtcamp $(env $2 python3 bench.py --scene rtcamp --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get)"; }
run "default" "X=1"
run "div only stages<=0" "MVRT_TRACE_DIV_MAX_STAGE=0"
run "div only stages<=1" "MVRT_TRACE_DIV_MAX_STAGE=1"
run "div only stages<=2" "MVRT_TRACE_DIV_MAX_STAGE=2"
run "no split, batch 4" "MVRT_SPLIT_SMALL=0"
run "waves/CU 20" "MVRT_TRACE_WAVES_PER_CU=20"
run "waves/CU 12" "MVRT_TRACE_WAVES_PER_CU=12"
run "small rpl 8 minw 2048" "MVRT_SMALL_RPL=8 MVRT_SMALL_MINW=2048"
