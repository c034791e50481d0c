#!/bin/bash
# usage (GPU box): tools/sweep_wpc.sh "<waves-per-CU list>"  -- full-grid traversal launches with fewer waves per CU (room for the shade waves of the other pipeline streams): pipelined frame, dragon + cave
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for w in $1; do
  if [ $w = 0 ]; then unset MVRT_TRACE_WAVES_PER_CU; else export MVRT_TRACE_WAVES_PER_CU=$w; fi
  echo "wpc=$w | dragon $(python3 bench.py --no-cpu-baseline --no-serial-pass --steps 8 --warmup 2 2>/dev/null | get) | cave $(python3 bench.py --scene cave --no-cpu-baseline --no-serial-pass --steps 8 --warmup 2 2>/dev/null | get)"
done; done
