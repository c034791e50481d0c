#!/bin/bash
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for w in 32 24 20 16 12; do
  export MVRT_TRACE_WAVES_PER_CU=$w
  echo "wpc=$w cave: $(python3 bench.py --scene cave --no-cpu-baseline --no-serial-pass 2>/dev/null | get) | dragon: $(python3 bench.py --no-cpu-baseline --no-serial-pass 2>/dev/null | get)"
done
