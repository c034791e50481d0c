#!/bin/bash
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for cfg in "3 2" "1 1" "1 2" "1 4" "3 1" "2 2"; do set -- $cfg
  export MVRT_PIPELINE_DEPTH=$1 MVRT_BATCH_STEPS=$2
  echo "cave depth=$1 batch=$2: $(python3 bench.py --scene cave --no-cpu-baseline --no-serial-pass 2>/dev/null | get)"
done
