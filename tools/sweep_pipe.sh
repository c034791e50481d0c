#!/bin/bash
# usage (GPU box): tools/sweep_pipe.sh [bench args]  -- pipeline depth x merged steps per pass on the 64-spp-frame bench
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for depth in 2 3 4; do for batch in 1 2 4; do
  export MVRT_PIPELINE_DEPTH=$depth MVRT_BATCH_STEPS=$batch
  echo "depth=$depth batch=$batch: $(python3 bench.py --no-cpu-baseline --no-serial-pass "$@" 2>/dev/null | get)"
done; done
