#!/bin/bash
# usage (GPU box): tools/sweep_small.sh "<rpl list>" "<minw list>"  -- the small-launch wave limit (MVRT_SMALL_RPL / MVRT_SMALL_MINW): full frame, serial mode, 1/8 tile shares
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"; }
for rpl in $1; do for mw in $2; do
  export MVRT_SMALL_RPL=$rpl MVRT_SMALL_MINW=$mw
  echo "rpl=$rpl minw=$mw | full $(python3 bench.py --no-cpu-baseline --no-serial-pass 2>/dev/null | get) | serial $(python3 bench.py --no-cpu-baseline --serial-only --warmup 1 2>/dev/null | get) | dragon 1/8 $(python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get) | rtcamp 1/8 $(python3 bench.py --scene rtcamp --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get) | cave 1/8 $(python3 bench.py --scene cave --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get)"
done; done
