#!/bin/bash
# usage (GPU box): tools/sweep_small.sh "<rpl list>" "<minw list>"  -- the small-launch wave limit (MVRT_SMALL_RPL / MVRT_SMALL_MINW) on the default frame, serial mode and an 8-way tile share
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('serial_pass_wall_ms'))"; }
for rpl in $1; do for mw in $2; do
  export MVRT_SMALL_RPL=$rpl MVRT_SMALL_MINW=$mw
  a=$(python3 bench.py --no-cpu-baseline --no-serial-pass 2>/dev/null | get)
  b=$(python3 bench.py --no-cpu-baseline --serial-only --warmup 1 2>/dev/null | get)
  c=$(python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get)
  echo "rpl=$rpl minw=$mw | default: $a | serial: $b | tile 1/8: $c"
done; done
