#!/bin/bash
# usage (GPU box): tools/ab4.sh <variant> ...   -- default frame, serial mode and a 1/8 tile share (dragon + rtcamp) of A/B builds
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do
  export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  echo "$v | default: $(python3 bench.py --no-cpu-baseline --no-serial-pass 2>/dev/null | get) | serial: $(python3 bench.py --no-cpu-baseline --serial-only --warmup 1 2>/dev/null | get) | dragon 1/8: $(python3 bench.py --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get) | rtcamp 1/8: $(python3 bench.py --scene rtcamp --no-cpu-baseline --no-serial-pass --emulate-tiles 8 2>/dev/null | get)"
done
