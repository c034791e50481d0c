#!/bin/bash
# usage (GPU box): tools/ab2.sh <variant|default> ...   -- default frame (pipelined) and serial mode of A/B builds (build/ab/libmvrt_<variant>.so)
cd ${GRAFT_REPO_ROOT:?run through gpurun}
get() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('trace_kernel_mrays_per_s'), r.get('serial_pass_wall_ms'))"; }
for v in "$@"; do
  if [ "$v" = default ]; then unset MVRT_LIB; else export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so; fi
  a=$(python3 bench.py --no-cpu-baseline --no-serial-pass 2>/dev/null | get)
  b=$(python3 bench.py --no-cpu-baseline --serial-only --warmup 1 2>/dev/null | get)
  echo "$v | default: $a | serial: $b"
done
