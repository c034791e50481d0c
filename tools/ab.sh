#!/bin/bash
# usage: tools/ab.sh <libname|default> ...  -- job throughput of A/B builds (build/ab/libmvrt_<name>.so): full frame, and one rank's share of an 8-way split
for v in "$@"; do
  if [ "$v" = default ]; then unset MVRT_LIB; else export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so; fi
  for tiles in 0 8; do
    r=$(python3 bench.py --no-cpu-baseline --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['trace_kernel_mrays_per_s'])")
    echo "$v tiles=$tiles -> $r"
  done
done
