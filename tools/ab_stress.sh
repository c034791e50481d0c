#!/bin/bash
# usage: tools/ab_stress.sh <libname|default> ...  -- config-5 stress throughput of A/B builds
for v in "$@"; do
  if [ "$v" = default ]; then unset MVRT_LIB; else export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so; fi
  r=$(python3 bench.py --mode stress --grid-res 8192 --voxels 6.5e8 --rays 1.6e7 --steps 2 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['line_gbs'])")
  echo "$v stress -> $r"
done
