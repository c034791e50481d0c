#!/bin/bash
for tiles in 8 0; do
for cfg in "4 1 1" "2 2 2" "2 2 1" "1 3 2" "1 3 3" "1 4 4" "2 3 2"; do
  set -- $cfg
  r=$(MVRT_BATCH_STEPS=$1 MVRT_PIPELINE_DEPTH=$2 MVRT_TRACE_GRID_DIV=$3 python3 bench.py --no-cpu-baseline --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "tiles=$tiles batch=$1 depth=$2 griddiv=$3 -> $r"
done
done
