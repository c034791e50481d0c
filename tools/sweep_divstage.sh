#!/bin/bash
for tiles in 8 4; do
for st in 99 0 1 2 3 4; do
  r=$(MVRT_TRACE_DIV_MAX_STAGE=$st python3 bench.py --no-cpu-baseline --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "tiles=$tiles div_max_stage=$st -> $r"
done; done
