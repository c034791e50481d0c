#!/bin/bash
for tiles in 0 8; do
for w in 0 28 24 20 16; do
  r=$(MVRT_TRACE_WAVES_PER_CU=$w python3 bench.py --no-cpu-baseline --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['trace_kernel_mrays_per_s'])")
  echo "tiles=$tiles waves_per_cu=$w -> $r"
done; done
