#!/bin/bash
# usage: tools/sweep_pipeline.sh  -- job throughput vs (batch steps per pass, pipeline depth) for the full frame and for one rank's share of an 8-way tile split
for tiles in 0 8; do
for cfg in "0 3" "1 4" "1 3" "2 2" "2 3" "4 1"; do
  set -- $cfg
  r=$(MVRT_BATCH_STEPS=$1 MVRT_PIPELINE_DEPTH=$2 python3 bench.py --no-cpu-baseline --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "tiles=$tiles batch=$1 depth=$2 -> $r"
done
done
