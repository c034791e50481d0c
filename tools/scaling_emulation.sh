#!/bin/bash
# usage: tools/scaling_emulation.sh [steps]  -- per-rank time of an N-way tile split, emulated on one GPU (default library configuration)
steps=${1:-4}
for tiles in 0 2 4 8; do
  r=$(python3 bench.py --no-cpu-baseline --steps $steps --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "steps=$steps tiles=$tiles -> $r"
done
