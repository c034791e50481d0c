#!/bin/bash
for steps in 4 16; do
for tiles in 4 8; do
for mx in 0 20000000 40000000 80000000; do
  r=$(MVRT_SPLIT_SMALL_MAX=$mx python3 bench.py --no-cpu-baseline --steps $steps --emulate-tiles $tiles 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "steps=$steps tiles=$tiles splitmax=$mx -> $r"
done; done; done
