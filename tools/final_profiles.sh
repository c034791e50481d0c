#!/bin/bash
# usage (on the GPU box, repo root): tools/final_profiles.sh <tag>   -- everything the round's profiles/ entries are made from, into gpurun_out/final_<tag>/
set -o pipefail
tag=${1:-r01b}; out=gpurun_out/final_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run through gpurun}
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/stats.log 2>&1; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --no-cpu-baseline > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python3 bench.py --mode primary --grid-res 1024 --no-cpu-baseline > $out/bench_primary_1024.json 2>/dev/null; echo "primary rc=$?"
python3 bench.py --scene rtcamp --grid-res 4096 --steps 16 --no-cpu-baseline > $out/bench_rtcamp_4096_256spp.json 2>/dev/null; echo "rtcamp rc=$?"
python3 bench.py --mode stress --grid-res 8192 --voxels 6.5e8 --rays 1.6e7 --steps 3 --warmup 1 > $out/bench_stress.json 2>/dev/null; echo "stress rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_stress -- python3 bench.py --mode stress --grid-res 8192 --voxels 6.5e8 --rays 1.6e7 --steps 3 --warmup 1 > $out/stats_stress.log 2>&1; echo "stress stats rc=$?"
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-tiles $n 2>/dev/null | tail -1 > $out/bench_emulate_tiles_$n.json; done
python3 bench.py --no-cpu-baseline --emulate-tiles 8 --steps 16 2>/dev/null | tail -1 > $out/bench_emulate_tiles_8_steps16.json
python3 bench.py --no-cpu-baseline --steps 16 2>/dev/null | tail -1 > $out/bench_steps16.json
tail -c 400 $out/bench_default.json
