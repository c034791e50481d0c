#!/bin/bash
# usage (GPU box): tools/refresh_stress.sh <tag>  -- the config-5 entries of tools/final_profiles.sh only (bench line + kernel stats) into gpurun_out/final_<tag>/
set -o pipefail
tag=${1:-r02}; root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/final_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
python3 bench.py --mode stress --steps 3 --warmup 1 > $out/bench_stress.json 2>/dev/null; echo "stress rc=$?"
rm -rf $out/stats_stress
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_stress -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/stats_stress.log 2>&1; echo "stress stats rc=$?"
tail -c 600 $out/bench_stress.json
