#!/bin/bash
# usage (on the GPU box, repo root): tools/gpu_check.sh <tag> [pytest args]  -- GPU tests, default bench line, serial-mode kernel stats into gpurun_out/<tag>/
set -o pipefail
tag=${1:?tag}; shift
root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q "$@" 2>&1 | tee $out/pytest_gpu.log | tail -15 &&
timeout -k 10 300 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err && tail -c 1500 $out/bench_default.json &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serial -- python3 bench.py --serial-only --no-cpu-baseline --warmup 0 > $out/stats_serial.log 2>&1 &&
tail -1 $out/stats_serial.log | tail -c 1500 &&
python3 tools/condense_rocprof.py $out/stats_serial $out/serial_kernel_stats.csv | head -14
