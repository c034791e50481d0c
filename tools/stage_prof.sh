#!/bin/bash
# usage (GPU box): tools/stage_prof.sh <tag> <variant|-> [bench args]  -- kernel trace of a serial-mode run, per-stage traversal / shade durations
cd ${GRAFT_REPO_ROOT:?run through gpurun}
tag=$1; v=$2; shift 2
[ "$v" != "-" ] && export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --serial-only --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
python3 tools/stage_times.py $out
