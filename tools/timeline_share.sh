#!/bin/bash
# usage (GPU box): tools/timeline_share.sh <tag> [bench args]  -- kernel timeline (all streams) of a pipelined run, e.g. a 1/8 tile share
cd ${GRAFT_REPO_ROOT:?run through gpurun}
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-serial-pass "$@" > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log | head -c 600; echo
python3 tools/timeline.py $out 400 > $out/timeline.txt
tail -130 $out/timeline.txt
