#!/bin/bash
# usage (GPU box): tools/share_ab.sh <reps> <variant|-> ... -- 64-spp frames of the dragon and rtcamp stand-ins: full frame and 1/8 tile share (A/B builds via MVRT_LIB; "-" = the shipped library)
cd ${GRAFT_REPO_ROOT:?run through gpurun}
reps=$1; shift
for r in $(seq $reps); do for v in "$@"; do for sc in dragon rtcamp; do for t in 0 8; do
  [ "$v" != "-" ] && export MVRT_LIB=$PWD/build/ab/libmvrt_$v.so
  python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --steps 8 --warmup 4 $([ $t != 0 ] && echo --emulate-tiles $t) $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1])
print('$v $sc tiles=$t', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"
done; done; done; done
