#!/bin/bash
# usage (GPU box): tools/ab.sh <serial|share|stress> <reps> <variant>[:ENV=V[,ENV=V...]] ...
#   A/B of tools/build_variant.sh builds (build/ab/libmvrt_<variant>.so; "-" = the shipped library), interleaved repetitions, each variant with its own
#   experiment knobs (mvrtKnob names; only -DMVRT_EXPERIMENT builds read them).  $BENCH_ARGS is appended to every bench.py call (e.g. "--scene cave").
#     serial : serial-mode traversal / shade kernel ms per 4 steps (dragon by default) + the pipelined frame
#     share  : dragon and rtcamp stand-ins, full frame and 1/8 tile share (64-spp frames); $SCENES / $TILES override ("0" = full frame)
#     stress : config 5 (bench.py --mode stress)
# Every sweep of profiles/r0x_experiments.txt (refill thresholds, sibling passes, pipeline depth, shade builds, small-launch limits ...) is a list of arguments to this.
cd ${GRAFT_REPO_ROOT:?run through gpurun}
mode=$1; reps=$2; shift 2
run() { # <spec> <label> <python summary> <bench args...>
  local spec=$1 label=$2 fmt=$3; shift 3
  local v=${spec%%:*} e=""; [[ $spec == *:* ]] && e=$(echo ${spec#*:} | tr ',' ' ')
  local lib=""; [ "$v" != "-" ] && lib="MVRT_LIB=$PWD/build/ab/libmvrt_$v.so"
  env $lib $e python3 bench.py --no-cpu-baseline "$@" $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline') or {}
print('$label', d['value'], d['unit'], d['ms_per_step'], 'ms/step', $fmt)"
}
for r in $(seq $reps); do for spec in "$@"; do case $mode in
  serial) run $spec "$spec" "'| serial: trace ms', round(r['avg_launch_ms']*r['launches'],2), 'shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'trace Mrays/s', r['trace_kernel_mrays_per_s']" --steps 8 --warmup 4 ;;
  share) for sc in ${SCENES:-dragon rtcamp}; do for t in ${TILES:-0 8}; do run $spec "$spec $sc tiles=$t" "''" --scene $sc --no-serial-pass --steps 8 --warmup 4 $([ $t != 0 ] && echo --emulate-tiles $t); done; done ;;
  stress) run $spec "$spec" "'| descents/ray', r.get('descents_per_ray')" --mode stress --steps 3 --warmup 1 ;;
esac; done; done
