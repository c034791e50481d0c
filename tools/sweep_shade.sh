#!/bin/bash
# usage (GPU box): tools/sweep_shade.sh -- which wavefront stages gain from the shade kernel's 8-waves-per-SIMD build (MVRT_EXPERIMENT build "exp"): serial-mode shade / frame times
cd ${GRAFT_REPO_ROOT:?run through gpurun}
export MVRT_LIB=$PWD/build/ab/libmvrt_exp.so
for r in 1 2; do for m in 0 1 510 511 3 6; do for sc in dragon cave; do
  MVRT_SHADE8_STAGES=$m python3 bench.py --scene $sc --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('mask=$m $sc', d['ms_per_step'], 'ms/step; serial shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'share', r['shade_share_of_kernel_time'])"
done; done; done
