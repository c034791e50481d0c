#!/bin/bash
# usage (GPU box): tools/sweep_shade.sh -- shade kernel: which stages run the 8-waves-per-SIMD build, workgroups per CU of the two builds' grids (MVRT_EXPERIMENT build "exp")
cd ${GRAFT_REPO_ROOT:?run through gpurun}
export MVRT_LIB=$PWD/build/ab/libmvrt_exp.so
IFS=";" read -ra CFGS <<< "${SWEEP:-3 8 8;3 8 5;3 8 6;3 8 10;3 8 16;0 8 5;0 8 6;3 16 6;3 32 6}"
for r in 1 2; do for cfg in "${CFGS[@]}"; do set -- $cfg; for sc in dragon cave; do
  MVRT_SHADE8_STAGES=$1 MVRT_SHADE_BPC8=$2 MVRT_SHADE_BPC5=$3 python3 bench.py --scene $sc --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dense=$1 bpc8=$2 bpc5=$3 $sc', d['ms_per_step'], 'ms/step; serial shade ms', round(r['shade_share_of_kernel_time']*r['sum_kernel_ms'],2), 'share', r['shade_share_of_kernel_time'])"
done; done; done
