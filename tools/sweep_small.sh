#!/bin/bash
# usage (GPU box): tools/sweep_small.sh -- 1/8 tile share: waves kept alive by small traversal launches (rays per lane x minimum waves; MVRT_EXPERIMENT build "exp")
cd ${GRAFT_REPO_ROOT:?run through gpurun}
export MVRT_LIB=$PWD/build/ab/libmvrt_exp.so
IFS=";" read -ra CFGS <<< "${SWEEP:-16 2048;8 2048;32 2048;16 1024;16 4096}"
for cfg in "${CFGS[@]}"; do set -- $cfg
  for sc in dragon rtcamp; do
  MVRT_SMALL_RPL=$1 MVRT_SMALL_MINW=$2 python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --steps 8 --warmup 4 --emulate-tiles 8 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1])
print('rpl=$1 minw=$2 $sc', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step')"
done; done
