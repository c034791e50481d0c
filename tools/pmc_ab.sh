#!/bin/bash
# usage (GPU box): tools/pmc_ab.sh <tag> "<bench args A>" "<bench args B>"  -- SQ instruction / wait counters + FETCH/WRITE of the serial-mode pass for two argument sets
cd ${GRAFT_REPO_ROOT:?run through gpurun}
tag=$1; export TMPDIR=/tmp
out=$PWD/gpurun_out/$tag; rm -rf $out; mkdir -p $out
i=0
for args in "$2" "$3"; do
  i=$((i+1))
  SER="python3 bench.py --serial-only --no-cpu-baseline --warmup 0 $args"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/sq1_$i -- $SER > $out/sq1_$i.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $out/sq2_$i -- $SER > $out/sq2_$i.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f_$i -- $SER > $out/f_$i.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w_$i -- $SER > $out/w_$i.log 2>&1
  echo "== [$args]"; python3 tools/pmc_summarize.py $out/sq1_$i $out/sq2_$i $out/f_$i $out/w_$i | grep Trace
done
