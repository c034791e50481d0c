#!/bin/bash
# usage (GPU box): tools/stress_sq.sh  -- issue / wait counters of the config-5 traversal launches
set -o pipefail
root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/stress_sq; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $out/pmc_sq2 -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/pmc_sq2.log 2>&1; echo "rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq1 -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/pmc_sq1.log 2>&1; echo "rc=$?"
python3 tools/pmc_summarize.py $out/pmc_sq1 $out/pmc_sq2 | tee $out/summary.txt
