#!/bin/bash
# usage (GPU box): tools/stress_rdreq.sh  -- sizes of the fabric read requests of the config-5 traversal launches
set -o pipefail
root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/stress_rdreq; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/pmc_rd -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/pmc_rd.log 2>&1; echo "rc=$?"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/pmc_tcc.log 2>&1; echo "rc=$?"
python3 tools/pmc_summarize.py $out/pmc_rd $out/pmc_tcc | tee $out/summary.txt
