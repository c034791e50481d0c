#!/bin/bash
# usage (GPU box): tools/cache_counters.sh <tag> [bench args]  -- L1 / L2 hit-rate counters of the serial-mode pass (separate rocprofv3 --pmc passes)
set -o pipefail
tag=${1:-cache}; shift; root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc -- python3 bench.py --serial-only --no-cpu-baseline --warmup 0 "$@" > $out/pmc_tcc.log 2>&1; echo "tcc rc=$?"
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_tcp -- python3 bench.py --serial-only --no-cpu-baseline --warmup 0 "$@" > $out/pmc_tcp.log 2>&1; echo "tcp rc=$?"
python3 tools/pmc_summarize.py $out/pmc_tcc $out/pmc_tcp | tee $out/summary.txt
