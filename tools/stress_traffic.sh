#!/bin/bash
# usage (GPU box): tools/stress_traffic.sh  -- fabric-side bytes of the config-5 traversal launches (separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --mode stress`)
set -o pipefail
root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/stress_traffic; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python3 tools/pmc_summarize.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE | tee $out/summary.txt
grep "^{\"metric\"" $out/pmc_FETCH_SIZE.log | tail -1 > $out/bench_stress.json
