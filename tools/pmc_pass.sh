#!/bin/bash
# usage: tools/pmc_pass.sh <outdir> <counter list...>   (one rocprofv3 --pmc pass over the DEFAULT bench.py command; kernel-trace only)
out=$1; shift
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out" -- python3 bench.py --no-cpu-baseline > "$out/run.log" 2>&1
echo "rc=$? $(tail -c 300 $out/run.log | head -c 200)"
