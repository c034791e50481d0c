#!/bin/bash
# usage (GPU box): tools/shade_traffic.sh <tag>  -- FETCH_SIZE / WRITE_SIZE calibration for 4-byte-per-lane coalesced streams + the shade kernel's counters in a serial-mode pass
cd ${GRAFT_REPO_ROOT:?run through gpurun}
tag=$1; export TMPDIR=/tmp; out=$PWD/gpurun_out/$tag; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/cal_$c -- build/fetch_calib > $out/cal_$c.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$out/cal_$c/**/*counter_collection.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("calib", r["Kernel_Name"].split("(")[0], r["Counter_Name"], float(r["Counter_Value"])*1024/2**30, "GiB counted (2 GiB moved by the stream kernels)")
PY
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pt_$c -- python3 bench.py --serial-only --no-cpu-baseline --warmup 0 > $out/pt_$c.log 2>&1
done
python3 tools/pmc_summarize.py $out/pt_FETCH_SIZE $out/pt_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --serial-only --no-cpu-baseline --warmup 0 > $out/stats.log 2>&1
python3 tools/condense_rocprof.py $out/stats $out/kernel_stats.csv | head -8
