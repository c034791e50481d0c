#!/bin/bash
# usage (on the GPU box, repo root): tools/final_profiles.sh <tag> [part]   -- everything the round's profiles/ entries are made from, into gpurun_out/final_<tag>/
# parts: a = default line + serial-mode kernel stats + PMC traffic / SQ counter passes; b = the other configurations; (default: both)
set -o pipefail
tag=${1:-r02}; part=${2:-ab}; root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/final_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
SER="python3 bench.py --serial-only --no-cpu-baseline --warmup 0"
if [[ $part == *a* ]]; then
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_serial -- $SER > $out/stats_serial.log 2>&1; echo "serial stats rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_pipelined -- python3 bench.py --no-cpu-baseline --no-serial-pass > $out/stats_pipelined.log 2>&1; echo "pipelined stats rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- $SER > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
  done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq1 -- $SER > $out/pmc_sq1.log 2>&1; echo "pmc sq1 rc=$?"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $out/pmc_sq2 -- $SER > $out/pmc_sq2.log 2>&1; echo "pmc sq2 rc=$?"
  python3 tools/pmc_summarize.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq1 $out/pmc_sq2 > $out/pmc_summary.txt 2>&1
  grep "^{\"metric\"" $out/stats_serial.log | tail -1 > $out/bench_serial.json
fi
if [[ $part == *b* ]]; then
  python3 bench.py --mode primary --grid-res 1024 --no-cpu-baseline > $out/bench_primary_1024.json 2>/dev/null; echo "primary rc=$?"
  python3 bench.py --scene rtcamp --steps 16 --no-cpu-baseline > $out/bench_rtcamp_4096_256spp.json 2>/dev/null; echo "rtcamp rc=$?"
  python3 bench.py --scene cave > $out/bench_cave_2048.json 2>/dev/null; echo "cave rc=$?"
  python3 bench.py --mode stress --steps 3 --warmup 1 > $out/bench_stress.json 2>/dev/null; echo "stress rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_stress -- python3 bench.py --mode stress --steps 3 --warmup 1 > $out/stats_stress.log 2>&1; echo "stress stats rc=$?"
  for sc in dragon rtcamp; do for n in 2 4 8; do python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --emulate-tiles $n 2>/dev/null | tail -1 > $out/bench_${sc}_emulate_tiles_$n.json; done
    python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass 2>/dev/null | tail -1 > $out/bench_${sc}_emulate_tiles_1.json; done
fi
tail -c 300 $out/bench_default.json 2>/dev/null; true
