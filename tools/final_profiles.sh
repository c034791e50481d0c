#!/bin/bash
# usage (on the GPU box, repo root): tools/final_profiles.sh <tag> [part]   -- everything the round's profiles/ entries are made from, into gpurun_out/final_<tag>/
# parts: a = default line + serial-mode kernel stats + PMC traffic / SQ counter passes; b = the other configurations; c = config 5 (stress) incl. its counter passes; d = tile-share emulation
set -o pipefail
tag=${1:-r03}; part=${2:-abcd}; root=${GRAFT_REPO_ROOT:?run through gpurun}; out=$root/gpurun_out/final_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
SER="python3 bench.py --serial-only --no-cpu-baseline --warmup 0"
if [[ $part == *a* ]]; then
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench default rc=$?"
  python3 bench.py --steps 20 --warmup 5 > $out/bench_default_steps20.json 2>/dev/null; echo "bench 20 steps rc=$?"
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
  python3 bench.py --mode cpu-primary --steps 3 > $out/bench_cpu_primary_256.json 2>/dev/null; echo "cpu-primary rc=$?"
  python3 bench.py --mode primary --grid-res 1024 --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_primary_1024.json 2>/dev/null; echo "primary 1024 rc=$?"
  python3 bench.py --mode primary --grid-res 8192 --no-cpu-baseline --steps 20 --warmup 3 > $out/bench_primary_8192.json 2>/dev/null; echo "primary 8192 rc=$?"
  python3 bench.py --scene rtcamp --steps 16 --no-cpu-baseline > $out/bench_rtcamp_4096_256spp.json 2>/dev/null; echo "rtcamp rc=$?"
  python3 bench.py --scene cave > $out/bench_cave_2048.json 2>/dev/null; echo "cave rc=$?"
  python3 bench.py --scene tunnel > $out/bench_tunnel_4096.json 2>/dev/null; echo "tunnel rc=$?"
fi
if [[ $part == *c* ]]; then
  ST="python3 bench.py --mode stress --steps 3 --warmup 1"
  $ST > $out/bench_stress.json 2>/dev/null; echo "stress rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_stress -- $ST > $out/stats_stress.log 2>&1; echo "stress stats rc=$?"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_stress_FETCH_SIZE -- $ST > $out/pmc_stress_f.log 2>&1; echo "stress fetch rc=$?"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_stress_WRITE_SIZE -- $ST > $out/pmc_stress_w.log 2>&1; echo "stress write rc=$?"
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_stress_RDREQ -- $ST > $out/pmc_stress_r.log 2>&1; echo "stress rdreq rc=$?"
fi
if [[ $part == *d* ]]; then
  for sc in dragon rtcamp tunnel; do for n in 1 2 4 8; do
    python3 bench.py --scene $sc --no-cpu-baseline --no-serial-pass --steps 8 --warmup 4 $([ $n != 1 ] && echo --emulate-tiles $n) 2>/dev/null | tail -1 > $out/bench_${sc}_emulate_tiles_$n.json; done; done
  for n in 1 8; do python3 bench.py --scene rtcamp --no-cpu-baseline --no-serial-pass --steps 16 --warmup 16 --frame-steps 16 $([ $n != 1 ] && echo --emulate-tiles $n) 2>/dev/null | tail -1 > $out/bench_rtcamp_256spp_emulate_tiles_$n.json; done
  python3 - <<PY
import json
for sc in ("dragon","rtcamp","tunnel"):
    t={n:json.load(open("$out/bench_%s_emulate_tiles_%d.json"%(sc,n)))["ms_per_step"] for n in (1,2,4,8)}
    print(sc, "64 spp ms/step", t, "speedup", {n:round(t[1]/t[n],2) for n in (2,4,8)})
t={n:json.load(open("$out/bench_rtcamp_256spp_emulate_tiles_%d.json"%n))["ms_per_step"] for n in (1,8)}
print("rtcamp 256 spp ms/step", t, "speedup at 8", round(t[1]/t[8],2))
PY
fi
tail -c 300 $out/bench_default.json 2>/dev/null; true
