#!/bin/bash
# usage (GPU box): tools/clock_probe.sh <variant|default> ...  -- effective shader clock of the big traversal launches: GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS)
root=${GRAFT_REPO_ROOT:?run through gpurun}; cd /tmp && export TMPDIR=/tmp && cd $root
for v in "$@"; do
  if [ "$v" = default ]; then unset MVRT_LIB; else export MVRT_LIB=$root/build/ab/libmvrt_$v.so; fi
  rm -rf gpurun_out/clk_$v
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_$v -- python3 bench.py --serial-only --no-cpu-baseline --warmup 1 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,glob,sys
v=sys.argv[1]
f=glob.glob('gpurun_out/clk_%s/**/*counter_collection.csv'%v,recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'kPtTraceStream' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE']
out=[]
for r in rows:
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    if dur>3e6: out.append((float(r['Counter_Value'])/8/dur, dur/1e6))
print(v, 'big launches:', len(out), 'clock GHz', ' '.join('%.3f'%c for c,_ in out[:8]), 'ms', ' '.join('%.2f'%d for _,d in out[:8]))
PY
done
