#!/usr/bin/env python3
"""usage: tools/stage_times.py <rocprofv3 output dir>  -- per-stage kPtTraceStream / kPtShade durations of the first serial-mode step in a kernel trace"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tr = [r for r in rows if "kPtTraceStream" in r["Kernel_Name"]]
sh = [r for r in rows if "kPtShade" in r["Kernel_Name"]]
n = len(tr) // 9
for step in (n - 1,):
    print("step", step, "trace us:", [round(d(tr[step * 9 + k])) for k in range(9)], "shade us:", [round(d(sh[step * 9 + k])) for k in range(9)])
