#!/usr/bin/env python3
"""Print the kPt* kernel timeline of a rocprofv3 --kernel-trace csv (relative ms, queue id), optionally only the last N rows."""
import csv, glob, sys, os
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].replace("void ", "").startswith(("kPt", "kScanBlock"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
if last: rows = rows[-last:]
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:16]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-16s q%-2s grid %8s %9.3f -> %9.3f (%7.3f ms)" % (n, r["Queue_Id"], r["Grid_Size_X"], s, e, e - s))
