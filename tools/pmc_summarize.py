#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel (short names)."""
import csv, glob, os, re, sys, collections
def short(n):
    return re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n))[:40]
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:  # the newest run
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "Trace" in k or "Shade" in k:
                print(d, k, {c: ("%.4g" % x) for c, x in v.items()})
