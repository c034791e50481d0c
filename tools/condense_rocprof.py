#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a short, committable summary.

    python tools/condense_rocprof.py gpurun_out/prof1 profiles/r01_bench_kernel_stats.csv
"""
import csv, glob, os, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if "radix_sort_onesweep_iteration" in name:
        return "rocprim::radix_sort_onesweep_iteration<...>"
    if "rocprim" in name:
        m = re.search(r"detail::(\w+)", name)
        return "rocprim::" + (m.group(1) if m else "kernel") + "<...>"
    return re.sub(r"\(.*", "", name)


def main(src, dst):
    stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]  # the newest run
    rows = {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            a = rows.setdefault(k, [0, 0.0, 1e30, 0.0])
            a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
    tot = sum(a[1] for a in rows.values())
    with open(dst, "w") as f:
        f.write("# source: rocprofv3 --kernel-trace --stats (kernel_stats.csv), names shortened, same-name rows merged\n")
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent\n")
        for k, a in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            f.write("%s,%d,%.3f,%.1f,%.1f,%.1f,%.2f\n" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / tot))
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
