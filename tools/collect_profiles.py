#!/usr/bin/env python3
"""usage: tools/collect_profiles.py gpurun_out/final_<tag> <prefix>   -- copy the judged summaries of a tools/final_profiles.sh run into profiles/<prefix>_*
and rewrite profiles/traffic_latest.json (memory-side bytes per ray of the traversal kernel from the FETCH_SIZE / WRITE_SIZE passes)."""
import csv, glob, json, os, shutil, subprocess, sys
src, prefix = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
for name in ("stats_serial", "stats_pipelined", "stats_stress"):
    if os.path.isdir(os.path.join(src, name)):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "condense_rocprof.py"), os.path.join(src, name), os.path.join(P, "%s_%s_kernel_stats.csv" % (prefix, name[6:]))], stdout=subprocess.DEVNULL)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    if os.path.getsize(f) > 10:
        shutil.copy(f, os.path.join(P, prefix + "_" + os.path.basename(f)))
if os.path.exists(os.path.join(src, "pmc_summary.txt")):
    txt = open(os.path.join(src, "pmc_summary.txt")).read().replace(os.path.abspath(src) + "/", "").replace("/tmp/code/Ushio__MassiveVoxelRayTracing/repo/gpurun_out/" + os.path.basename(src) + "/", "")
    open(os.path.join(P, prefix + "_pmc_summary.txt"), "w").write(
        "# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --serial-only --no-cpu-baseline --warmup 0   (one counter group per pass; sums over the 36 launches of\n"
        "# kPtTraceStream / kPtShade of 4 serial-mode steps; FETCH_SIZE / WRITE_SIZE in KB, gather-calibrated factor 1.0 -- profiles/r01_traffic_pmc.txt)\n" + txt)
def counter(d, kernel, name):
    tot = 0.0
    files = sorted(glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # the newest run only (a re-used tag keeps the older pid-named files beside it)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                tot += float(r["Counter_Value"])
    return tot
bs = os.path.join(src, "bench_serial.json")
if os.path.exists(bs) and os.path.isdir(os.path.join(src, "pmc_FETCH_SIZE")):
    d = json.load(open(bs))
    rays = d["rays"]
    fetch = counter("pmc_FETCH_SIZE", "kPtTraceStream", "FETCH_SIZE") * 1024 / rays
    write = counter("pmc_WRITE_SIZE", "kPtTraceStream", "WRITE_SIZE") * 1024 / rays
    json.dump({"_source": "profiles/%s_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes (separate runs, tools/final_profiles.sh) of `python3 bench.py --serial-only "
                          "--no-cpu-baseline --warmup 0`; gather-calibrated factor 1.0 (tools/calib/fetch_calib.hip, profiles/r01_traffic_pmc.txt); fabric-side bytes, "
                          "Infinity-Cache hits included" % prefix,
               "scene": "dragon", "grid_res": 2048, "kernel": "kPtTraceStream", "rays_in_profiled_run": rays, "fetch_bytes_per_ray": round(fetch, 2),
               "write_bytes_per_ray": round(write, 2), "traffic_bytes_per_ray": round(fetch + write, 2)}, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
    print("traffic B/ray", fetch, write)
