#!/usr/bin/env python3
"""usage: tools/collect_profiles.py gpurun_out/final_<tag> <prefix>   -- copy the judged summaries of a tools/final_profiles.sh run into profiles/<prefix>_*
and rewrite profiles/traffic_latest.json (memory-side bytes per ray of the traversal kernel from the FETCH_SIZE / WRITE_SIZE passes)."""
import csv, glob, json, os, shutil, subprocess, sys
src, prefix = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from massivevoxelraytracing_amd import build as _build
DIGEST = _build.source_digest()  # of the library sources these counters were taken with (collect right after the run): bench.py reports the traffic only for this binary
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
        "# kPtTraceStream / kPtShade of 4 serial-mode steps; FETCH_SIZE / WRITE_SIZE in KB.  Calibration (tools/calib/fetch_calib.hip, r03): FETCH_SIZE counts 64 B per\n"
        "# divergent gather (factor 1.0: the traversal kernel) and HALF of a coalesced stream, 4 or 16 bytes per lane (the shade kernel's SoA reads: factor 2); WRITE_SIZE is exact)\n" + txt)
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
               "source_digest": DIGEST, "scene": "dragon", "grid_res": 2048, "kernel": "kPtTraceStream", "rays_in_profiled_run": rays, "fetch_bytes_per_ray": round(fetch, 2),
               "write_bytes_per_ray": round(write, 2), "traffic_bytes_per_ray": round(fetch + write, 2)}, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
    print("traffic B/ray", fetch, write)

st = os.path.join(src, "bench_stress.json")
if os.path.exists(st) and os.path.isdir(os.path.join(src, "pmc_stress_FETCH_SIZE")):
    d = json.load(open(st))
    rays = d["config"]["workload"]
    n_rays = int(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 + 0.5) * (d["steps"] + d["warmup"])
    fetch = counter("pmc_stress_FETCH_SIZE", "kTraceBatchStream", "FETCH_SIZE") * 1024 / n_rays
    write = counter("pmc_stress_WRITE_SIZE", "kTraceBatchStream", "WRITE_SIZE") * 1024 / n_rays
    req = counter("pmc_stress_RDREQ", "kTraceBatchStream", "TCC_EA0_RDREQ_sum") / n_rays
    json.dump({"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum passes (separate runs, tools/final_profiles.sh) of `python3 bench.py --mode stress --steps 3 --warmup 1`; fabric-side",
               "source_digest": DIGEST, "mode": "stress", "kernel": "kTraceBatchStream<2>", "rays_in_profiled_run": n_rays, "fetch_bytes_per_ray": round(fetch, 1),
               "write_bytes_per_ray": round(write, 1), "traffic_bytes_per_ray": round(fetch + write, 1), "read_requests_per_ray": round(req, 1)},
              open(os.path.join(P, "traffic_stress.json"), "w"), indent=1)
    print("stress traffic B/ray", fetch, write, "read requests/ray", req)
