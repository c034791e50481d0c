#!/usr/bin/env python3
"""Scan the gfx950 ISA of kernels_rt.hip for a hazard the compiler cannot pad because the reader sits in inline asm:
on gfx940+ a VALU instruction that writes an SGPR (v_cmp into an SGPR pair / vcc, v_readfirstlane ...) must be followed by 2 wait
states before a VALU instruction reads that SGPR (here: the asm v_cndmask_b32 selects, recognisable by their missing _e32/_e64
suffix).  Exit code 1 if a suspicious pair is found.   usage: tools/check_asm_hazards.py [file.s]  (compiles the ISA if no file)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = os.path.join(tempfile.mkdtemp(), "kernels_rt.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                           "-Wno-unused-command-line-argument", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "massivevoxelraytracing_amd", "csrc", "kernels_rt.hip"), "-o", path])
ins = []
for l in open(path):
    if re.match(r"\s+[vs]_|\s+ds_|\s+global_|\s+flat_|\s+scratch_", l):
        ins.append(l.strip())
        m = re.match(r"\s+s_nop (\d+)", l)
        if m:  # s_nop N = N + 1 wait states
            ins.extend(["s_nop (wait state)"] * int(m.group(1)))
bad = n = 0
for k, l in enumerate(ins):
    m = re.match(r"v_cndmask_b32 (\S+), (\S+), (\S+), (s\[(\d+):(\d+)\]|vcc)$", l)
    if not m:
        continue
    n += 1
    mask = m.group(4)
    lo = m.group(5)
    for back in (1, 2):
        p = ins[k - back]
        writes = p.startswith("v_cmp") and (mask in p.split(",")[0] or (mask == "vcc" and "_e32" in p.split()[0]))
        writes |= p.startswith(("v_readfirstlane", "v_readlane")) and lo is not None and re.search(r"\bs%s\b" % lo, p.split(",")[0]) is not None
        if writes:
            bad += 1
            print("HAZARD?", p, " ->", l)
print("asm v_cndmask selects: %d, suspicious producer within 2 instructions: %d" % (n, bad))
sys.exit(1 if bad else 0)
