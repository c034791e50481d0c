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
def sgprs(tok):
    """the SGPR numbers an operand names: s5 -> {5}, s[4:5] -> {4, 5}, vcc -> {'vcc'}"""
    tok = tok.strip()
    if tok.startswith("vcc"):
        return {"vcc"}
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def valu_sgpr_writes(p):
    """SGPRs a VALU instruction writes: v_cmp* / v_cmpx* destinations (implicit vcc for _e32), the carry-out of v_add_co / v_sub_co / v_addc_co / ...,
    v_div_scale's second destination, v_readlane / v_readfirstlane, v_mad_u64_u32's carry"""
    op = p.split()[0]
    ops = [t for t in p[len(op):].split(",")]
    if op.startswith(("v_cmp", "v_cmpx")):
        return {"vcc"} if op.endswith("_e32") else sgprs(ops[0])
    if op.startswith(("v_readfirstlane", "v_readlane")):
        return sgprs(ops[0])
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co", op) or op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")):
        return sgprs(ops[1]) if len(ops) > 1 else set()
    return set()


bad = n = 0
for k, l in enumerate(ins):
    m = re.match(r"v_cndmask_b32 (\S+), (\S+), (\S+), (s\[\d+:\d+\]|vcc)$", l)
    if not m:
        continue
    n += 1
    mask = sgprs(m.group(4))
    for back in (1, 2):
        p = ins[k - back]
        if p.startswith("v_") and valu_sgpr_writes(p) & mask:  # any overlap with the mask pair
            bad += 1
            print("HAZARD?", p, " ->", l)
# (the scan is linear: a select that is the target of a branch is checked against the instructions that precede it in the text, not against the
# branch's source block -- every asm select of traverse_stream.h sits in straight-line code behind SALU-produced masks)
print("asm v_cndmask selects: %d, suspicious producer within 2 instructions: %d" % (n, bad))
sys.exit(1 if bad else 0)
