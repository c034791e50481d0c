"""The traversal step reads lane masks from SGPR pairs inside inline asm (v_cndmask_b32 selects).  On gfx940+ an SGPR written by a
VALU instruction needs two wait states before a VALU instruction reads it; the compiler pads for its own instructions but cannot
see a reader inside asm.  tools/check_asm_hazards.py compiles kernels_rt.hip to gfx950 ISA and scans for such a pair (found the
hard way: a related store-data hazard made 69 of 200 000 rays differ from run to run in an experiment of this round)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="hipcc not available")
def test_no_valu_written_sgpr_read_by_asm_select():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asm v_cndmask selects" in r.stdout
