#!/usr/bin/env python3
"""Debug aid: batch traversal vs the oracle on bunny 256^3; prints mismatch statistics, runs the GPU twice (determinism)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import bunny_tris
from oracle import oracle as O
import massivevoxelraytracing_amd as mv
from test_gpu_parity import random_rays
sc = O.build_scene_from_triangles(bunny_tris(), 256)
svo = mv.IntersectorOctreeGPU()
svo.upload(sc.nodes, sc.attrs, sc.origin, sc.dps, sc.grid_res, sc.has_emission, embeddedMask=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ro, rd = random_rays(sc, n, 7)
sh = (np.arange(n) % 3 == 0).astype(np.uint8)
want = sc.trace(ro, rd, sh, threads=8, want_descents=True)
a = svo.intersect(ro, rd, sh, want_descents=True)
b = svo.intersect(ro, rd, sh, want_descents=True)
for k in ("t", "nMajor", "vIndex", "descents"):
    print(k, "gpu!=oracle:", int((a[k] != want[k]).sum()), " gpu run1!=run2:", int((a[k] != b[k]).sum()))
bad = np.nonzero((a["t"] != want["t"]) | (a["descents"] != want["descents"]))[0]
print("bad rays", len(bad), bad[:10])
for i in bad[:8]:
    print(i, "want t %.6g nm %d d %d | got t %.6g nm %d d %d | shadow %d" % (want["t"][i], want["nMajor"][i], want["descents"][i], a["t"][i], a["nMajor"][i], a["descents"][i], sh[i]))
