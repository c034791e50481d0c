#!/usr/bin/env python3
"""Traversal event tallies from the CPU oracle (candidate tests, node visits, descents, pushes, fruitful pops per ray) on a
path-traced bunny -- the numbers the node-visit step of traverse_stream.h was designed from.   usage: tools/traversal_events.py [res ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import bunny_tris, hdr_bytes, position_colors, probe_camera
from oracle import oracle as O
O.build()
lib = C.CDLL(os.path.join(ROOT, "oracle", "libmvrt_oracle.so"))
def events(reset=True):
    out = (C.c_uint64 * 8)(); lib.orc_trace_events(out, int(reset)); return np.array(out[:], dtype=np.float64)
tris = bunny_tris(); cols, emis = position_colors(tris)
names = ["rays", "candidate tests", "descents", "pushes", "pops", "fruitful pops", "leaf checks", "node visits"]
lib.orc_trace_events_enable(1)
for res in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    sc = O.build_scene_from_triangles(tris, res, cols, emis)
    rgba, hw, hh = O.decode_rgbe(hdr_bytes())
    cam = probe_camera(sc.origin, sc.dps, res, focus=9.0, lens_r=0.05)
    events()
    sc.render_pt(O.HDRI(rgba, hw, hh, rgba, hw, hh, 1), cam, 320, 180, 0, math_mode=1, threads=8)
    e = events()
    print("bunny %d^3, 320x180x16spp path trace: %d rays entering the root box" % (res, int(e[0])))
    for k in range(1, 8):
        print("  %-16s %6.2f per ray" % (names[k], e[k] / e[0]))
lib.orc_trace_events_enable(0)
