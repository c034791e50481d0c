#!/usr/bin/env python3
"""usage: tools/live_counts.py [scene] [tiles]  -- live paths per stage of one 16-spp step (debug capture), full frame or tile 0 of N"""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import massivevoxelraytracing_amd as mv
from massivevoxelraytracing_amd import scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
res = {"dragon": 2048, "rtcamp": 4096, "cave": 2048}[scene]
v, c, e = scenes.SCENES[scene](1.0)
origin, dps = scenes.bounding_grid(v, res)
pt = mv.PathTracer(); pt.setup(None); pt.set_tile(0, tiles); pt.resizeFrameBufferIfNeeded(None, 1920, 1080)
hdr = "tests/golden/monks_forest_s.hdr"; pt.loadHDRI(None, hdr, hdr)
pt.updateScene(v, c, e, None, origin, dps, res)
info = pt.m_intersectorOctreeGPU.info(); lo, hi = np.array(info.lower[:]), np.array(info.upper[:]); centre = (lo + hi) / 2
if scene == "cave": cam = scenes.cave_camera(lo, hi)
else:
    eye = centre + (np.array([2.6, 1.5, 3.1]) if scene == "dragon" else np.array([4.2, 2.2, 5.0]))
    cam = scenes.look_at_camera(eye, centre, 40.0, float(np.linalg.norm(eye - centre)), 0.02)
pt.set_debug_capture(True); pt.set_batch_steps(1); pt.step(None, cam)
n = pt.owned_pixels() * 16
print(scene, "samples", n, "live after stage k:", [len(pt.debug_stage_survivors(s, n)) for s in range(8)], "rays", pt.stats()["rays"])
