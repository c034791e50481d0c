#!/usr/bin/env python3
"""GPU box: tiny frames and tile shares without pixels; handles destroyed with work pending; two path tracers side by side."""
import sys, os, gc
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import massivevoxelraytracing_amd as mv
from common import bunny_tris, probe_camera
tris = bunny_tris(); v = tris.reshape(-1, 3); white = np.ones_like(v); black = np.zeros_like(v)
lo = v.min(0); ext = float((v.max(0) - lo).max()); res = 64
cam = probe_camera(lo, np.float32(ext / res), res)
def mk(w, h, tile=(0, 1)):
    pt = mv.PathTracer(); pt.setup(None); pt.set_tile(*tile); pt.resizeFrameBufferIfNeeded(None, w, h); pt.set_hdri_scale(0.0)
    pt.updateScene(v, white, black, None, lo, ext / res, res); return pt
for w, h, tile in ((1, 1, (0, 1)), (3, 2, (0, 1)), (100, 37, (31, 32)), (100, 37, (14, 32)), (17, 1, (1, 2))):
    pt = mk(w, h, tile)
    for _ in range(2): pt.step(None, cam)
    fb = pt.read_framebuffer()
    print("frame %dx%d tile %s: owned %d, weights %s, rays %d" % (w, h, tile, pt.owned_pixels(), sorted(set(fb[:, 3].tolist()))[:3], pt.stats()["rays"]), flush=True)
a, b = mk(64, 36), mk(48, 27)
for _ in range(3): a.step(None, cam); b.step(None, cam)
print("two tracers:", a.read_framebuffer()[:, 3].max(), b.read_framebuffer()[:, 3].max(), flush=True)
c = mk(640, 360)
for _ in range(5): c.step(None, cam)
del c; gc.collect()   # destroyed with deferred / in-flight steps
d = mk(64, 36); d.step(None, cam); d.updateScene(v, white, black, None, lo, ext / res, 128); d.step(None, cam)
print("rebuild between steps:", d.read_framebuffer()[:, 3].max(), flush=True)
print("probe finished")
