#!/usr/bin/env python3
"""GPU box: inputs the reference never sees (its apps derive origin / dps from the mesh's bounding box): triangles outside the grid, NaN / inf vertices,
degenerate triangles, zero rays, NaN cameras.  Each case must come back with a result or an error -- never a fault or a hang."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import massivevoxelraytracing_amd as mv
from common import bunny_tris, probe_camera

def case(name, fn):
    try:
        r = fn()
        print("%-44s -> %s" % (name, r), flush=True)
    except mv.MvrtError as e:
        print("%-44s -> MvrtError: %s" % (name, str(e)[:100]), flush=True)

tris = bunny_tris()
v = tris.reshape(-1, 3).copy()
white = np.ones_like(v); black = np.zeros_like(v)
lo = v.min(0); ext = float((v.max(0) - lo).max())
res = 64
def build(vv, origin, dps):
    s = mv.IntersectorOctreeGPU()
    s.build(vv, white[: len(vv)], black[: len(vv)], None, origin, dps, res)
    i = s.info()
    return s, "voxels %d nodes %d" % (i.numberOfVoxels, i.numberOfNodes)
case("mesh inside the grid", lambda: build(v, lo, ext / res)[1])
case("grid covers a corner of the mesh only", lambda: build(v, lo + 0.4 * ext, 0.2 * ext / res)[1])
case("grid far away from the mesh", lambda: build(v, lo + 100 * ext, ext / res)[1])
vn = v.copy(); vn[5] = np.nan
case("one NaN vertex", lambda: build(vn, lo, ext / res)[1])
vi = v.copy(); vi[7, 1] = np.inf
case("one inf vertex", lambda: build(vi, lo, ext / res)[1])
vd = v.copy(); vd[3:6] = vd[3]
case("degenerate (point) triangle", lambda: build(vd, lo, ext / res)[1])
vh = v.copy(); vh[9:12] *= 1e30
case("huge triangle", lambda: build(vh, lo, ext / res)[1])
s, _ = build(v, lo, ext / res)
case("zero rays", lambda: {k: len(x) for k, x in s.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)).items()})
ro = np.array([[np.nan, 0, 0], [0, 0, 0], [1e30, 1e30, 1e30], [0, 0, 0]], np.float32)
rd = np.array([[1, 0, 0], [0, 0, 0], [1, 1, 1], [np.inf, 1, np.nan]], np.float32)
case("NaN / zero / huge / inf rays", lambda: s.intersect(ro, rd)["t"])
pt = mv.PathTracer(); pt.setup(None); pt.resizeFrameBufferIfNeeded(None, 64, 36); pt.set_hdri_scale(0.0)
pt.updateScene(v, white, black, None, lo, ext / res, res)
cam = np.array(probe_camera(lo, np.float32(ext / res), res), np.float32, copy=True)
cam[:3] = np.nan
def nan_cam():
    pt.step(None, cam); fb = pt.read_framebuffer(); return "frame buffer finite: %s, weight %g" % (np.isfinite(fb[:, :3]).all(), fb[:, 3].max())
case("NaN camera origin", nan_cam)
print("probe finished")
