"""Shared helpers for the test-suite (scene fixtures, probe camera)."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
f32 = np.float32


def golden():
    with open(os.path.join(GOLDEN, "survey_appendix_a.json")) as f:
        return json.load(f)


def bunny_tris():
    """scenes/bunny.obj of the reference as a raw triangle soup (tools/make_fixtures.py)."""
    return np.fromfile(os.path.join(GOLDEN, "bunny_tris.f32"), np.float32).reshape(-1, 9)


def hdr_bytes():
    with open(os.path.join(GOLDEN, "monks_forest_s.hdr"), "rb") as f:
        return f.read()


def probe_camera(origin, dps, res, focus=1.0, lens_r=0.0, fovy=45.0, offset=(6, 4, 6)):
    """SURVEY.md Appendix A 'Probe camera': look-at from scene centre + (6,4,6), fovy 45 deg.
    Returns the 15 floats of CameraPinhole {o, front, up, right, tanHthetaY, lensR, focus}."""
    origin = np.asarray(origin, f32)
    c = (origin + f32(0.5) * f32(dps) * f32(res) * np.ones(3, f32)).astype(f32)
    o = (c + np.asarray(offset, f32)).astype(f32)
    d = (c - o).astype(f32)

    def norm(v):
        l2 = f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
        return (v / f32(np.sqrt(l2))).astype(f32)

    front = norm(d)
    right = norm(np.array([-front[2], 0, front[0]], f32))
    up = np.cross(right, front).astype(f32)
    tan_h = f32(math.tan(float(f32(f32(f32(0.5) * f32(fovy)) * f32(3.14159265)) / f32(180))))
    cam = np.zeros(15, f32)
    cam[0:3], cam[3:6], cam[6:9], cam[9:12] = o, front, up, right
    cam[12], cam[13], cam[14] = tan_h, lens_r, focus
    return cam


def position_colors(tris):
    """Deterministic per-vertex colours/emissions for PT scenes: colour from normalised position,
    emission on the top 8 % of the bbox height (so hasEmission = 1, voxKernel.cu:720-739)."""
    v = tris.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    n = ((v - lo) / (hi - lo)).astype(f32)
    cols = (f32(0.25) + f32(0.7) * n).astype(f32)
    emis = np.zeros_like(cols)
    top = n[:, 1] > f32(0.92)
    emis[top] = np.array([1.0, 0.85, 0.6], f32)
    return cols.reshape(-1, 9), emis.reshape(-1, 9)
