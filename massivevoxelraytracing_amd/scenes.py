"""Seeded procedural triangle scenes -- stand-ins for the reference assets that are not available
(.MISSING_LARGE_BLOBS of the reference: xyzrgb_dragon.abc/.ply; rtcamp9.abc is an external download).

Host-side input preparation only (the role of voxUtil.hpp:trianglesFlattened + getBoundingBox,
voxUtil.hpp:19-77): each scene is a flat list of triangle vertices with per-vertex colour and
emission, exactly what IntersectorOctreeGPU::build / PathTracer::updateScene consume.
Everything is float32 and deterministic (closed-form, no RNG state).
"""
import math

import numpy as np

f32 = np.float32


def _grid_tris(P):
    """P: (nu, nv, 3) vertex grid -> (2*(nu-1)*(nv-1), 3, 3) triangles"""
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([a, b, c], axis=2)
    t2 = np.stack([a, c, d], axis=2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0)


def _bumps(u, v, k):
    return (np.sin(u * k) * np.cos(v * (k + 3)) + 0.5 * np.sin(u * (2 * k + 1) + 1.3) * np.sin(v * (2 * k - 1) + 0.7) + 0.25 * np.cos(u * 4 * k + v * 3 * k))


def torus_knot(p=2, q=3, R=1.0, r=0.28, nu=2048, nv=192, bump=0.06):
    """bumpy tube around a (p,q) torus knot: a closed, self-occluding body (dragon-like occlusion)"""
    u = np.linspace(0, 2 * math.pi, nu, dtype=np.float64)[:, None]
    v = np.linspace(0, 2 * math.pi, nv, dtype=np.float64)[None, :]

    def centre(t):
        rr = R * (1 + 0.45 * np.cos(q * t))
        return np.stack([rr * np.cos(p * t), R * 0.55 * np.sin(q * t), rr * np.sin(p * t)], -1)

    c = centre(u)
    t = centre(u + 1e-4) - c
    t /= np.linalg.norm(t, axis=-1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=-1, keepdims=True)
    n2 = np.cross(t, n1)
    rad = r * (1 + bump / r * _bumps(u, v, 9))
    P = c + rad[..., None] * (np.cos(v)[..., None] * n1 + np.sin(v)[..., None] * n2)
    return _grid_tris(P.astype(f32))


def heightfield(n=1024, size=4.0, amp=0.35, y0=-1.0):
    x = np.linspace(-size / 2, size / 2, n, dtype=np.float64)[:, None]
    z = np.linspace(-size / 2, size / 2, n, dtype=np.float64)[None, :]
    y = y0 + amp * (0.5 * np.sin(1.7 * x + 0.3) * np.cos(1.3 * z) + 0.25 * np.sin(4.1 * x + 2.0 * z) + 0.12 * np.cos(9.0 * x - 7.0 * z) + 0.06 * np.sin(21.0 * x) * np.sin(19.0 * z))
    P = np.stack([np.broadcast_to(x, y.shape), y, np.broadcast_to(z, y.shape)], -1)
    return _grid_tris(P.astype(f32))


def uv_sphere(center, radius, nu=256, nv=128, bump=0.0):
    u = np.linspace(0, 2 * math.pi, nu, dtype=np.float64)[:, None]
    v = np.linspace(1e-3, math.pi - 1e-3, nv, dtype=np.float64)[None, :]
    r = radius * (1 + bump * _bumps(u, v, 5))
    P = np.stack([r * np.cos(u) * np.sin(v), r * np.cos(v) + 0 * u, r * np.sin(u) * np.sin(v)], -1) + np.asarray(center, np.float64)
    return _grid_tris(P.astype(f32))


def box(lo, hi):
    lo, hi = np.asarray(lo, f32), np.asarray(hi, f32)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], f32)
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    t = []
    for a, b, cc, d in quads:
        t.append([c[a], c[b], c[cc]])
        t.append([c[a], c[cc], c[d]])
    return np.asarray(t, f32)


def _colorize(tris, emissive_mask=None, emission=(1.0, 0.8, 0.55)):
    v = tris.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    n = ((v - lo) / np.maximum(hi - lo, 1e-6)).astype(f32)
    cols = np.stack([f32(0.30) + f32(0.65) * n[:, 0], f32(0.35) + f32(0.55) * n[:, 1], f32(0.40) + f32(0.5) * (f32(1) - n[:, 2])], -1).astype(f32)
    emis = np.zeros_like(cols)
    if emissive_mask is not None:
        m = np.repeat(emissive_mask, 3)
        emis[m] = np.asarray(emission, f32)
        cols[m] = f32(0.9)
    return cols.reshape(-1, 3), emis.reshape(-1, 3)


def dragon_standin(detail=1.0):
    """Stand-in for xyzrgb_dragon (configs 2-3 of BASELINE.json): a bumpy (2,3) torus-knot body over a small
    pedestal, with three emissive beads so hasEmission = 1 exercises the extra-sample branch
    (voxKernel.cu:720-739).  ~1.6 M triangles at detail 1."""
    nu, nv = int(2048 * detail), int(192 * detail)
    body = torus_knot(nu=max(nu, 64), nv=max(nv, 16))
    ped = heightfield(n=max(int(384 * detail), 16), size=3.6, amp=0.08, y0=-0.95)
    beads = [uv_sphere(c, 0.09, max(int(96 * detail), 12), max(int(48 * detail), 8)) for c in ((0.0, 0.95, 0.0), (1.1, -0.2, 0.9), (-1.2, 0.1, -0.8))]
    tris = np.concatenate([body, ped] + beads, axis=0)
    em = np.zeros(len(tris), bool)
    em[len(body) + len(ped):] = True
    cols, emis = _colorize(tris, em)
    return tris.reshape(-1, 3), cols, emis


def rtcamp_standin(detail=1.0):
    """Stand-in for rtcamp9.abc (config 4): rolling terrain, a field of bumpy boulders and emissive slabs."""
    parts = [heightfield(n=max(int(1536 * detail), 16), size=8.0, amp=0.6, y0=-1.2)]
    em_parts = []
    k = 0
    for ix in range(-3, 4):
        for iz in range(-3, 4):
            cx, cz = ix * 1.05 + 0.31 * math.sin(3.1 * iz + 0.5), iz * 1.05 + 0.29 * math.cos(2.3 * ix)
            rad = 0.22 + 0.12 * (0.5 + 0.5 * math.sin(1.7 * ix + 2.9 * iz))
            parts.append(uv_sphere((cx, -0.9 + rad * 0.8, cz), rad, max(int(160 * detail), 12), max(int(80 * detail), 8), bump=0.08))
            k += 1
    for s in range(5):
        x = -3.2 + 1.6 * s
        em_parts.append(box((x, 0.9, -0.15 + 0.4 * math.sin(s)), (x + 0.9, 0.98, 0.15 + 0.4 * math.sin(s))))
    geo = np.concatenate(parts, axis=0)
    ems = np.concatenate(em_parts, axis=0)
    tris = np.concatenate([geo, ems], axis=0)
    em = np.zeros(len(tris), bool)
    em[len(geo):] = True
    cols, emis = _colorize(tris, em, emission=(1.0, 0.9, 0.75))
    return tris.reshape(-1, 3), cols, emis


def bounding_grid(vertices, grid_res):
    """origin = bbox min, dps = max extent / gridRes -- voxPTGPU.cpp:159-163 / voxRT.cpp:188-190"""
    v = np.asarray(vertices, f32).reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    size = (hi - lo).astype(f32)
    dps = f32(f32(size.max()) / f32(grid_res))
    return lo.astype(f32), dps


def look_at_camera(eye, target, fovy_deg, focus, lens_r, up=(0.0, 1.0, 0.0)):
    """15 floats of CameraPinhole for a look-at camera (what GetCameraMatrix + initFromPerspective yield)"""
    eye, target, up = (np.asarray(a, np.float64) for a in (eye, target, up))
    front = target - eye
    front /= np.linalg.norm(front)
    right = np.cross(front, up)
    right /= np.linalg.norm(right)
    upv = np.cross(right, front)
    cam = np.zeros(15, f32)
    cam[0:3], cam[3:6], cam[6:9], cam[9:12] = eye, front, upv, right
    cam[12] = math.tan(math.radians(fovy_deg) * 0.5)
    cam[13], cam[14] = lens_r, focus
    return cam
