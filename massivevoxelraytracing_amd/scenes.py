"""Seeded procedural triangle scenes -- stand-ins for the reference assets that are not available
(.MISSING_LARGE_BLOBS of the reference: xyzrgb_dragon.abc/.ply; rtcamp9.abc is an external download).

Host-side input preparation only (the role of voxUtil.hpp:trianglesFlattened + getBoundingBox,
voxUtil.hpp:19-77): each scene is a flat list of triangle vertices with per-vertex colour and
emission, exactly what IntersectorOctreeGPU::build / PathTracer::updateScene consume.
Everything is float32 and deterministic (closed-form, no RNG state).  sin / cos / tan come from `dsin` / `dcos` below -- float64
polynomials made of IEEE +, -, * only -- not from libm or numpy's SIMD kernels, whose last bits depend on the host CPU (AVX-512 vs AVX2
code paths): the triangle soups are therefore bit-identical on every machine, and tests/test_scenes.py pins their SHA-256.
"""
import numpy as np

f32 = np.float32
PI = 3.141592653589793

_PIO2_HI = 1.5707963267341256  # pi/2 split in two (Cody-Waite): the high part has 33 significant bits, so k * _PIO2_HI is exact for |k| < 2^20
_PIO2_LO = 6.077100506506192e-11
_S = [-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800, -1.0 / 1307674368000, 1.0 / 355687428096000]
_C = [-1.0 / 2, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600, -1.0 / 87178291200, 1.0 / 20922789888000]


def _sincos(x):
    x = np.asarray(x, np.float64)
    k = np.rint(x * (2.0 / PI))
    r = (x - k * _PIO2_HI) - k * _PIO2_LO  # |r| <= pi/4 (+ rounding)
    r2 = r * r
    ps = np.full_like(r, _S[-1])
    pc = np.full_like(r, _C[-1])
    for a, b in zip(_S[-2::-1], _C[-2::-1]):
        ps = ps * r2 + a
        pc = pc * r2 + b
    sn = r + r * (r2 * ps)
    cs = 1.0 + r2 * pc
    q = k.astype(np.int64) & 3
    s = np.where(q == 0, sn, np.where(q == 1, cs, np.where(q == 2, -sn, -cs)))
    c = np.where(q == 0, cs, np.where(q == 1, -sn, np.where(q == 2, -cs, sn)))
    return s, c


def dsin(x):
    return _sincos(x)[0]


def dcos(x):
    return _sincos(x)[1]


def _fsin(x):
    return float(dsin(np.float64(x)))


def _fcos(x):
    return float(dcos(np.float64(x)))


def _grid_tris(P):
    """P: (nu, nv, 3) vertex grid -> (2*(nu-1)*(nv-1), 3, 3) triangles"""
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([a, b, c], axis=2)
    t2 = np.stack([a, c, d], axis=2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0)


def _bumps(u, v, k):
    return (dsin(u * k) * dcos(v * (k + 3)) + 0.5 * dsin(u * (2 * k + 1) + 1.3) * dsin(v * (2 * k - 1) + 0.7) + 0.25 * dcos(u * 4 * k + v * 3 * k))


def torus_knot(p=2, q=3, R=1.0, r=0.28, nu=2048, nv=192, bump=0.06):
    """bumpy tube around a (p,q) torus knot: a closed, self-occluding body (dragon-like occlusion)"""
    u = np.linspace(0, 2 * PI, nu, dtype=np.float64)[:, None]
    v = np.linspace(0, 2 * PI, nv, dtype=np.float64)[None, :]

    def centre(t):
        rr = R * (1 + 0.45 * dcos(q * t))
        return np.stack([rr * dcos(p * t), R * 0.55 * dsin(q * t), rr * dsin(p * t)], -1)

    c = centre(u)
    t = centre(u + 1e-4) - c
    t /= np.linalg.norm(t, axis=-1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=-1, keepdims=True)
    n2 = np.cross(t, n1)
    rad = r * (1 + bump / r * _bumps(u, v, 9))
    P = c + rad[..., None] * (dcos(v)[..., None] * n1 + dsin(v)[..., None] * n2)
    return _grid_tris(P.astype(f32))


def heightfield(n=1024, size=4.0, amp=0.35, y0=-1.0):
    x = np.linspace(-size / 2, size / 2, n, dtype=np.float64)[:, None]
    z = np.linspace(-size / 2, size / 2, n, dtype=np.float64)[None, :]
    y = y0 + amp * (0.5 * dsin(1.7 * x + 0.3) * dcos(1.3 * z) + 0.25 * dsin(4.1 * x + 2.0 * z) + 0.12 * dcos(9.0 * x - 7.0 * z) + 0.06 * dsin(21.0 * x) * dsin(19.0 * z))
    P = np.stack([np.broadcast_to(x, y.shape), y, np.broadcast_to(z, y.shape)], -1)
    return _grid_tris(P.astype(f32))


def uv_sphere(center, radius, nu=256, nv=128, bump=0.0):
    u = np.linspace(0, 2 * PI, nu, dtype=np.float64)[:, None]
    v = np.linspace(1e-3, PI - 1e-3, nv, dtype=np.float64)[None, :]
    r = radius * (1 + bump * _bumps(u, v, 5))
    P = np.stack([r * dcos(u) * dsin(v), r * dcos(v) + 0 * u, r * dsin(u) * dsin(v)], -1) + np.asarray(center, np.float64)
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
            cx, cz = ix * 1.05 + 0.31 * _fsin(3.1 * iz + 0.5), iz * 1.05 + 0.29 * _fcos(2.3 * ix)
            rad = 0.22 + 0.12 * (0.5 + 0.5 * _fsin(1.7 * ix + 2.9 * iz))
            parts.append(uv_sphere((cx, -0.9 + rad * 0.8, cz), rad, max(int(160 * detail), 12), max(int(80 * detail), 8), bump=0.08))
            k += 1
    for s in range(5):
        x = -3.2 + 1.6 * s
        em_parts.append(box((x, 0.9, -0.15 + 0.4 * _fsin(s)), (x + 0.9, 0.98, 0.15 + 0.4 * _fsin(s))))
    geo = np.concatenate(parts, axis=0)
    ems = np.concatenate(em_parts, axis=0)
    tris = np.concatenate([geo, ems], axis=0)
    em = np.zeros(len(tris), bool)
    em[len(geo):] = True
    cols, emis = _colorize(tris, em, emission=(1.0, 0.9, 0.75))
    return tris.reshape(-1, 3), cols, emis


def _panel(p0, eu, ev, en, nu, nv, amp, k=7, hole=None):
    """bumpy rectangular sheet p0 + u*eu + v*ev + bump(u,v)*en, u,v in [0,1]; `hole` = (u0,u1,v0,v1) leaves a window"""
    u = np.linspace(0.0, 1.0, nu, dtype=np.float64)[:, None]
    v = np.linspace(0.0, 1.0, nv, dtype=np.float64)[None, :]
    b = amp * _bumps(u * 2 * PI, v * 2 * PI, k)
    P = (np.asarray(p0, np.float64) + u[..., None] * np.asarray(eu, np.float64) + v[..., None] * np.asarray(ev, np.float64) + b[..., None] * np.asarray(en, np.float64))
    t = _grid_tris(P.astype(f32))
    if hole is not None:
        nq = (nu - 1) * (nv - 1)
        iu = np.repeat(np.arange(nu - 1), nv - 1)
        iv = np.tile(np.arange(nv - 1), nu - 1)
        cu, cv = (iu + 0.5) / (nu - 1), (iv + 0.5) / (nv - 1)
        keep = ~((cu > hole[0]) & (cu < hole[1]) & (cv > hole[2]) & (cv < hole[3]))
        t = np.concatenate([t[:nq][keep], t[nq:][keep]], axis=0)
    return t


def cave_standin(detail=1.0):
    """A CLOSED scene (the rtcamp9 class: camera inside, almost every bounce hits, ~16-18 rays per sample; voxKernel.cu:675,691-760):
    a bumpy room of 4 x 2.4 x 4 units with columns and boulders, lit by emissive ceiling panels and two small windows
    through which the HDRI reaches.  Camera: `cave_camera()`."""
    n = max(int(512 * detail), 16)
    X, Y, Z = 2.0, 1.2, 2.0
    parts = [
        _panel((-X, -Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), (0, 1, 0), n, n, 0.10, 5),                       # floor
        _panel((-X, Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), (0, -1, 0), n, n, 0.08, 6, hole=(0.42, 0.58, 0.40, 0.60)),  # ceiling with a skylight
        _panel((-X, -Y, -Z), (2 * X, 0, 0), (0, 2 * Y, 0), (0, 0, 1), n, n * 3 // 5, 0.06, 8),              # back wall
        _panel((-X, -Y, Z), (2 * X, 0, 0), (0, 2 * Y, 0), (0, 0, -1), n, n * 3 // 5, 0.06, 9, hole=(0.70, 0.82, 0.45, 0.75)),  # front wall with a window
        _panel((-X, -Y, -Z), (0, 0, 2 * Z), (0, 2 * Y, 0), (1, 0, 0), n, n * 3 // 5, 0.06, 7),              # left wall
        _panel((X, -Y, -Z), (0, 0, 2 * Z), (0, 2 * Y, 0), (-1, 0, 0), n, n * 3 // 5, 0.06, 10),             # right wall
    ]
    for i, (cx, cz) in enumerate(((-1.0, -0.9), (0.9, -1.1), (-0.8, 1.0), (1.1, 0.8), (0.1, 0.0))):
        rad = 0.22 + 0.05 * (i % 3)
        for j in range(5):  # a column of stacked bumpy boulders, floor to ceiling
            parts.append(uv_sphere((cx + 0.05 * _fsin(3.0 * j + i), -Y + 0.25 + 0.48 * j, cz + 0.05 * _fcos(2.0 * j + i)), rad, max(int(192 * detail), 12), max(int(96 * detail), 8), bump=0.10))
    geo = np.concatenate(parts, axis=0)
    ems = np.concatenate([box((x, Y - 0.16, z), (x + 0.7, Y - 0.12, z + 0.25)) for x, z in ((-1.6, -1.5), (0.6, -1.4), (-1.5, 1.1), (0.8, 1.2), (-0.4, -0.2))], axis=0)
    tris = np.concatenate([geo, ems], axis=0)
    em = np.zeros(len(tris), bool)
    em[len(geo):] = True
    cols, emis = _colorize(tris, em, emission=(1.0, 0.92, 0.8))
    return tris.reshape(-1, 3), cols, emis


def cave_camera(lower, upper):
    """camera inside the room of cave_standin, in a corner at eye height, looking across the columns"""
    lo, hi = np.asarray(lower, np.float64), np.asarray(upper, np.float64)
    c = (lo + hi) / 2
    ext = (hi - lo).max()
    eye = c + np.array([-0.36, -0.05, 0.37]) * ext
    tgt = c + np.array([0.25, -0.08, -0.30]) * ext
    return look_at_camera(eye, tgt, 60.0, float(np.linalg.norm(tgt - eye)), 0.01)


def tunnel_standin(detail=1.0):
    """A CLOSED scene sized like the reference's published path-tracing figure (seminar slide 67: 'RT Camp scene', 4096^3, 41 M voxels): a bumpy
    tunnel of 8 x 3.4 x 3.4 units -- its length spans the grid, its cross-section 0.42 of it, which puts ~41 M surface voxels into a 4096^3 grid
    (a room that fills the grid would have ~150 M) -- with boulders along the floor, emissive lamps under the ceiling and a window in the far end wall
    through which the HDRI reaches.  Camera inside: `tunnel_camera()`; nearly every bounce hits."""
    n = max(int(1536 * detail), 24)
    m = max(int(448 * detail), 12)
    X, Y, Z = 4.0, 1.7, 1.7
    parts = [
        _panel((-X, -Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), (0, 1, 0), n, m, 0.07, 5),     # floor
        _panel((-X, Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), (0, -1, 0), n, m, 0.06, 6),     # ceiling
        _panel((-X, -Y, -Z), (2 * X, 0, 0), (0, 2 * Y, 0), (0, 0, 1), n, m, 0.05, 8),     # side walls
        _panel((-X, -Y, Z), (2 * X, 0, 0), (0, 2 * Y, 0), (0, 0, -1), n, m, 0.05, 9),
        _panel((-X, -Y, -Z), (0, 0, 2 * Z), (0, 2 * Y, 0), (1, 0, 0), m, m, 0.04, 7),     # near end wall
        _panel((X, -Y, -Z), (0, 0, 2 * Z), (0, 2 * Y, 0), (-1, 0, 0), m, m, 0.04, 10, hole=(0.35, 0.65, 0.40, 0.75)),  # far end wall with a window
    ]
    for i in range(9):  # boulders along the floor, alternating sides
        cx = -3.3 + 0.82 * i
        cz = (0.9 if i % 2 else -0.9) + 0.12 * _fsin(2.1 * i)
        rad = 0.26 + 0.07 * (0.5 + 0.5 * _fcos(1.3 * i))
        parts.append(uv_sphere((cx, -Y + rad * 0.85, cz), rad, max(int(224 * detail), 12), max(int(112 * detail), 8), bump=0.09))
    geo = np.concatenate(parts, axis=0)
    ems = np.concatenate([box((x, Y - 0.14, -0.12), (x + 0.5, Y - 0.11, 0.12)) for x in (-3.4, -2.0, -0.6, 0.8, 2.2)], axis=0)
    tris = np.concatenate([geo, ems], axis=0)
    em = np.zeros(len(tris), bool)
    em[len(geo):] = True
    cols, emis = _colorize(tris, em, emission=(1.0, 0.92, 0.8))
    return tris.reshape(-1, 3), cols, emis


def tunnel_camera(lower, upper):
    """camera inside the tunnel of tunnel_standin: near the closed end, at mid height, looking down its length.  (The grid is a cube over the scene's
    LONGEST extent with its origin at the bounding box's minimum: the tunnel lies along the grid's lower y / z edge, not around its centre.)"""
    lo, hi = np.asarray(lower, np.float64), np.asarray(upper, np.float64)
    ext = (hi - lo).max()
    eye = lo + np.array([0.09, 0.21, 0.215]) * ext
    tgt = lo + np.array([0.80, 0.19, 0.205]) * ext
    return look_at_camera(eye, tgt, 55.0, float(np.linalg.norm(tgt - eye)) * 0.5, 0.01)


SCENES = {"dragon": dragon_standin, "rtcamp": rtcamp_standin, "cave": cave_standin, "tunnel": tunnel_standin}


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
    h = np.float64(fovy_deg) * (PI / 180.0) * 0.5
    cam[12] = float(dsin(h)) / float(dcos(h))
    cam[13], cam[14] = lens_r, focus
    return cam
