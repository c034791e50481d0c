"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE -- never imported by the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps oracle/libmvrt_oracle.so (the C++ restatement, oracle/mvrt_oracle.cpp) and, when
present, oracle/_ref/libmvrt_ref.so (the two reference sources that compile as they lie).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmvrt_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libmvrt_ref.so")

MAX_FLOAT = np.float32(3.402823466e38)
PMJ_FLOATS = 2 * 4096 * 128


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    targets = ["all"] + (["ref"] if os.path.isdir(os.environ.get("MVRT_REFERENCE", "/root/reference")) else [])
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-C", _HERE] + targets)


def _load():
    if not os.path.exists(_LIB):
        build()
    return C.CDLL(_LIB)


_lib = _load()


def _p(a, ty=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _sig(name, res, args):
    f = getattr(_lib, name)
    f.restype = res
    f.argtypes = args
    return f


_vp, _i32, _i64, _u32, _u64, _f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float

_morton_encode = _sig("orc_morton_encode", _u64, [_u32, _u32, _u32])
_morton_encode_naive = _sig("orc_morton_encode_naive", _u64, [_u32, _u32, _u32])
_morton_decode = _sig("orc_morton_decode", None, [_u64, _vp])
_morton_encode_batch = _sig("orc_morton_encode_batch", None, [_vp, _i64, _vp])
_murmur = _sig("orc_murmur", _u32, [_u32, _vp, _i32])
_pcg = _sig("orc_pcg32_sequence", None, [_u64, _u64, _i32, _vp])
_uniformf = _sig("orc_uniformf", _f32, [_u32])
_pmj_table = _sig("orc_pmj_table", None, [_vp])
_rev = _sig("orc_reverse_bits", _u32, [_u32])
_nus = _sig("orc_nested_uniform_scramble", _u32, [_u32, _u32])
_scr = _sig("orc_scramble_f32", _f32, [_f32, _u32])
_s2d = _sig("orc_pmj_sample2d", None, [_vp, _u32, _u32, _u32, _vp])
_bsearch = _sig("orc_bsearch_i32", _i32, [_vp, _i32, _i32])
_voxelize = _sig("orc_voxelize", _i64, [_vp, _vp, _vp, _i64, _vp, _f32, _i32, _i32, _vp, _vp, _i64])
_merge = _sig("orc_merge_voxels", _i64, [_vp, _vp, _i64, _vp])
_build_octree = _sig("orc_build_octree", _i64, [_vp, _i64, _i32, _i32, _i32, _vp, _i64])
_scene_create = _sig("orc_scene_create", _vp, [_vp, _i64, _vp, _i64, _vp, _f32, _i32, _i32, _i32])
_scene_destroy = _sig("orc_scene_destroy", None, [_vp])
_scene_set_emission_scale = _sig("orc_scene_set_emission_scale", None, [_vp, _f32])
_scene_bounds = _sig("orc_scene_bounds", None, [_vp, _vp, _vp])
_trace = _sig("orc_trace_batch", None, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32])
_cam_mat = _sig("orc_camera_from_matrices", None, [_vp, _vp, _f32, _f32, _vp])
_cam_shoot = _sig("orc_camera_shoot", None, [_vp, _i32, _i32, _f32, _f32, _i32, _i32, _i32, _f32, _f32, _vp, _vp])
_lambert = _sig("orc_sample_lambertian", None, [_i32, _f32, _f32, _vp, _vp])
_hdri_create = _sig("orc_hdri_create", _vp, [_vp, _i32, _i32, _vp, _i32, _i32, _i32])
_hdri_destroy = _sig("orc_hdri_destroy", None, [_vp])
_hdri_scale = _sig("orc_hdri_set_scale", None, [_vp, _f32])
_hdri_sat = _sig("orc_hdri_get_sat", None, [_vp, _i32, _vp])
_hdri_is = _sig("orc_hdri_importance_sample", None, [_vp, _vp, _i32, _vp, _vp, _vp, _vp])
_hdri_near = _sig("orc_hdri_sample_nearest", None, [_vp, _vp, _i32, _vp])
_rgbe = _sig("orc_decode_rgbe", _i32, [_vp, _i64, _vp, _vp, _vp, _i64])
_render_primary = _sig("orc_render_primary", None, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32])
_render_pt = _sig("orc_render_pt", None, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _i32, _vp])
_resolve = _sig("orc_resolve", None, [_vp, _i64, _i32, _vp])
_compact = _sig("orc_compact_indices", _i64, [_vp, _i64, _vp, _vp])
_detmath = _sig("orc_detmath_eval", None, [_i32, _vp, _vp, _i64, _vp])
_sizes = _sig("orc_struct_sizes", _i32, [_vp])

NODE_DTYPE = np.dtype([("mask", "u1"), ("_pad", "u1", 3), ("children", "<u4", 8), ("psum", "<u4", 8)])
assert NODE_DTYPE.itemsize == 68


# ---- integer helpers -------------------------------------------------------------------------
def morton_encode(x, y, z):
    return int(_morton_encode(x, y, z))


def morton_encode_naive(x, y, z):
    return int(_morton_encode_naive(x, y, z))


def morton_decode(m):
    out = np.zeros(3, np.uint32)
    _morton_decode(m, _p(out))
    return tuple(int(v) for v in out)


def morton_encode_batch(xyz):
    xyz = np.ascontiguousarray(xyz, np.uint32)
    out = np.zeros(len(xyz), np.uint64)
    _morton_encode_batch(_p(xyz), len(xyz), _p(out))
    return out


def murmur(seed, words):
    w = np.ascontiguousarray(words, np.uint32)
    return int(_murmur(seed, _p(w), len(w)))


def pcg32(seed, stream, n):
    out = np.zeros(n, np.uint32)
    _pcg(seed, stream, n, _p(out))
    return out


def uniformf(x):
    return float(_uniformf(x))


def reverse_bits(v):
    return int(_rev(v))


def nested_uniform_scramble(x, seed):
    return int(_nus(x, seed))


def scramble_f32(x, seed):
    return np.float32(_scr(x, seed))


_pmj_cache = None


def pmj_table():
    global _pmj_cache
    if _pmj_cache is None:
        t = np.zeros(PMJ_FLOATS, np.float32)
        _pmj_table(_p(t))
        _pmj_cache = t
    return _pmj_cache


def pmj_sample2d(sample_idx, dim, stream, table=None):
    table = pmj_table() if table is None else table
    out = np.zeros(2, np.float32)
    _s2d(_p(table), sample_idx, dim, stream, _p(out))
    return out


def bsearch(xs, x):
    xs = np.ascontiguousarray(xs, np.int32)
    return int(_bsearch(_p(xs), len(xs), x))


# ---- scene construction ----------------------------------------------------------------------
def voxelize(tris, origin, dps, grid_res, cols=None, emis=None, six_separating=True):
    """tris: (n,9) float32.  Returns (morton u64[n], attrs u8[n,8]) with duplicates kept."""
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    cols = None if cols is None else np.ascontiguousarray(cols, np.float32).reshape(-1, 9)
    emis = None if emis is None else np.ascontiguousarray(emis, np.float32).reshape(-1, 9)
    origin = np.ascontiguousarray(origin, np.float32)
    n = _voxelize(_p(tris), _p(cols), _p(emis), len(tris), _p(origin), dps, grid_res, int(six_separating), None, None, 0)
    morton = np.zeros(n, np.uint64)
    attrs = np.zeros((n, 8), np.uint8)
    _voxelize(_p(tris), _p(cols), _p(emis), len(tris), _p(origin), dps, grid_res, int(six_separating), _p(morton), _p(attrs), n)
    return morton, attrs


def merge_voxels(morton, attrs):
    morton = np.array(morton, np.uint64, copy=True)
    attrs = np.array(attrs, np.uint8, copy=True).reshape(-1, 8)
    he = C.c_int(0)
    n = _merge(_p(morton), _p(attrs), len(morton), C.byref(he))
    return morton[:n].copy(), attrs[:n].copy(), int(he.value)


def build_octree(morton, grid_res, dag=True, embed=True):
    morton = np.ascontiguousarray(morton, np.uint64)
    n = _build_octree(_p(morton), len(morton), grid_res, int(dag), int(embed), None, 0)
    nodes = np.zeros(n, NODE_DTYPE)
    _build_octree(_p(morton), len(morton), grid_res, int(dag), int(embed), _p(nodes), n)
    return nodes


class Scene:
    def __init__(self, nodes, attrs, origin, dps, grid_res, has_emission=0, embedded=True):
        self.nodes = np.ascontiguousarray(nodes)
        assert self.nodes.dtype.itemsize == 68
        self.attrs = np.ascontiguousarray(attrs, np.uint8).reshape(-1, 8)
        self.origin = np.ascontiguousarray(origin, np.float32)
        self.dps = float(np.float32(dps))
        self.grid_res = int(grid_res)
        self.has_emission = int(has_emission)
        self.embedded = bool(embedded)
        self._h = _scene_create(_p(self.nodes), len(self.nodes), _p(self.attrs), len(self.attrs), _p(self.origin), self.dps, self.grid_res, self.has_emission,
                                int(embedded))

    def __del__(self):
        if getattr(self, "_h", None):
            _scene_destroy(self._h)
            self._h = None

    def set_emission_scale(self, s):
        _scene_set_emission_scale(self._h, float(s))

    def bounds(self):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        _scene_bounds(self._h, _p(lo), _p(hi))
        return lo, hi

    def trace(self, ro, rd, is_shadow=None, threads=1, want_descents=False):
        ro = np.ascontiguousarray(ro, np.float32).reshape(-1, 3)
        rd = np.ascontiguousarray(rd, np.float32).reshape(-1, 3)
        n = len(ro)
        sh = None if is_shadow is None else np.ascontiguousarray(is_shadow, np.uint8)
        t = np.zeros(n, np.float32)
        nm = np.zeros(n, np.int32)
        vi = np.zeros(n, np.uint32)
        de = np.zeros(n, np.uint32) if want_descents else None
        msp = C.c_int(0)
        _trace(self._h, n, _p(ro), _p(rd), _p(sh), _p(t), _p(nm), _p(vi), _p(de), C.byref(msp), threads)
        out = {"t": t, "nMajor": nm, "vIndex": vi, "maxSp": int(msp.value)}
        if want_descents:
            out["descents"] = de
        return out

    def render_primary(self, cam, W, H, show_vertex_color=False, threads=1):
        cam = np.ascontiguousarray(cam, np.float32)
        n = W * H
        rgba = np.zeros((n, 4), np.uint8)
        t = np.zeros(n, np.float32)
        nm = np.zeros(n, np.int32)
        vi = np.zeros(n, np.uint32)
        de = np.zeros(n, np.uint32)
        _render_primary(self._h, _p(cam), W, H, int(show_vertex_color), _p(rgba), _p(t), _p(nm), _p(vi), _p(de), threads)
        return {"rgba": rgba, "t": t, "nMajor": nm, "vIndex": vi, "descents": de}

    def render_pt(self, hdri, cam, W, H, iteration, math_mode=1, fb=None, pixel_begin=0, pixel_end=-1, want_samples=False, threads=1, pmj=None, path_hits=None):
        """path_hits: optional uint8 array ((pixel_end - pixel_begin) * 16) that receives, per sample, how many of the path's own rays hit"""
        cam = np.ascontiguousarray(cam, np.float32)
        pmj = pmj_table() if pmj is None else pmj
        if fb is None:
            fb = np.zeros((W * H, 4), np.float32)
        pe = W * H if pixel_end < 0 else pixel_end
        # the C side writes fb[pixel_begin:pe], 16 samples per pixel and one byte per sample without any check of its own (r02: a band that ran
        # past the frame wrote past a caller's buffer and took the test process down)
        assert 0 <= pixel_begin <= pe <= W * H, "render_pt: pixel range [%d, %d) outside the %dx%d frame" % (pixel_begin, pe, W, H)
        assert fb.dtype == np.float32 and fb.flags.c_contiguous and fb.size >= W * H * 4, "render_pt: fb must be a contiguous float32 array of W*H*4"
        assert cam.size == 15 and pmj.dtype == np.float32 and pmj.size == PMJ_FLOATS
        sl = np.zeros(((pe - pixel_begin) * 16, 3), np.float32) if want_samples else None
        cnt = np.zeros(6, np.uint64)
        if path_hits is not None:
            assert path_hits.dtype == np.uint8 and path_hits.flags.c_contiguous and path_hits.size == (pe - pixel_begin) * 16
        _render_pt(self._h, hdri._h, _p(pmj), _p(cam), W, H, iteration, math_mode, _p(fb), pixel_begin, pe, _p(sl), _p(cnt), threads, _p(path_hits))
        counters = dict(zip(["rays", "shadowRays", "descents", "shadowDescents", "hits", "samples"], (int(v) for v in cnt)))
        return fb, sl, counters


def build_scene_from_triangles(tris, grid_res, cols=None, emis=None, origin=None, dps=None, dag=True, embed=True):
    """voxRT.cpp:188-270: origin = bbox min, dps = max extent / gridRes, voxelize, merge, DAG build."""
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    v = tris.reshape(-1, 3)
    if origin is None:
        origin = v.min(0)
    if dps is None:
        size = v.max(0) - v.min(0)
        dps = np.float32(np.float32(size.max()) / np.float32(grid_res))
    m, a = voxelize(tris, origin, dps, grid_res, cols, emis)
    dumped = len(m)
    m, a, he = merge_voxels(m, a)
    nodes = build_octree(m, grid_res, dag=dag, embed=embed)
    sc = Scene(nodes, a, origin, dps, grid_res, he, embedded=embed)
    sc.morton = m
    sc.dumped = dumped
    return sc


# ---- camera ------------------------------------------------------------------------------------
def camera_from_matrices(view, proj, focus=1.0, lens_r=0.0):
    """view/proj: 4x4 column-major (glm) as 16 floats each."""
    view = np.ascontiguousarray(view, np.float32).reshape(16)
    proj = np.ascontiguousarray(proj, np.float32).reshape(16)
    cam = np.zeros(15, np.float32)
    _cam_mat(_p(view), _p(proj), focus, lens_r, _p(cam))
    return cam


def camera_shoot(cam, x, y, xo, yo, W, H, thin_lens=False, u0=0.0, u1=0.0):
    cam = np.ascontiguousarray(cam, np.float32)
    ro, rd = np.zeros(3, np.float32), np.zeros(3, np.float32)
    _cam_shoot(_p(cam), x, y, xo, yo, W, H, int(thin_lens), u0, u1, _p(ro), _p(rd))
    return ro, rd


def sample_lambertian(a, b, n, math_mode=0):
    n = np.ascontiguousarray(n, np.float32)
    out = np.zeros(3, np.float32)
    _lambert(math_mode, a, b, _p(n), _p(out))
    return out


# ---- HDRI ---------------------------------------------------------------------------------------
def decode_rgbe(data):
    buf = np.frombuffer(data, np.uint8)
    w, h = C.c_int(0), C.c_int(0)
    rc = _rgbe(_p(buf), len(buf), C.byref(w), C.byref(h), None, 0)
    if rc:
        raise ValueError("bad .hdr header (%d)" % rc)
    out = np.zeros((h.value * w.value, 4), np.float32)
    rc = _rgbe(_p(buf), len(buf), C.byref(w), C.byref(h), _p(out), len(out))
    if rc:
        raise ValueError("bad .hdr body (%d)" % rc)
    return out, w.value, h.value


class HDRI:
    def __init__(self, rgba, w, h, primary=None, wp=0, hp=0, math_mode=0):
        self.rgba = np.ascontiguousarray(rgba, np.float32)
        self.w, self.h = w, h
        self.primary = None if primary is None else np.ascontiguousarray(primary, np.float32)
        self._h = _hdri_create(_p(self.rgba), w, h, _p(self.primary), wp, hp, math_mode)

    def __del__(self):
        if getattr(self, "_h", None):
            _hdri_destroy(self._h)
            self._h = None

    def set_scale(self, s):
        _hdri_scale(self._h, s)

    def sat(self, which):
        out = np.zeros(self.w * self.h, np.uint32)
        _hdri_sat(self._h, which, _p(out))
        return out

    def importance_sample(self, n, u, axis_aligned=True):
        n = np.ascontiguousarray(n, np.float32)
        u = np.ascontiguousarray(u, np.float32)
        d, L = np.zeros(3, np.float32), np.zeros(3, np.float32)
        pdf = C.c_float(0)
        _hdri_is(self._h, _p(n), int(axis_aligned), _p(u), _p(d), _p(L), C.byref(pdf))
        return d, L, np.float32(pdf.value)

    def sample_nearest(self, d, is_primary):
        d = np.ascontiguousarray(d, np.float32)
        out = np.zeros(3, np.float32)
        _hdri_near(self._h, _p(d), int(is_primary), _p(out))
        return out


def resolve(fb, math_mode=1):
    fb = np.ascontiguousarray(fb, np.float32).reshape(-1, 4)
    out = np.zeros((len(fb), 4), np.uint8)
    _resolve(_p(fb), len(fb), math_mode, _p(out))
    return out


def compact_indices(keep):
    keep = np.ascontiguousarray(keep, np.uint8)
    dst = np.zeros(len(keep), np.uint32)
    src = np.zeros(len(keep), np.uint32)
    k = _compact(_p(keep), len(keep), _p(dst), _p(src))
    return dst, src[:k].copy()


def detmath(which, x, y=None):
    names = {"sin": 0, "cos": 1, "atan2": 2, "pow": 3, "log": 4, "exp": 5}
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, np.float32)
    out = np.zeros_like(x)
    _detmath(names[which], _p(x), _p(y), len(x), _p(out))
    return out


def struct_sizes():
    out = np.zeros(8, np.int32)
    n = _sizes(_p(out))
    return dict(zip(["OctreeNode", "StackElement", "OctreeTask", "VoxelAttirb", "CameraPinhole"], (int(v) for v in out[:n])))


# ---- oracle/_ref: the reference's own morton.hpp + smhasher MurmurHash3, compiled as they lie ---
def load_ref():
    if not os.path.exists(_REF):
        return None
    r = C.CDLL(_REF)
    r.ref_morton_batch.argtypes = [_vp, _i64, _vp, _vp, _vp]
    r.ref_murmur3_x86_32.restype = _u32
    r.ref_murmur3_x86_32.argtypes = [_vp, _i32, _u32]
    for nm in ("ref_morton_decode_naive", "ref_morton_decode_pext", "ref_morton_decode_magicbits"):
        getattr(r, nm).argtypes = [_u64, _vp]
    return r
