"""massivevoxelraytracing_amd -- MI355X-native SVO path tracer behind the reference's host API.

Python mirror of the reference's two host structs for the GPU hot path:

* ``IntersectorOctreeGPU``  (reference IntersectorOctreeGPU.hpp:21-275)
* ``PathTracer``            (reference PathTracer.hpp:14-170)

Both are thin ctypes views over the C-ABI library ``libmvrt_hip.so`` (include/mvrt.h) -- the same
entry points a C++ caller binds through include/mvrt/*.hpp.  There is NO CPU fallback: importing
``lib()`` fails loudly if the HIP library is missing or cannot be loaded.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVRT_LIB", os.path.join(_HERE, "libmvrt_hip.so"))  # MVRT_LIB: A/B builds of the same library
MAX_FLOAT = np.float32(3.402823466e38)

_vp, _i32, _u32, _u64, _f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float


class MvrtError(RuntimeError):
    pass


class SvoInfo(C.Structure):
    _fields_ = [("numberOfNodes", _u32), ("numberOfVoxels", _u32), ("lower", _f32 * 3), ("upper", _f32 * 3), ("dps", _f32), ("emissionScale", _f32),
                ("hasEmission", _u32), ("embeddedMask", _u32), ("gridRes", _u32), ("levels", _u32), ("totalDumpedVoxels", _u64), ("flavour", _u32), ("reserved", _u32)]


class PtStats(C.Structure):
    _fields_ = [("samples", _u64), ("rays", _u64), ("shadowRays", _u64), ("descents", _u64), ("shadowDescents", _u64), ("hits", _u64), ("traceLaunches", _u64),
                ("traceKernelMs", C.c_double), ("shadeKernelMs", C.c_double), ("totalKernelMs", C.c_double)]


# every symbol include/mvrt.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mvrt_last_error": (C.c_char_p, []),
    "mvrt_device_count": (_i32, [_vp]),
    "mvrt_set_device": (_i32, [_i32]),
    "mvrt_device_name": (_i32, [_vp, _i32]),
    "mvrt_stream_create": (_i32, [_vp]),
    "mvrt_stream_destroy": (_i32, [_vp]),
    "mvrt_stream_synchronize": (_i32, [_vp]),
    "mvrt_device_synchronize": (_i32, []),
    "mvrt_malloc": (_i32, [_vp, _u64]),
    "mvrt_free": (_i32, [_vp]),
    "mvrt_memcpy_h2d": (_i32, [_vp, _vp, _u64, _vp]),
    "mvrt_memcpy_d2h": (_i32, [_vp, _vp, _u64, _vp]),
    "mvrt_memcpy_d2d": (_i32, [_vp, _vp, _u64, _vp]),
    "mvrt_svo_create": (_i32, [_vp]),
    "mvrt_svo_destroy": (_i32, [_vp]),
    "mvrt_svo_build": (_i32, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _f32, _i32]),
    "mvrt_svo_build_ex": (_i32, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _f32, _i32, _i32]),
    "mvrt_svo_build_synthetic": (_i32, [_vp, _i32, _u64, _u64, _vp, _f32, _i32, _vp]),
    "mvrt_svo_upload": (_i32, [_vp, _vp, _u32, _vp, _u32, _vp, _f32, _i32, _i32, _i32, _vp]),
    "mvrt_svo_get_info": (_i32, [_vp, _vp]),
    "mvrt_svo_set_emission_scale": (_i32, [_vp, _f32]),
    "mvrt_svo_traversal_bytes": (_u64, [_vp]),
    "mvrt_svo_node_buffer_dev": (_vp, [_vp]),
    "mvrt_svo_attribute_buffer_dev": (_vp, [_vp]),
    "mvrt_pt_download_pmj": (_i32, [_vp, _vp]),
    "mvrt_svo_download": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "mvrt_trace_batch": (_i32, [_vp, _u64] + [_vp] * 11 + [_vp]),
    "mvrt_trace_batch_hinted": (_i32, [_vp, _u64] + [_vp] * 12 + [_vp]),
    "mvrt_trace_batch_host": (_i32, [_vp, _u64] + [_vp] * 7),
    "mvrt_render_primary": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvrt_camera_from_matrices": (_i32, [_vp, _vp, _f32, _f32, _vp]),
    "mvrt_compact_indices": (_i32, [_vp, _u64, _vp, _vp, _vp]),
    "mvrt_pt_create": (_i32, [_vp]),
    "mvrt_pt_destroy": (_i32, [_vp]),
    "mvrt_pt_setup": (_i32, [_vp, _vp]),
    "mvrt_pt_resize_framebuffer_if_needed": (_i32, [_vp, _vp, _i32, _i32]),
    "mvrt_pt_clear_framebuffer": (_i32, [_vp, _vp]),
    "mvrt_pt_load_hdri": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32]),
    "mvrt_pt_load_hdri_file": (_i32, [_vp, _vp, C.c_char_p, C.c_char_p]),
    "mvrt_pt_download_hdri_sat": (_i32, [_vp, _i32, _vp]),
    "mvrt_rgbe_read_file": (_i32, [C.c_char_p, _vp, _u64, _vp, _vp]),
    "mvrt_pt_set_hdri_scale": (_i32, [_vp, _f32]),
    "mvrt_pt_update_scene": (_i32, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _f32, _i32]),
    "mvrt_pt_intersector": (_vp, [_vp]),
    "mvrt_pt_step": (_i32, [_vp, _vp, _vp]),
    "mvrt_pt_step_matrices": (_i32, [_vp, _vp, _vp, _vp, _f32, _f32]),
    "mvrt_pt_set_pipeline_depth": (_i32, [_vp, _i32]),
    "mvrt_pt_set_origin_hints": (_i32, [_vp, _i32]),
    "mvrt_pt_set_batch_steps": (_i32, [_vp, _i32]),
    "mvrt_pt_set_split_small_passes": (_i32, [_vp, _i32]),
    "mvrt_pt_join": (_i32, [_vp, _vp]),
    "mvrt_pt_resolve": (_i32, [_vp, _vp]),
    "mvrt_pt_to_image_async": (_i32, [_vp, _vp, _vp]),
    "mvrt_pt_get_steps": (_i32, [_vp]),
    "mvrt_pt_get_number_of_voxels": (_u64, [_vp]),
    "mvrt_pt_get_octree_bytes": (_u64, [_vp]),
    "mvrt_pt_read_framebuffer": (_i32, [_vp, _vp, _vp]),
    "mvrt_pt_framebuffer_dev": (_vp, [_vp]),
    "mvrt_pt_framebuffer_u8_dev": (_vp, [_vp]),
    "mvrt_pt_set_tile": (_i32, [_vp, _i32, _i32]),
    "mvrt_pt_owned_pixels": (_u64, [_vp]),
    "mvrt_pt_assemble_tiles": (_i32, [_vp, _i32, _u64, _i32, _i32, _vp, _vp]),
    "mvrt_resolve_buffer": (_i32, [_vp, _u64, _vp, _vp]),
    "mvrt_pt_sample_radiance_dev": (_vp, [_vp]),
    "mvrt_pt_read_sample_radiance": (_i32, [_vp, _vp, _u64]),
    "mvrt_pt_set_debug_capture": (_i32, [_vp, _i32]),
    "mvrt_pt_read_debug_stage": (_i32, [_vp, _i32, _vp, _u64, _vp]),
    "mvrt_pt_set_test_free_bytes": (_i32, [_vp, _u64]),
    "mvrt_pt_set_profiling": (_i32, [_vp, _i32]),
    "mvrt_pt_reset_stats": (_i32, [_vp]),
    "mvrt_pt_get_stats": (_i32, [_vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load libmvrt_hip.so (once).  Raises MvrtError if the HIP library is absent -- no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MvrtError("libmvrt_hip.so is not built (%s); run `python -m massivevoxelraytracing_amd.build`. "
                            "There is no CPU fallback for the GPU path." % LIB_PATH)
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:
            raise MvrtError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise MvrtError(lib().mvrt_last_error().decode("utf-8", "replace"))


def _hp(a):
    """host pointer of a numpy array (or None)"""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    _check(lib().mvrt_device_count(C.byref(n)))
    return n.value


def set_device(i):
    _check(lib().mvrt_set_device(int(i)))


def device_name():
    buf = C.create_string_buffer(256)
    _check(lib().mvrt_device_name(buf, 256))
    return buf.value.decode()


def synchronize():
    _check(lib().mvrt_device_synchronize())


class DeviceArray:
    """A typed device buffer owned through mvrt_malloc/mvrt_free (hipUtil.hpp:48-74 'Buffer')."""

    def __init__(self, shape, dtype):
        self.shape = (shape,) if np.isscalar(shape) else tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p(0)
        _check(lib().mvrt_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        if d.nbytes:
            _check(lib().mvrt_memcpy_h2d(d.ptr, _hp(a), d.nbytes, None))
        return d

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _check(lib().mvrt_memcpy_d2h(_hp(out), self.ptr, self.nbytes, None))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            lib().mvrt_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _dev_ptr(x):
    """Accept DeviceArray, raw int pointers or anything with data_ptr() (torch tensors)."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


def camera_from_matrices(view, proj, focus=1.0, lens_r=0.0):
    """CameraPinhole::initFromPerspective (renderCommon.hpp:21-35); matrices column-major."""
    view = np.ascontiguousarray(view, np.float32).reshape(16)
    proj = np.ascontiguousarray(proj, np.float32).reshape(16)
    cam = np.zeros(15, np.float32)
    _check(lib().mvrt_camera_from_matrices(_hp(view), _hp(proj), focus, lens_r, _hp(cam)))
    return cam


class IntersectorOctreeGPU:
    """reference IntersectorOctreeGPU.hpp:21-275 (host side) + batch form of its device methods."""

    def __init__(self, _borrowed=None):
        self._own = _borrowed is None
        if self._own:
            h = C.c_void_p(0)
            _check(lib().mvrt_svo_create(C.byref(h)))
            self._h = h.value
        else:
            self._h = _borrowed

    def cleanUp(self):
        if self._own and getattr(self, "_h", None):
            lib().mvrt_svo_destroy(self._h)
            self._h = None

    __del__ = cleanUp

    BUILD_NO_DAG = 1
    BUILD_NO_EMBEDDED_MASK = 2
    BUILD_CONSERVATIVE = 4

    def build_synthetic(self, gridRes, n_random_voxels, seed, origin=(0.0, 0.0, 0.0), dps=None, flags=0, stream=None):
        """seeded random-voxel octree built on the GPU (HBM-bound stress, mvrt_svo_build_synthetic)"""
        o = np.ascontiguousarray(origin, np.float32)
        dps = np.float32(1.0 / gridRes) if dps is None else np.float32(dps)
        _check(lib().mvrt_svo_build_synthetic(self._h, int(gridRes), int(n_random_voxels), int(seed), _hp(o), float(dps), int(flags), stream))

    def build(self, vertices, vcolors, vemissions, stream, origin, dps, gridRes, flags=0):
        """IntersectorOctreeGPU::build(vertices, vcolors, vemissions, Shader*, stream, origin, dps, gridRes)
        (:40-47; the Shader* argument has no counterpart -- kernels are precompiled)."""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        c = None if vcolors is None else np.ascontiguousarray(vcolors, np.float32).reshape(-1, 3)
        e = None if vemissions is None else np.ascontiguousarray(vemissions, np.float32).reshape(-1, 3)
        o = np.ascontiguousarray(origin, np.float32)
        _check(lib().mvrt_svo_build_ex(self._h, _hp(v), _hp(c), _hp(e), len(v), stream, _hp(o), float(np.float32(dps)), int(gridRes), int(flags)))

    def upload(self, nodes68, attribs, origin, dps, gridRes, hasEmission=0, embeddedMask=True, stream=None):
        nodes68 = np.ascontiguousarray(nodes68)
        assert nodes68.dtype.itemsize == 68 or nodes68.dtype == np.uint8
        n_nodes = nodes68.nbytes // 68
        attribs = np.ascontiguousarray(attribs, np.uint8).reshape(-1, 8)
        o = np.ascontiguousarray(origin, np.float32)
        _check(lib().mvrt_svo_upload(self._h, _hp(nodes68), n_nodes, _hp(attribs), len(attribs), _hp(o), float(np.float32(dps)), int(gridRes), int(hasEmission),
                                     int(embeddedMask), stream))

    def info(self):
        i = SvoInfo()
        _check(lib().mvrt_svo_get_info(self._h, C.byref(i)))
        return i

    # reference public fields (:265-274)
    m_numberOfNodes = property(lambda s: s.info().numberOfNodes)
    m_numberOfVoxels = property(lambda s: s.info().numberOfVoxels)
    m_lower = property(lambda s: np.array(s.info().lower[:], np.float32))
    m_upper = property(lambda s: np.array(s.info().upper[:], np.float32))
    m_dps = property(lambda s: s.info().dps)
    m_hasEmission = property(lambda s: s.info().hasEmission)
    m_nodeBuffer = property(lambda s: lib().mvrt_svo_node_buffer_dev(s._h))  # device pointers (:265-266)
    m_vAttributeBuffer = property(lambda s: lib().mvrt_svo_attribute_buffer_dev(s._h))

    def traversal_bytes(self):
        return lib().mvrt_svo_traversal_bytes(self._h)

    def hasEmission(self):
        return bool(self.info().hasEmission)

    def set_emission_scale(self, s):
        _check(lib().mvrt_svo_set_emission_scale(self._h, s))

    def download(self, want_morton=False, stream=None):
        from numpy import dtype
        i = self.info()
        nodes = np.zeros(i.numberOfNodes * 68, np.uint8)
        attrs = np.zeros((i.numberOfVoxels, 8), np.uint8)
        morton = np.zeros(i.numberOfVoxels, np.uint64) if want_morton else None
        _check(lib().mvrt_svo_download(self._h, _hp(nodes), _hp(attrs), _hp(morton), stream))
        return nodes, attrs, morton

    def intersect(self, ro, rd, isShadowRay=None, want_descents=False):
        """Batch IntersectorOctreeGPU::intersect (:243-251) on packed host arrays (n,3)."""
        ro = np.ascontiguousarray(ro, np.float32).reshape(-1, 3)
        rd = np.ascontiguousarray(rd, np.float32).reshape(-1, 3)
        n = len(ro)
        sh = None if isShadowRay is None else np.ascontiguousarray(isShadowRay, np.uint8)
        t = np.zeros(n, np.float32)
        nm = np.zeros(n, np.int32)
        vi = np.zeros(n, np.uint32)
        de = np.zeros(n, np.uint32)
        _check(lib().mvrt_trace_batch_host(self._h, n, _hp(ro), _hp(rd), _hp(sh), _hp(t), _hp(nm), _hp(vi), _hp(de)))
        out = {"t": t, "nMajor": nm, "vIndex": vi}
        if want_descents:
            out["descents"] = de
        return out

    def intersect_device(self, n, rox, roy, roz, rdx, rdy, rdz, isShadow, t, nMajor, vIndex, descents=None, stream=None):
        _check(lib().mvrt_trace_batch(self._h, n, *[_dev_ptr(a) for a in (rox, roy, roz, rdx, rdy, rdz, isShadow, t, nMajor, vIndex, descents)], stream))

    def intersect_hinted(self, ro, rd, origin_voxel_morton, isShadowRay=None):
        """mvrt_trace_batch_hinted on packed host arrays: per ray the Morton code of an EXISTING voxel (or 2^64-1 = no hint) to start below the root from"""
        ro = np.ascontiguousarray(ro, np.float32).reshape(-1, 3)
        rd = np.ascontiguousarray(rd, np.float32).reshape(-1, 3)
        n = len(ro)
        dev = [DeviceArray.from_host(np.ascontiguousarray(a)) for a in (ro[:, 0], ro[:, 1], ro[:, 2], rd[:, 0], rd[:, 1], rd[:, 2])]
        sh = None if isShadowRay is None else DeviceArray.from_host(np.ascontiguousarray(isShadowRay, np.uint8))
        hint = DeviceArray.from_host(np.ascontiguousarray(origin_voxel_morton, np.uint64))
        t, nm, vi, de = DeviceArray(n, np.float32), DeviceArray(n, np.int32), DeviceArray(n, np.uint32), DeviceArray(n, np.uint32)
        _check(lib().mvrt_trace_batch_hinted(self._h, n, *[_dev_ptr(a) for a in dev], _dev_ptr(sh), hint.ptr, t.ptr, nm.ptr, vi.ptr, de.ptr, None))
        synchronize()
        return {"t": t.to_host(), "nMajor": nm.to_host(), "vIndex": vi.to_host(), "descents": de.to_host()}

    def render(self, camera, width, height, showVertexColor=False, want_hits=True, stream=None):
        """the `render` kernel launch of voxRTGPU.cpp:191-203; returns host arrays"""
        cam = np.ascontiguousarray(camera, np.float32)
        n = width * height
        rgba = DeviceArray((n, 4), np.uint8)
        t = DeviceArray(n, np.float32) if want_hits else None
        nm = DeviceArray(n, np.int32) if want_hits else None
        vi = DeviceArray(n, np.uint32) if want_hits else None
        de = DeviceArray(n, np.uint32) if want_hits else None
        _check(lib().mvrt_render_primary(self._h, _hp(cam), width, height, int(showVertexColor), rgba.ptr, *[_dev_ptr(a) for a in (t, nm, vi, de)], stream))
        _check(lib().mvrt_stream_synchronize(stream))
        out = {"rgba": rgba.to_host()}
        if want_hits:
            out.update(t=t.to_host(), nMajor=nm.to_host(), vIndex=vi.to_host(), descents=de.to_host())
        return out

    def render_device(self, camera, width, height, showVertexColor, rgba_dev, stream=None):
        cam = np.ascontiguousarray(camera, np.float32)
        _check(lib().mvrt_render_primary(self._h, _hp(cam), width, height, int(showVertexColor), _dev_ptr(rgba_dev), None, None, None, None, stream))


def read_rgbe_file(path):
    """host-only: the .hdr decoder behind PathTracer.loadHDRI -> (rgba float32 (h*w, 4), w, h)"""
    w, h = C.c_int(0), C.c_int(0)
    _check(lib().mvrt_rgbe_read_file(path.encode(), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((w.value * h.value, 4), np.float32)
    _check(lib().mvrt_rgbe_read_file(path.encode(), _hp(out), len(out), C.byref(w), C.byref(h)))
    return out, w.value, h.value


def compact_indices(keep):
    """Stable compaction indices of host flags through the device path (StreamCompaction semantics)."""
    keep = np.ascontiguousarray(keep, np.uint8)
    n = len(keep)
    d_keep = DeviceArray.from_host(keep)
    d_dst = DeviceArray(max(n, 1), np.uint32)
    d_kept = DeviceArray(1, np.uint32)
    _check(lib().mvrt_compact_indices(d_keep.ptr, n, d_dst.ptr, d_kept.ptr, None))
    synchronize()
    return d_dst.to_host()[:n], int(d_kept.to_host()[0])


class PathTracer:
    """reference PathTracer.hpp:14-170.  Same method names and call order; prlib types are replaced by
    plain arrays (camera = 15 floats or view/proj matrices; images = numpy arrays)."""

    def __init__(self):
        h = C.c_void_p(0)
        _check(lib().mvrt_pt_create(C.byref(h)))
        self._h = h.value
        self.m_intersectorOctreeGPU = IntersectorOctreeGPU(_borrowed=lib().mvrt_pt_intersector(self._h))

    def cleanUp(self):
        if getattr(self, "_h", None):
            lib().mvrt_pt_destroy(self._h)
            self._h = None

    __del__ = cleanUp

    def setup(self, stream=None, kernel=None, includeDir=None, isNvidia=False):
        """PathTracer::setup(stream, kernel, includeDir, isNvidia) (:43-69); the last three are ignored."""
        _check(lib().mvrt_pt_setup(self._h, stream))

    def pmj_table(self):
        """PMJSampler::m_samples as the host generated it (128 x 4096 float2)"""
        out = np.zeros(2 * 4096 * 128, np.float32)
        _check(lib().mvrt_pt_download_pmj(self._h, _hp(out)))
        return out

    def set_tile(self, tile_index, tile_count):
        _check(lib().mvrt_pt_set_tile(self._h, tile_index, tile_count))

    def resizeFrameBufferIfNeeded(self, stream, width, height):
        _check(lib().mvrt_pt_resize_framebuffer_if_needed(self._h, stream, width, height))
        self.m_width, self.m_height = width, height

    def clearFrameBuffer(self, stream=None):
        _check(lib().mvrt_pt_clear_framebuffer(self._h, stream))

    def loadHDRI(self, stream, file, filePrimary=None):
        _check(lib().mvrt_pt_load_hdri_file(self._h, stream, file.encode(), None if filePrimary is None else filePrimary.encode()))

    def loadHDRIPixels(self, stream, rgba, w, h, rgbaPrimary=None, wp=0, hp=0):
        rgba = np.ascontiguousarray(rgba, np.float32)
        prim = None if rgbaPrimary is None else np.ascontiguousarray(rgbaPrimary, np.float32)
        _check(lib().mvrt_pt_load_hdri(self._h, stream, _hp(rgba), w, h, _hp(prim), wp, hp))

    def hdri_sat(self, which, w, h):
        out = np.zeros(w * h, np.uint32)
        _check(lib().mvrt_pt_download_hdri_sat(self._h, which, _hp(out)))
        return out

    def set_hdri_scale(self, s):
        _check(lib().mvrt_pt_set_hdri_scale(self._h, s))

    def updateScene(self, vertices, vcolors, vemissions, stream, origin, dps, gridRes):
        self.m_intersectorOctreeGPU.build(vertices, vcolors, vemissions, stream, origin, dps, gridRes)

    def step(self, stream, camera, focus=None, lensR=None):
        """PathTracer::step(stream, camera, focus, lensR) (:150-169).  `camera` is either the 15 CameraPinhole
        floats (focus/lensR already inside) or a (view, proj) pair of column-major 4x4 matrices."""
        if isinstance(camera, (tuple, list)) and len(camera) == 2:
            view = np.ascontiguousarray(camera[0], np.float32).reshape(16)
            proj = np.ascontiguousarray(camera[1], np.float32).reshape(16)
            _check(lib().mvrt_pt_step_matrices(self._h, stream, _hp(view), _hp(proj), focus, lensR))
        else:
            cam = np.array(camera, np.float32, copy=True)
            if focus is not None:
                cam[14] = focus
            if lensR is not None:
                cam[13] = lensR
            _check(lib().mvrt_pt_step(self._h, stream, _hp(cam)))

    def set_batch_steps(self, n):
        _check(lib().mvrt_pt_set_batch_steps(self._h, n))

    def set_split_small_passes(self, enable):
        _check(lib().mvrt_pt_set_split_small_passes(self._h, 1 if enable else 0))

    def set_origin_hints(self, enable):
        _check(lib().mvrt_pt_set_origin_hints(self._h, 1 if enable else 0))

    def set_pipeline_depth(self, depth):
        _check(lib().mvrt_pt_set_pipeline_depth(self._h, depth))

    def join(self, stream=None):
        """make `stream` wait for the steps still in flight on the internal streams"""
        _check(lib().mvrt_pt_join(self._h, stream))

    def resolve(self, stream=None):
        _check(lib().mvrt_pt_resolve(self._h, stream))

    def toImageAsync(self, stream=None, output=None):
        n = self.owned_pixels()
        out = np.zeros((n, 4), np.uint8) if output is None else output
        _check(lib().mvrt_pt_to_image_async(self._h, stream, _hp(out)))
        return out

    def getSteps(self):
        return lib().mvrt_pt_get_steps(self._h)

    def getNumberOfVoxels(self):
        return lib().mvrt_pt_get_number_of_voxels(self._h)

    def getOctreeBytes(self):
        return lib().mvrt_pt_get_octree_bytes(self._h)

    def owned_pixels(self):
        return lib().mvrt_pt_owned_pixels(self._h)

    def read_framebuffer(self, stream=None):
        out = np.zeros((self.owned_pixels(), 4), np.float32)
        _check(lib().mvrt_pt_read_framebuffer(self._h, stream, _hp(out)))
        return out

    def framebuffer_dev(self):
        return lib().mvrt_pt_framebuffer_dev(self._h)

    def sample_radiance(self, n_samples=None):
        """per-sample radiance of the last pass: (n, 3) host array (debug / parity)"""
        n = self.owned_pixels() * 16 if n_samples is None else n_samples
        out = np.zeros((3, n), np.float32)
        _check(lib().mvrt_pt_read_sample_radiance(self._h, _hp(out), n))
        return out.T.copy()

    def set_debug_capture(self, on):
        _check(lib().mvrt_pt_set_debug_capture(self._h, int(on)))

    def debug_stage_survivors(self, stage, capacity=None):
        """sample ids of the paths that survived shade stage `stage` of the last pass, in compacted-slot order (needs set_debug_capture)"""
        cap = self.owned_pixels() * 16 * 8 if capacity is None else capacity
        out = np.zeros(cap, np.uint32)
        n = C.c_uint32(0)
        _check(lib().mvrt_pt_read_debug_stage(self._h, stage, _hp(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def set_test_free_bytes(self, nbytes):
        """failure-path tests: budget the path state against `nbytes` of free HBM (0 = what the device reports)"""
        _check(lib().mvrt_pt_set_test_free_bytes(self._h, int(nbytes)))

    def set_profiling(self, on):
        _check(lib().mvrt_pt_set_profiling(self._h, int(on)))

    def reset_stats(self):
        _check(lib().mvrt_pt_reset_stats(self._h))

    def stats(self, stream=None):
        s = PtStats()
        _check(lib().mvrt_pt_get_stats(self._h, stream, C.byref(s)))
        return {k: getattr(s, k) for k, _ in PtStats._fields_}


def memcpy_d2d(dst_dev, src_dev, nbytes, stream=None):
    _check(lib().mvrt_memcpy_d2d(_dev_ptr(dst_dev), _dev_ptr(src_dev), nbytes, stream))


def assemble_tiles(gathered_dev, tile_count, rank_stride_pixels, width, height, frame_dev, stream=None):
    _check(lib().mvrt_pt_assemble_tiles(_dev_ptr(gathered_dev), tile_count, rank_stride_pixels, width, height, _dev_ptr(frame_dev), stream))


def resolve_buffer(rgba_f32_dev, n_pixels, rgba_u8_dev, stream=None):
    _check(lib().mvrt_resolve_buffer(_dev_ptr(rgba_f32_dev), n_pixels, _dev_ptr(rgba_u8_dev), stream))
