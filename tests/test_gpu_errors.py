"""Error behaviour of the boundary on the GPU box (SURVEY.md 8b "Errors": the reference aborts or asserts -- hipUtil.hpp:141-179,
IntersectorOctreeGPU.hpp:48-51,231; the C functions return a status and keep a message, the handle stays usable)."""
import numpy as np
import pytest

from common import bunny_tris, hdr_bytes, position_colors, probe_camera

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def mv():
    import massivevoxelraytracing_amd as m
    m.lib()
    return m


def test_calls_out_of_order_report_and_recover(mv, O):
    pt = mv.PathTracer()
    cam = probe_camera(np.zeros(3, np.float32), np.float32(1.0 / 64), 64)
    with pytest.raises(mv.MvrtError, match="mvrt_pt_setup first"):
        pt.step(None, cam)
    pt.setup(None)
    with pytest.raises(mv.MvrtError, match="no scene"):
        pt.step(None, cam)
    svo = pt.m_intersectorOctreeGPU
    ro = np.zeros((4, 3), np.float32)
    rd = np.ones((4, 3), np.float32)
    with pytest.raises(mv.MvrtError, match="no octree"):
        svo.intersect(ro, rd)
    with pytest.raises(mv.MvrtError, match="power of two"):
        svo.build_synthetic(100, 10, seed=1)
    with pytest.raises(mv.MvrtError, match="bad resolution"):
        pt.resizeFrameBufferIfNeeded(None, 0, 16)
    with pytest.raises(mv.MvrtError, match="bad tile"):
        pt.set_tile(3, 3)
    with pytest.raises(mv.MvrtError, match="pipeline depth"):
        pt.set_pipeline_depth(9)
    # ... and the same handles work afterwards, bit for bit
    tris = bunny_tris()
    cols, emis = position_colors(tris)
    res, w, h = 64, 96, 54
    sc = O.build_scene_from_triangles(tris, res, cols, emis)
    pt.updateScene(tris.reshape(-1, 3), cols.reshape(-1, 3), emis.reshape(-1, 3), None, sc.origin, sc.dps, res)
    with pytest.raises(mv.MvrtError, match="no frame buffer"):
        pt.step(None, cam)
    pt.resizeFrameBufferIfNeeded(None, w, h)
    with pytest.raises(mv.MvrtError, match="HDRI enabled"):  # the reference's default scale is 1.75: lighting is on until a map is loaded or the scale is set to 0
        pt.step(None, cam)
    rgba, hw, hh = O.decode_rgbe(hdr_bytes())
    pt.loadHDRIPixels(None, rgba, hw, hh, rgba, hw, hh)
    cam = probe_camera(sc.origin, sc.dps, res, focus=9.0, lens_r=0.05)
    pt.step(None, cam)
    H = O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1)
    fb, _, cnt = sc.render_pt(H, cam, w, h, 0, math_mode=1, threads=8)
    assert np.array_equal(pt.read_framebuffer()[: w * h], fb) and pt.stats()["rays"] == cnt["rays"]


def test_lighting_switched_off_without_a_map_renders_black_sky(mv, O):
    """scale 0 and no HDRI loaded (the reference would read a null image for every primary miss, voxKernel.cu:682): equals a loaded map at scale 0"""
    tris = bunny_tris()
    cols, emis = position_colors(tris)
    res, w, h = 64, 96, 54
    sc = O.build_scene_from_triangles(tris, res, cols, emis)
    cam = probe_camera(sc.origin, sc.dps, res, focus=9.0, lens_r=0.05)
    pt = mv.PathTracer()
    pt.setup(None)
    pt.resizeFrameBufferIfNeeded(None, w, h)
    pt.updateScene(tris.reshape(-1, 3), cols.reshape(-1, 3), emis.reshape(-1, 3), None, sc.origin, sc.dps, res)
    pt.set_hdri_scale(0.0)
    pt.step(None, cam)
    rgba, hw, hh = O.decode_rgbe(hdr_bytes())
    H = O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1)
    H.set_scale(0.0)
    fb, _, cnt = sc.render_pt(H, cam, w, h, 0, math_mode=1, threads=8)
    got = pt.read_framebuffer()[: w * h]
    assert np.array_equal(got, fb) and pt.stats()["rays"] == cnt["rays"] and np.isfinite(got).all()


def test_a_frame_that_cannot_fit_names_the_way_out(mv):
    """a frame whose wavefront state exceeds the free HBM is refused before anything is allocated, and the message names the remedy"""
    pt = mv.PathTracer()
    pt.setup(None)
    pt.m_intersectorOctreeGPU.build_synthetic(64, 5000, seed=3)
    pt.set_hdri_scale(0.0)
    pt.set_batch_steps(1)
    cam = probe_camera(np.zeros(3, np.float32), np.float32(1.0 / 64), 64)
    with pytest.raises(mv.MvrtError, match="mvrt_pt_set_tile"):
        pt.resizeFrameBufferIfNeeded(None, 32768, 32768)  # 1.07 G pixels: 17 G samples per step, 3.3 TB of path state
        pt.step(None, cam)
    pt.set_tile(0, 4096)  # a 1/4096 share of it fits
    pt.resizeFrameBufferIfNeeded(None, 32768, 32768)
    pt.step(None, cam)
    assert pt.stats()["samples"] == pt.owned_pixels() * 16


def test_a_reallocation_that_fails_leaves_no_stale_buffers(mv):
    """a frame is allocated, then a RE-allocation at the same frame size fails (deeper pipeline / bigger batch against a pretended
    smaller free HBM).  The handle must not keep pointers into the freed path state: step() returns an error (no launch onto freed memory), a
    resize at the very same size allocates again, and the frame then renders exactly what an untouched handle renders."""
    def fresh():
        pt = mv.PathTracer()
        pt.setup(None)
        pt.m_intersectorOctreeGPU.build_synthetic(64, 5000, seed=3)
        pt.set_hdri_scale(0.0)
        return pt
    cam = probe_camera(np.zeros(3, np.float32), np.float32(1.0 / 64), 64)
    w, h = 256, 128
    ref = fresh()
    ref.resizeFrameBufferIfNeeded(None, w, h)
    ref.step(None, cam)
    want = ref.read_framebuffer()

    pt = fresh()
    pt.set_pipeline_depth(1)
    pt.set_batch_steps(1)
    pt.resizeFrameBufferIfNeeded(None, w, h)
    pt.step(None, cam)
    assert np.array_equal(pt.read_framebuffer(), want)
    per_pass = w * h * 16 * 190
    pt.set_test_free_bytes(int(per_pass * 0.5))  # pretend: not even one pass of one step fits the free HBM
    with pytest.raises(mv.MvrtError, match="mvrt_pt_set_tile"):
        pt.set_pipeline_depth(2)  # re-allocates the path state at the unchanged frame size: releases what is there, then fails its budget
    with pytest.raises(mv.MvrtError):
        pt.step(None, cam)  # no frame any more: an error, not a launch onto freed buffers
    with pytest.raises(mv.MvrtError):
        pt.resizeFrameBufferIfNeeded(None, w, h)  # same size, still no room: fails again (does NOT return early as "nothing to do")
    pt.set_test_free_bytes(0)
    pt.resizeFrameBufferIfNeeded(None, w, h)  # same size, room again: allocates
    pt.step(None, cam)
    assert np.array_equal(pt.read_framebuffer(), want)


def test_inputs_the_reference_never_sees_do_not_fault(mv):
    """the reference's applications derive origin / dps from the mesh's bounding box and never trace NaN rays; a library is handed anything: a grid
    that covers part of the mesh or none of it, NaN / inf / degenerate / huge triangles, zero rays, NaN / zero / inf rays, a NaN camera -- every
    call comes back with a result or an error (tools/robust_probe.py prints the same cases)"""
    tris = bunny_tris()
    v = tris.reshape(-1, 3).copy()
    white, black = np.ones_like(v), np.zeros_like(v)
    lo = v.min(0)
    ext = float((v.max(0) - lo).max())
    res = 64

    def build(vv, origin, dps):
        s = mv.IntersectorOctreeGPU()
        s.build(vv, white, black, None, origin, dps, res)
        return s

    full = build(v, lo, ext / res).info().numberOfVoxels
    part = build(v, lo + 0.4 * ext, 0.2 * ext / res).info().numberOfVoxels  # the grid covers a corner of the mesh only: clipped
    assert 0 < part < full
    with pytest.raises(mv.MvrtError, match="touch no voxel"):
        build(v, lo + 100 * ext, ext / res)
    for mutate in (lambda a: a.__setitem__(5, np.nan), lambda a: a.__setitem__((7, 1), np.inf), lambda a: a.__setitem__(slice(3, 6), a[3].copy()),
                   lambda a: a.__setitem__(slice(9, 12), a[9:12] * 1e30)):
        vv = v.copy()
        mutate(vv)
        n = build(vv, lo, ext / res).info().numberOfVoxels
        assert abs(int(n) - int(full)) < 64  # the one bad triangle is dropped or clipped, the rest of the mesh is there
    s = build(v, lo, ext / res)
    out = s.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert all(len(x) == 0 for x in out.values())
    ro = np.array([[np.nan, 0, 0], [0, 0, 0], [1e30, 1e30, 1e30], [0, 0, 0]], np.float32)
    rd = np.array([[1, 0, 0], [0, 0, 0], [1, 1, 1], [np.inf, 1, np.nan]], np.float32)
    assert (s.intersect(ro, rd)["t"] == np.float32(3.4028235e38)).all()  # all of them miss
    pt = mv.PathTracer()
    pt.setup(None)
    pt.resizeFrameBufferIfNeeded(None, 64, 36)
    pt.set_hdri_scale(0.0)
    pt.updateScene(v, white, black, None, lo, ext / res, res)
    cam = np.array(probe_camera(lo, np.float32(ext / res), res), np.float32, copy=True)
    cam[:3] = np.nan
    pt.step(None, cam)
    fb = pt.read_framebuffer()[: 64 * 36]
    assert np.isfinite(fb).all() and (fb[:, 3] == 16).all()


def test_tiny_frames_empty_tile_shares_and_handle_lifetimes(mv):
    """1x1 and 3x2 frames, tile shares that own no pixel of a small frame, two path tracers side by side, a handle destroyed with steps in flight, a
    rebuild between steps (tools/robust_probe2.py prints the same)"""
    import gc
    tris = bunny_tris()
    v = tris.reshape(-1, 3)
    white, black = np.ones_like(v), np.zeros_like(v)
    lo = v.min(0)
    ext = float((v.max(0) - lo).max())
    res = 64
    cam = probe_camera(lo, np.float32(ext / res), res)

    def mk(w, h, tile=(0, 1)):
        pt = mv.PathTracer()
        pt.setup(None)
        pt.set_tile(*tile)
        pt.resizeFrameBufferIfNeeded(None, w, h)
        pt.set_hdri_scale(0.0)
        pt.updateScene(v, white, black, None, lo, ext / res, res)
        return pt

    for w, h, tile, pixels in ((1, 1, (0, 1), 1), (3, 2, (0, 1), 6), (100, 37, (31, 32), 0), (100, 37, (14, 32), 116), (17, 1, (1, 2), 0)):
        pt = mk(w, h, tile)
        for _ in range(2):
            pt.step(None, cam)
        fb = pt.read_framebuffer()
        assert pt.owned_pixels() == 256 and int((fb[:, 3] == 32).sum()) == pixels and int((fb[:, 3] == 0).sum()) == 256 - pixels
        assert pt.stats()["samples"] == pixels * 32 and (pt.stats()["rays"] > 0) == (pixels > 0)
    a, b = mk(64, 36), mk(48, 27)
    for _ in range(3):
        a.step(None, cam)
        b.step(None, cam)
    assert a.read_framebuffer()[: 64 * 36, 3].min() == 48 and b.read_framebuffer()[: 48 * 27, 3].min() == 48
    c = mk(640, 360)
    for _ in range(5):
        c.step(None, cam)
    del c
    gc.collect()  # destroyed with deferred / in-flight steps
    d = mk(64, 36)
    d.step(None, cam)
    d.updateScene(v, white, black, None, lo, ext / res, 128)
    d.step(None, cam)
    assert d.read_framebuffer()[: 64 * 36, 3].min() == 32


def test_a_hint_that_names_no_voxel_does_not_fault(mv):
    """mvrt_trace_batch_hinted documents a hint without a voxel as an error the library does not detect: the results of such rays are unspecified,
    but nothing may be read outside the prefix tables (codes beyond the grid, cells that are empty) and the call comes back"""
    tris = bunny_tris()
    v = tris.reshape(-1, 3)
    lo = v.min(0)
    dps = np.float32((v.max(0) - lo).max() / 256)
    svo = mv.IntersectorOctreeGPU()
    svo.build(v, None, None, None, lo, dps, 256)
    rng = np.random.default_rng(5)
    n = 50_000
    ro = (lo + rng.random((n, 3)) * dps * 256).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    junk = rng.integers(0, 2**63, n, dtype=np.uint64)          # codes far beyond a 256^3 grid
    empty = rng.integers(0, 2**24, n, dtype=np.uint64)         # cells of the grid, almost all of them empty
    for hints in (junk, empty):
        out = svo.intersect_hinted(ro, rd, hints)
        assert np.isfinite(out["t"]).all() and len(out["t"]) == n
    plain = svo.intersect(ro, rd, want_descents=True)
    assert np.array_equal(svo.intersect_hinted(ro, rd, np.full(n, 2**64 - 1, np.uint64))["t"], plain["t"])  # "no hint" for every ray
