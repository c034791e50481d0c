"""`-m gpu` coverage of every BASELINE.json configuration at its stated size (config 1 = bunny 256^3 lives in test_gpu_parity.py):

  config 2  dragon stand-in 1024^3, primary cast 1920x1080      -- whole frame bit-exact against the oracle
  config 3  dragon stand-in 2048^3, path trace 1920x1080        -- builder == oracle builder (depth-11 embedded DAG), 200 k rays and a
                                                                   131 072-pixel band bit-exact, frame properties, compaction indices
  config 4  rtcamp stand-in 4096^3, path trace 1920x1080        -- oracle band, properties, 8 tile shares assembled == 1-tile frame
  config 5  synthetic 8192^3 non-DAG octree at the stress size  -- properties that need no oracle (too big for it)
  closed    cave stand-in 1024^3 (camera inside)                -- the deep-bounce regime (>= 10 rays per sample), oracle band

The stand-in scenes are procedural (the reference's assets are not in its tree) and pinned by tests/test_scenes.py.  The oracle builds its own
octree from the same triangles, so the GPU voxelizer + DAG builder are checked at these sizes too (nodes, attributes, voxel codes)."""
import os

import numpy as np
import pytest

from common import GOLDEN
from test_gpu_parity import assert_hits_equal, random_rays

pytestmark = pytest.mark.gpu
W, H = 1920, 1080
THREADS = min(16, len(os.sched_getaffinity(0)))


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def mv():
    import massivevoxelraytracing_amd as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def hdr(O):
    return O.decode_rgbe(open(os.path.join(GOLDEN, "monks_forest_s.hdr"), "rb").read())


def scene_and_oracle(O, name, res):
    from massivevoxelraytracing_amd import scenes
    v, c, e = scenes.SCENES[name](1.0)
    origin, dps = scenes.bounding_grid(v, res)
    sc = O.build_scene_from_triangles(v.reshape(-1, 9), res, c.reshape(-1, 9), e.reshape(-1, 9), origin=origin, dps=dps)
    return (v, c, e, origin, dps), sc


def assert_same_octree(O, svo, sc):
    info = svo.info()
    assert (info.numberOfNodes, info.numberOfVoxels, info.hasEmission) == (len(sc.nodes), len(sc.morton), sc.has_emission)
    nodes, attrs, morton = svo.download(want_morton=True)
    got = nodes.view(O.NODE_DTYPE)
    assert np.array_equal(morton, sc.morton) and np.array_equal(attrs, sc.attrs)
    for f in ("mask", "children", "psum"):
        assert np.array_equal(got[f], sc.nodes[f]), f


def bench_camera(scenes, info, name):
    lo, hi = np.array(info.lower[:]), np.array(info.upper[:])
    centre = (lo + hi) / 2
    if name == "cave":
        return scenes.cave_camera(lo, hi)
    if name == "tunnel":
        return scenes.tunnel_camera(lo, hi)
    eye = centre + (np.array([2.6, 1.5, 3.1]) if name == "dragon" else np.array([4.2, 2.2, 5.0]))
    return scenes.look_at_camera(eye, centre, 40.0, float(np.linalg.norm(eye - centre)), 0.02)


def make_pt(mv, hdr, tile=(0, 1), w=W, h=H):
    rgba, hw, hh = hdr
    pt = mv.PathTracer()
    pt.setup(None)
    pt.set_tile(*tile)
    pt.resizeFrameBufferIfNeeded(None, w, h)
    pt.loadHDRIPixels(None, rgba, hw, hh, rgba, hw, hh)
    return pt


def frame_properties(fb, st, n_pixels, steps=1):
    assert np.isfinite(fb).all() and (fb[:, :3] >= 0).all()
    assert (fb[:n_pixels, 3] == 16.0 * steps).all()
    assert st["samples"] == n_pixels * 16 * steps
    assert st["samples"] <= st["rays"] <= 18 * st["samples"]  # 1 + 8 x (shadow + bounce) + 1 extra, voxKernel.cu:675,691-760
    assert st["shadowRays"] <= 8 * st["samples"] and st["hits"] <= st["rays"] - st["shadowRays"]


def densest_band(fb, w, n):
    """start pixel of the n-pixel run of the frame with the most accumulated radiance variation (i.e. through geometry, not sky)"""
    lum = fb[:, :3].sum(1)
    rows = np.abs(np.diff(lum.reshape(-1, w), axis=1)).sum(1)
    k = -(-n // w)  # whole rows that cover n pixels: the band must end inside the frame (the oracle writes every pixel of it)
    best = int(np.argmax(np.convolve(rows, np.ones(k), "valid")))
    assert best * w + n <= len(fb)
    return best * w


# ---------------------------------------------------------------------------------------------------------------------------
def test_config2_dragon_1024_primary_cast(mv, O):
    """voxRTGPU path (`render`, voxKernel.cu:437-483): 2 073 600 pixel-centre rays, hits / colours / descents bit-exact"""
    from massivevoxelraytracing_amd import scenes
    (v, c, e, origin, dps), sc = scene_and_oracle(O, "dragon", 1024)
    svo = mv.IntersectorOctreeGPU()
    svo.build(v, c, e, None, origin, dps, 1024)
    assert_same_octree(O, svo, sc)
    assert svo.info().embeddedMask == 1 and svo.info().levels == 10
    cam = bench_camera(scenes, svo.info(), "dragon")
    for vertex_colour in (True, False):
        want = sc.render_primary(cam, W, H, show_vertex_color=vertex_colour, threads=THREADS)
        got = svo.render(cam, W, H, showVertexColor=vertex_colour)
        assert np.array_equal(want["rgba"], got["rgba"])
        assert_hits_equal(want, got)
    hit = got["t"] != O.MAX_FLOAT
    assert 200_000 < hit.sum() < W * H and got["descents"][hit].mean() > 25


@pytest.fixture(scope="module")
def dragon2048(O):
    return scene_and_oracle(O, "dragon", 2048)


def test_config3_dragon_2048_octree_and_rays(mv, O, dragon2048):
    """the headline octree: depth-11 embedded DAG (18.9 M voxels).  GPU builder == oracle builder node for node; 200 k mixed rays
    (shadow flags, axis-parallel directions, interior origins) bit-exact incl. descents"""
    (v, c, e, origin, dps), sc = dragon2048
    svo = mv.IntersectorOctreeGPU()
    svo.build(v, c, e, None, origin, dps, 2048)
    assert_same_octree(O, svo, sc)
    info = svo.info()
    assert info.embeddedMask == 1 and info.levels == 11 and info.numberOfVoxels > 15_000_000
    ro, rd = random_rays(sc, 200_000, 2048)
    sh = (np.arange(len(ro)) % 3 == 0).astype(np.uint8)
    want = sc.trace(ro, rd, sh, threads=THREADS, want_descents=True)
    assert_hits_equal(want, svo.intersect(ro, rd, sh, want_descents=True))
    assert want["descents"].max() >= 60  # deep walks through shared (DAG) subtrees


def test_config3_dragon_2048_path_trace_full_hd(mv, O, dragon2048, hdr):
    """one 1920x1080 step: frame properties, a 131 072-pixel band (2.1 M samples) bit-exact, stable compaction of the live paths checked
    index by index against the oracle's per-sample path lengths, determinism of a second run"""
    from massivevoxelraytracing_amd import scenes
    (v, c, e, origin, dps), sc = dragon2048
    rgba, hw, hh = hdr
    pt = make_pt(mv, hdr)
    pt.updateScene(v, c, e, None, origin, dps, 2048)
    cam = bench_camera(scenes, pt.m_intersectorOctreeGPU.info(), "dragon")
    pt.set_debug_capture(True)
    pt.set_batch_steps(1)
    pt.step(None, cam)
    fb = pt.read_framebuffer()
    st = pt.stats()
    frame_properties(fb, st, W * H)
    survivors = [pt.debug_stage_survivors(s, W * H * 16) for s in range(8)]
    n = 131072
    p0 = densest_band(fb, W, n)
    Hd = O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1)
    ref = np.zeros((W * H, 4), np.float32)
    hits = np.zeros(n * 16, np.uint8)
    _, _, cnt = sc.render_pt(Hd, cam, W, H, 0, math_mode=1, fb=ref, pixel_begin=p0, pixel_end=p0 + n, threads=THREADS, path_hits=hits)
    assert np.array_equal(fb[p0:p0 + n], ref[p0:p0 + n])
    assert cnt["rays"] > 3 * cnt["samples"]  # the band runs through geometry
    # compaction: after stage s the survivor list must be strictly increasing (stable) over the WHOLE frame, and inside the band hold exactly the
    # samples whose path made more than s hits
    for s, tasks in enumerate(survivors):
        assert (np.diff(tasks.astype(np.int64)) > 0).all(), s
        band = tasks[(tasks >= p0 * 16) & (tasks < (p0 + n) * 16)] - p0 * 16
        assert np.array_equal(band, np.nonzero(hits > s)[0]), s
    pt.set_debug_capture(False)
    pt.clearFrameBuffer(None)
    pt.step(None, cam)
    assert np.array_equal(pt.read_framebuffer(), fb)


def test_config4_rtcamp_4096_tiles(mv, O, hdr):
    """rtcamp stand-in at 4096^3 (32 M voxels, depth 12): oracle band, properties, and the 8-way tile split of configs[3] --
    each rank's share rendered on its own PathTracer, gathered and assembled on the device == the 1-tile frame, bit for bit"""
    from massivevoxelraytracing_amd import scenes, tiles
    (v, c, e, origin, dps), sc = scene_and_oracle(O, "rtcamp", 4096)
    rgba, hw, hh = hdr
    full = make_pt(mv, hdr)
    full.updateScene(v, c, e, None, origin, dps, 4096)
    info = full.m_intersectorOctreeGPU.info()
    assert_same_octree(O, full.m_intersectorOctreeGPU, sc)
    assert info.levels == 12 and info.numberOfVoxels > 25_000_000 and info.embeddedMask == 1
    cam = bench_camera(scenes, info, "rtcamp")
    full.step(None, cam)
    fb = full.read_framebuffer()
    frame_properties(fb, full.stats(), W * H)
    n = 65536
    p0 = densest_band(fb, W, n)
    ref = np.zeros((W * H, 4), np.float32)
    sc.render_pt(O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1), cam, W, H, 0, math_mode=1, fb=ref, pixel_begin=p0, pixel_end=p0 + n, threads=THREADS)
    assert np.array_equal(fb[p0:p0 + n], ref[p0:p0 + n])
    nodes, attrs, _ = full.m_intersectorOctreeGPU.download()
    del full
    ranks = 8
    owned = tiles.owned_pixels(W, H, ranks)
    gathered = mv.DeviceArray((ranks * owned, 4), np.float32)
    rays = 0
    for r in range(ranks):
        pt = make_pt(mv, hdr, tile=(r, ranks))
        assert pt.owned_pixels() == owned
        pt.m_intersectorOctreeGPU.upload(nodes, attrs, origin, dps, 4096, info.hasEmission)  # the replicated SVO
        pt.step(None, cam)
        pt.join(None)
        mv.memcpy_d2d(gathered.ptr + r * owned * 16, pt.framebuffer_dev(), owned * 16)
        mv.synchronize()
        rays += pt.stats()["rays"]
        del pt
    frame = mv.DeviceArray((W * H, 4), np.float32)
    mv.assemble_tiles(gathered, ranks, owned, W, H, frame)
    mv.synchronize()
    assert np.array_equal(frame.to_host(), fb[: W * H])


def test_config5_synthetic_8192_stress_size(mv, O):
    """BASELINE.json configs[4] at the size `bench.py --mode stress` runs: 8192^3 non-DAG octree far beyond the Infinity Cache and beyond
    the embedded-mask limit (plain 32-bit child indices, 64-bit addressing).  No oracle at this size: run-to-run determinism, geometry of
    the hit point, full-depth walks, shadow/normal agreement, vIndex range and monotonicity along a ray bundle."""
    res, n_vox, n_rays = 8192, int(float(os.environ.get("MVRT_TEST_STRESS_VOXELS", "1.05e9"))), 2_000_000
    svo = mv.IntersectorOctreeGPU()
    svo.build_synthetic(res, n_vox, seed=2024, flags=svo.BUILD_NO_DAG | svo.BUILD_NO_EMBEDDED_MASK)
    info = svo.info()
    assert info.levels == 13 and info.embeddedMask == 0
    assert info.numberOfNodes * 68 > 190e9 and svo.traversal_bytes() < 40e9  # ~200 GB in the reference's layout; 16-byte two-level bricks + 5 B per node here
    assert 0.97 * n_vox < info.numberOfVoxels <= n_vox and info.numberOfNodes > 2 * info.numberOfVoxels
    rng = np.random.default_rng(5)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ro = (0.5 + 1.2 * d).astype(np.float32)
    rd = (rng.random((n_rays, 3), dtype=np.float32) - ro).astype(np.float32)
    a = svo.intersect(ro, rd, want_descents=True)
    b = svo.intersect(ro, rd, want_descents=True)
    for k in ("t", "nMajor", "vIndex", "descents"):
        assert np.array_equal(a[k], b[k]), k
    hit = a["t"] != O.MAX_FLOAT
    assert hit.sum() > 0.9 * n_rays
    assert (a["t"][hit] > 0).all() and (a["vIndex"][hit] < info.numberOfVoxels).all()
    assert (a["descents"][hit] >= info.levels).all()
    p = ro[hit].astype(np.float64) + rd[hit].astype(np.float64) * a["t"][hit][:, None].astype(np.float64)
    axis = np.array([2, 0, 1])[a["nMajor"][hit]]
    cc = p[np.arange(len(p)), axis] * res
    assert np.abs(cc - np.round(cc)).max() < 1e-2
    assert ((p > -1e-4) & (p < 1 + 1e-4)).all()
    # voxels are numbered in morton order, so the vIndex ranges of the 8^3 coarse cells are disjoint and ordered by the cells' morton codes.
    # Voxel of a hit: behind the hit face along nMajor, floor() on the other two axes (hits within 0.01 voxel of a coarse cell boundary
    # in those two axes are left out: their cell is decided by the last bits of p, which is recomputed here in float64)
    vox = p * res
    k = np.round(cc).astype(np.int64)
    rda = rd[hit][np.arange(len(p)), axis]
    coord = np.floor(vox).astype(np.int64)
    coord[np.arange(len(p)), axis] = np.where(rda > 0, k, k - 1)
    near = np.abs(vox - np.round(vox / 1024.0) * 1024.0) < 0.01
    near[np.arange(len(p)), axis] = False
    ok = ~near.any(1) & (coord >= 0).all(1) & (coord < res).all(1)
    cell = coord[ok] >> 10
    code = np.zeros(len(cell), np.int64)
    for bit in range(3):
        code |= ((cell[:, 0] >> bit) & 1) << (3 * bit) | ((cell[:, 1] >> bit) & 1) << (3 * bit + 1) | ((cell[:, 2] >> bit) & 1) << (3 * bit + 2)
    vi = a["vIndex"][hit][ok].astype(np.int64)
    lo = np.full(512, np.iinfo(np.int64).max)
    hi = np.full(512, -1)
    np.minimum.at(lo, code, vi)
    np.maximum.at(hi, code, vi)
    seen = hi >= 0
    assert seen.sum() > 400
    assert (lo[seen][1:] > hi[seen][:-1]).all()
    s = svo.intersect(ro[:500_000], rd[:500_000], np.ones(500_000, np.uint8))
    assert np.array_equal(s["t"] != O.MAX_FLOAT, hit[:500_000]) and (s["vIndex"] == 0).all()
    # download of a slice is not offered at this size; the reference-layout round trip is covered at 512^3 / 2048^3 in test_gpu_large_octree.py


def test_closed_scene_cave_1024(mv, O, hdr):
    """the rtcamp9 regime the open stand-ins miss: camera INSIDE a closed room, nearly every bounce hits (>= 10 rays per sample, close to
    the 18-ray bound), all nine wavefront stages stay populated.  Full-HD step: properties, oracle band, stable compaction in every stage."""
    from massivevoxelraytracing_amd import scenes
    (v, c, e, origin, dps), sc = scene_and_oracle(O, "cave", 1024)
    rgba, hw, hh = hdr
    pt = make_pt(mv, hdr)
    pt.updateScene(v, c, e, None, origin, dps, 1024)
    assert_same_octree(O, pt.m_intersectorOctreeGPU, sc)
    cam = bench_camera(scenes, pt.m_intersectorOctreeGPU.info(), "cave")
    pt.set_debug_capture(True)
    pt.set_batch_steps(1)
    pt.step(None, cam)
    fb = pt.read_framebuffer()
    st = pt.stats()
    frame_properties(fb, st, W * H)
    assert st["rays"] >= 10 * st["samples"]
    n = 32768
    p0 = (H // 2) * W
    hits = np.zeros(n * 16, np.uint8)
    ref = np.zeros((W * H, 4), np.float32)
    _, _, cnt = sc.render_pt(O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1), cam, W, H, 0, math_mode=1, fb=ref, pixel_begin=p0, pixel_end=p0 + n, threads=THREADS, path_hits=hits)
    assert np.array_equal(fb[p0:p0 + n], ref[p0:p0 + n])
    assert cnt["rays"] >= 10 * cnt["samples"]
    live = []
    for s in range(8):
        tasks = pt.debug_stage_survivors(s, W * H * 16)
        live.append(len(tasks))
        assert (np.diff(tasks.astype(np.int64)) > 0).all(), s
        band = tasks[(tasks >= p0 * 16) & (tasks < (p0 + n) * 16)] - p0 * 16
        assert np.array_equal(band, np.nonzero(hits > s)[0]), s
    assert live[7] > 0.5 * W * H * 16  # most paths are still alive going into the last bounce: buffers sized for the bound are really used


def test_closed_scene_tunnel_4096_the_size_of_the_reference_figure(mv, O, hdr):
    """the closed scene that is sized like the reference's published path-tracing figure (seminar slide 67: RT Camp scene, 4096^3, 41 M voxels,
    12.5 ms per sample on an RX 7900 XTX): ~38 M voxels at 4096^3, camera inside, >= 15 rays per sample.  GPU builder == oracle builder at this
    size, full-HD step: properties + a 16 384-pixel oracle band bit for bit + stable compaction in every stage."""
    from massivevoxelraytracing_amd import scenes
    (v, c, e, origin, dps), sc = scene_and_oracle(O, "tunnel", 4096)
    rgba, hw, hh = hdr
    pt = make_pt(mv, hdr)
    pt.updateScene(v, c, e, None, origin, dps, 4096)
    info = pt.m_intersectorOctreeGPU.info()
    assert 36_000_000 < info.numberOfVoxels < 42_000_000 and info.embeddedMask == 1
    assert_same_octree(O, pt.m_intersectorOctreeGPU, sc)
    cam = bench_camera(scenes, info, "tunnel")
    pt.set_debug_capture(True)
    pt.set_batch_steps(1)
    pt.step(None, cam)
    fb = pt.read_framebuffer()
    st = pt.stats()
    frame_properties(fb, st, W * H)
    assert st["rays"] >= 15 * st["samples"]
    n = 16384
    p0 = (H // 2) * W
    hits = np.zeros(n * 16, np.uint8)
    ref = np.zeros((W * H, 4), np.float32)
    _, _, cnt = sc.render_pt(O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1), cam, W, H, 0, math_mode=1, fb=ref, pixel_begin=p0, pixel_end=p0 + n, threads=THREADS, path_hits=hits)
    assert np.array_equal(fb[p0:p0 + n], ref[p0:p0 + n])
    for s in range(8):
        tasks = pt.debug_stage_survivors(s, W * H * 16)
        assert (np.diff(tasks.astype(np.int64)) > 0).all(), s
        band = tasks[(tasks >= p0 * 16) & (tasks < (p0 + n) * 16)] - p0 * 16
        assert np.array_equal(band, np.nonzero(hits > s)[0]), s
