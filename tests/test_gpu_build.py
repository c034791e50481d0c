"""GPU SVO construction (mvrt_svo_build = IntersectorOctreeGPU::build) against the oracle's CPU builder
(voxelize -> mergeVoxels -> buildOctreeDAGReference -> embedMasks).  Bar: the voxel list, the
attributes AND the node array are bit-identical -- node numbering included (the reference's own GPU
builder numbers DAG nodes racily; ours reproduces the CPU reference's creation order)."""
import numpy as np
import pytest

from common import bunny_tris, golden, hdr_bytes, position_colors, probe_camera

pytestmark = pytest.mark.gpu
G = golden()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def mv():
    import massivevoxelraytracing_amd as m
    m.lib()
    return m


def gpu_build(mv, tris, res, cols=None, emis=None):
    from massivevoxelraytracing_amd import scenes
    v = tris.reshape(-1, 3)
    origin, dps = scenes.bounding_grid(v, res)
    svo = mv.IntersectorOctreeGPU()
    svo.build(v, None if cols is None else cols.reshape(-1, 3), None if emis is None else emis.reshape(-1, 3), None, origin, dps, res)
    return svo, origin, dps


def assert_same_svo(O, mv, svo, sc):
    info = svo.info()
    assert info.totalDumpedVoxels == sc.dumped
    assert info.numberOfVoxels == len(sc.morton)
    assert info.numberOfNodes == len(sc.nodes)
    assert info.hasEmission == sc.has_emission
    assert np.array_equal(np.array(info.lower[:], np.float32), sc.bounds()[0])
    assert np.array_equal(np.array(info.upper[:], np.float32), sc.bounds()[1])
    nodes, attrs, morton = svo.download(want_morton=True)
    assert np.array_equal(morton, sc.morton)
    assert np.array_equal(attrs, sc.attrs)
    want = sc.nodes.copy()
    want["_pad"] = 0
    got = nodes.view(O.NODE_DTYPE).copy()
    got["_pad"] = 0
    assert np.array_equal(got["mask"], want["mask"])
    assert np.array_equal(got["children"], want["children"])
    assert np.array_equal(got["psum"], want["psum"])


def test_bunny256_build_matches_golden_and_oracle(mv, O):
    tris = bunny_tris()
    svo, origin, dps = gpu_build(mv, tris, 256)
    g = G["bunny"]["256"]
    info = svo.info()
    assert (info.totalDumpedVoxels, info.numberOfVoxels, info.numberOfNodes) == (g["dumped"], g["voxels"], g["dag_nodes"])
    assert np.float32(dps) == np.float32(G["bunny"]["dps_256"])
    assert_same_svo(O, mv, svo, O.build_scene_from_triangles(tris, 256))
    # and the built octree traces like the golden
    cam = probe_camera(origin, dps, 256)
    r = svo.render(cam, 1920, 1080)
    assert int((r["t"] != O.MAX_FLOAT).sum()) == g["primary_1080p"]["hits"]


def test_bunny_with_attributes_and_emission(mv, O):
    tris = bunny_tris()
    cols, emis = position_colors(tris)
    svo, _, _ = gpu_build(mv, tris, 128, cols, emis)
    sc = O.build_scene_from_triangles(tris, 128, cols, emis)
    assert sc.has_emission == 1
    assert_same_svo(O, mv, svo, sc)


def test_bunny1024_build(mv, O):
    tris = bunny_tris()
    svo, _, _ = gpu_build(mv, tris, 1024)
    g = G["bunny"]["1024"]
    info = svo.info()
    assert (info.totalDumpedVoxels, info.numberOfVoxels, info.numberOfNodes) == (g["dumped"], g["voxels"], g["dag_nodes"])
    assert_same_svo(O, mv, svo, O.build_scene_from_triangles(tris, 1024))


def test_procedural_scene_build(mv, O):
    from massivevoxelraytracing_amd import scenes
    v, c, e = scenes.dragon_standin(0.25)
    svo, _, _ = gpu_build(mv, v, 512, c, e)
    sc = O.build_scene_from_triangles(v.reshape(-1, 9), 512, c.reshape(-1, 9), e.reshape(-1, 9))
    assert_same_svo(O, mv, svo, sc)


@pytest.mark.parametrize("res", [256, 2048])
def test_huge_triangles_are_voxelized_by_the_whole_wave(mv, O, res):
    """the reference voxelizes every triangle on one thread whatever its size (voxKernel.cu:58-166); here triangles with a large footprint are done by the whole wave,
    columns dealt to the lanes, which changes who emits which voxel but not the voxel list: walls of two triangles across the grid in all three major axes (incl. a
    slanted one with colours and emission), mixed with the bunny's small triangles in the same waves -- voxels, attributes and nodes equal the oracle's, also in
    conservative mode.  (At 2048^3 a wall is 4.2 M cells: seconds on one lane, milliseconds here.)"""
    import time
    small = bunny_tris()[:3000].reshape(-1, 3)
    lo, hi = small.min(0), small.max(0)
    a, b = lo - 0.3, hi + 0.3
    walls = np.array([
        [[a[0], a[1], a[2]], [b[0], a[1], a[2]], [b[0], a[1], b[2]]], [[a[0], a[1], a[2]], [b[0], a[1], b[2]], [a[0], a[1], b[2]]],   # floor (major y)
        [[a[0], a[1], a[2]], [a[0], b[1], a[2]], [a[0], b[1], b[2]]], [[a[0], a[1], a[2]], [a[0], b[1], b[2]], [a[0], a[1], b[2]]],   # wall (major x)
        [[a[0], a[1], b[2]], [b[0], a[1], b[2]], [b[0], b[1], b[2]]],                                                                   # half a wall (major z)
        [[a[0], a[1], a[2]], [b[0], b[1], a[2] + 0.2], [a[0] + 0.1, b[1], b[2]]],                                                       # slanted
    ], np.float32).reshape(-1, 3)
    v = np.concatenate([small[:1500], walls[:9], small[1500:], walls[9:]], axis=0).astype(np.float32)
    rng = np.random.default_rng(3)
    cols = rng.random(v.shape).astype(np.float32)
    emis = np.where(rng.random((len(v), 1)) < 0.1, rng.random(v.shape), 0.0).astype(np.float32)
    from massivevoxelraytracing_amd import scenes
    origin, dps = scenes.bounding_grid(v, res)
    for flags in (0, 4):
        svo = mv.IntersectorOctreeGPU()
        t0 = time.perf_counter()
        svo.build(v, cols, emis, None, origin, dps, res, flags=flags)
        mv.synchronize()
        dt = time.perf_counter() - t0
        assert dt < 5.0, dt
        if flags == 0 or res == 256:
            m, at = O.voxelize(v.reshape(-1, 9), origin, dps, res, cols.reshape(-1, 9), emis.reshape(-1, 9), six_separating=(flags == 0))
            dumped = len(m)
            m, at, he = O.merge_voxels(m, at)
            sc = O.Scene(O.build_octree(m, res), at, origin, dps, res, he)
            sc.morton, sc.dumped = m, dumped
            assert_same_svo(O, mv, svo, sc)


def test_tiny_and_degenerate_inputs(mv, O):
    one = np.array([[0.1, 0.2, 0.3, 0.9, 0.25, 0.35, 0.4, 0.8, 0.7]], np.float32)
    for res in (2, 4, 32):
        svo = mv.IntersectorOctreeGPU()
        svo.build(one.reshape(-1, 3), None, None, None, np.zeros(3, np.float32), np.float32(1.0 / res), res)
        sc = O.build_scene_from_triangles(one, res, origin=np.zeros(3, np.float32), dps=np.float32(1.0 / res))
        assert_same_svo(O, mv, svo, sc)
    svo = mv.IntersectorOctreeGPU()
    with pytest.raises(mv.MvrtError, match="power of two"):  # IntersectorOctreeGPU.hpp:48-51 aborts
        svo.build(one.reshape(-1, 3), None, None, None, np.zeros(3, np.float32), 0.01, 100)
    with pytest.raises(mv.MvrtError, match="touch no voxel"):
        svo.build(one.reshape(-1, 3) + 50.0, None, None, None, np.zeros(3, np.float32), np.float32(1 / 32), 32)


def test_update_scene_then_step_matches_oracle(mv, O):
    """PathTracer::updateScene -> step, the call order of voxPTGPU.cpp:169-189"""
    from massivevoxelraytracing_amd import scenes
    tris = bunny_tris()
    cols, emis = position_colors(tris)
    origin, dps = scenes.bounding_grid(tris.reshape(-1, 3), 256)
    rgba, hw, hh = O.decode_rgbe(hdr_bytes())
    w, h = 160, 90
    pt = mv.PathTracer()
    pt.setup(None)
    pt.resizeFrameBufferIfNeeded(None, w, h)
    pt.loadHDRIPixels(None, rgba, hw, hh, rgba, hw, hh)
    pt.updateScene(tris.reshape(-1, 3), cols.reshape(-1, 3), emis.reshape(-1, 3), None, origin, dps, 256)
    cam = probe_camera(origin, dps, 256, focus=9.0, lens_r=0.05)
    pt.step(None, cam)
    sc = O.build_scene_from_triangles(tris, 256, cols, emis)
    want, _, _ = sc.render_pt(O.HDRI(rgba, hw, hh, rgba, hw, hh, 1), cam, w, h, 0, math_mode=1, threads=8)
    assert np.array_equal(pt.read_framebuffer()[: w * h], want)
    assert pt.getNumberOfVoxels() == len(sc.morton) and pt.getOctreeBytes() == len(sc.nodes) * 68


@pytest.mark.parametrize("res", [64, 256])
def test_conservative_voxelization_mode(mv, O, res):
    """MVRT_BUILD_CONSERVATIVE = VTContext's sixSeparating == false (voxelization.hpp:186-189,296-301): every voxel a triangle
    touches.  Voxel list, attributes and node array equal the oracle's conservative build; it is a superset of the six-separating one."""
    from massivevoxelraytracing_amd import scenes
    tris = bunny_tris()
    cols, emis = position_colors(tris)
    v = tris.reshape(-1, 3)
    origin, dps = scenes.bounding_grid(v, res)
    svo = mv.IntersectorOctreeGPU()
    svo.build(v, cols.reshape(-1, 3), emis.reshape(-1, 3), None, origin, dps, res, flags=svo.BUILD_CONSERVATIVE)
    morton_d, attrs_d = O.voxelize(tris, origin, dps, res, cols, emis, six_separating=False)
    morton_w, attrs_w, he = O.merge_voxels(morton_d, attrs_d)
    nodes_w = O.build_octree(morton_w, res, dag=True, embed=True)
    info = svo.info()
    assert info.totalDumpedVoxels == len(morton_d) and info.numberOfVoxels == len(morton_w) and info.numberOfNodes == len(nodes_w)
    nodes, attrs, morton = svo.download(want_morton=True)
    assert np.array_equal(morton, morton_w) and np.array_equal(attrs, attrs_w)
    got = nodes.view(O.NODE_DTYPE)
    for f in ("mask", "children", "psum"):
        assert np.array_equal(got[f], nodes_w[f]), f
    six = O.build_scene_from_triangles(tris, res, cols, emis)
    assert len(morton_w) > len(six.morton) and np.isin(six.morton, morton_w).all()


@pytest.mark.parametrize("res,flags", [(2, 0), (4, 0), (8, 0), (16, 0), (64, 0), (512, 0), (1024, 0), (4096, 0), (256, 2)])
def test_cell_index_gives_the_rank_the_psum_walk_gives(mv, O, res, flags):
    """An octree BUILT here resolves a hit voxel's index through the cell index (Morton rank of the cell's first voxel + popcount: two gathers), an UPLOADED one by
    walking nVoxelsPSum along the path as the reference does (voxCommon.hpp:388-391).  Same node array, same rays: the two must agree voxel for voxel -- and with the
    oracle -- at every block shape (octrees of fewer than 4 levels have blocks of fewer than 8 x 8 x 8 cells) and for the plain-index flavour (flags = 2: masks not embedded)."""
    tris = bunny_tris()
    v = tris.reshape(-1, 3)
    from massivevoxelraytracing_amd import scenes
    origin, dps = scenes.bounding_grid(v, res)
    built = mv.IntersectorOctreeGPU()
    built.build(v, None, None, None, origin, dps, res, flags=flags)
    nodes, attrs, morton = built.download(want_morton=True)
    walked = mv.IntersectorOctreeGPU()
    walked.upload(nodes, attrs, origin, dps, res, 0, embeddedMask=(flags & 2) == 0)
    sc = O.build_scene_from_triangles(tris, res, embed=(flags & 2) == 0)
    rng = np.random.default_rng(res)
    lo, hi = sc.bounds()
    n = 200_000
    ro = ((lo + hi) / 2 + (rng.random((n, 3)) - 0.5) * (hi - lo).max() * 2.5).astype(np.float32)
    ro[: n // 4] = (lo + rng.random((n // 4, 3)) * (hi - lo)).astype(np.float32)  # a quarter starts inside the grid
    rd = ((lo + rng.random((n, 3)) * (hi - lo)) - ro).astype(np.float32)
    a = built.intersect(ro, rd)
    b = walked.intersect(ro, rd)
    ref = sc.trace(ro, rd, threads=8)
    hit = ref["t"] != O.MAX_FLOAT
    assert hit.sum() > n // 20
    for r in (a, b):
        assert np.array_equal(r["t"], ref["t"])
        assert np.array_equal(r["vIndex"][hit], ref["vIndex"][hit])
    # every voxel, not only the ones random rays reach: a ray straight down onto each voxel centre from just above it hits THAT voxel or one above it in the column;
    # where it hits the voxel itself the index must be its Morton rank
    xyz = np.array([O.morton_decode(int(m)) for m in morton[:: max(1, len(morton) // 50_000)]], np.float64)
    centre = (np.asarray(origin, np.float64) + (xyz + 0.5) * float(dps))
    ro2 = (centre + np.array([0.0, 0.75 * float(dps), 0.0])).astype(np.float32)
    rd2 = np.tile(np.array([0.0, -1.0, 0.0], np.float32), (len(ro2), 1))
    a2, b2, ref2 = built.intersect(ro2, rd2), walked.intersect(ro2, rd2), sc.trace(ro2, rd2, threads=8)
    h2 = ref2["t"] != O.MAX_FLOAT
    assert np.array_equal(a2["t"], ref2["t"]) and np.array_equal(a2["vIndex"][h2], ref2["vIndex"][h2]) and np.array_equal(b2["vIndex"][h2], ref2["vIndex"][h2])
