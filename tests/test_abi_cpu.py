"""CPU-only checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/mvrt.h declares; the Python mirror binds exactly that set; no product module
touches the oracle."""
import ctypes
import os
import re

import numpy as np

import massivevoxelraytracing_amd as mv
from massivevoxelraytracing_amd import tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mvrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvrt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(mv.LIB_PATH)
    syms = header_symbols()
    assert len(syms) > 40
    for s in syms:
        assert hasattr(lib, s), "libmvrt_hip.so does not export " + s


def test_python_mirror_binds_exactly_the_header():
    assert sorted(mv.SIGNATURES) == header_symbols()
    mv.lib()  # binding every symbol must succeed without touching the GPU


def test_error_reporting_without_gpu_call():
    lib = mv.lib()
    # argument validation happens before any HIP call
    assert lib.mvrt_svo_upload(None, None, 0, None, 0, None, 0.0, 256, 0, 1, None) != 0
    assert b"empty octree" in lib.mvrt_last_error()


def test_camera_from_matrices_matches_reference_formula():
    # a rigid view matrix (column-major) and a GL perspective; renderCommon.hpp:21-35
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    eye = np.array([1.0, 2.0, 3.0], np.float32)
    view = np.eye(4, dtype=np.float32)
    view[:3, :3] = R
    view[:3, 3] = -R @ eye
    proj = np.zeros((4, 4), np.float32)
    f = 1.0 / np.tan(np.radians(45.0) / 2)
    proj[0, 0], proj[1, 1], proj[2, 2], proj[3, 2], proj[2, 3] = f / (16 / 9), f, -1.0, -1.0, -0.2
    cam = mv.camera_from_matrices(view.T.reshape(-1), proj.T.reshape(-1), focus=2.5, lens_r=0.1)  # .T -> column-major
    assert np.allclose(cam[0:3], eye, atol=1e-6)          # m_o
    assert np.allclose(cam[3:6], -R[2], atol=1e-7)        # m_front = -row 2
    assert np.allclose(cam[6:9], R[1], atol=1e-7)         # m_up
    assert np.allclose(cam[9:12], R[0], atol=1e-7)        # m_right
    assert np.isclose(cam[12], 1.0 / f) and cam[13] == np.float32(0.1) and cam[14] == np.float32(2.5)
    from oracle import oracle as O
    assert np.array_equal(cam, O.camera_from_matrices(view.T.reshape(-1), proj.T.reshape(-1), 2.5, 0.1))


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "massivevoxelraytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert not re.search(r'#include\s+"[^"]*oracle', text), fn
                assert "import oracle" not in text and "from oracle" not in text, fn
                assert "libmvrt_oracle" not in text, fn
    for fn in os.listdir(os.path.join(ROOT, "include")):
        p = os.path.join(ROOT, "include", fn)
        if os.path.isfile(p):
            assert "libmvrt_oracle" not in open(p).read()


def test_tile_partition_is_a_bijection():
    for (w, h, n) in [(1920, 1080, 8), (1440, 900, 3), (100, 37, 2), (64, 4, 5), (17, 3, 1)]:
        seen = np.zeros(w * h, np.int32)
        for r in range(n):
            g = tiles.global_pixel_index(w, h, r, n)
            assert len(g) == tiles.owned_pixels(w, h, n)
            ok = g[g >= 0]
            seen[ok] += 1
            # padding only at the tail of a rank's list
            assert (np.diff((g >= 0).astype(np.int8)) <= 0).all()
        assert (seen == 1).all()
