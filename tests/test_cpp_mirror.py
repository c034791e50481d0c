"""The header-only C++ mirrors of the reference host structs compile and link against libmvrt_hip.so."""
import os
import shutil
import subprocess

import pytest

import massivevoxelraytracing_amd as mv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_compiles_and_links(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "mirror_usage"
    libdir = os.path.dirname(mv.LIB_PATH)
    cmd = [gxx, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_usage.cpp"), "-o", str(exe),
           "-L", libdir, "-l:libmvrt_hip.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    out = subprocess.check_output([str(exe)], env=dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert b"usage" in out
