"""Build libmvrt_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree.

    python -m massivevoxelraytracing_amd.build [--force]

-ffp-contract=off is part of the contract, not an optimisation choice: every fp32 traversal decision
must equal the CPU oracle's (DESIGN.md "FP rules").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("MVRT_LIB_OUT", os.path.join(HERE, "libmvrt_hip.so"))
SOURCES = ["api.hip", "kernels_rt.hip", "kernels_setup.hip", "svo_build.hip"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
    "-fgpu-rdc" if False else "-fno-gpu-rdc",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
] + os.environ.get("MVRT_EXTRA_FLAGS", "").split()


STAMP = OUT + ".stamp"  # travels with the .so (git-ignored, not gpurun-ignored)


def source_digest():
    """SHA-256 over the flags and every source/header the library is compiled from: the binary that travels to the GPU box
    is rebuilt whenever its sources differ from what it was built from -- file times say nothing after a checkout or a copy."""
    import hashlib
    h = hashlib.sha256()
    h.update("\0".join(FLAGS).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".hpp")))
    deps += [os.path.join(HERE, "..", "include", f) for f in ("mvrt.h", "mvrt_detmath.h")]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_digest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src))
        objs.append(obj)
    for p, src in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_digest())
    return OUT


APPS = {
    # name: (source, extra compile/link flags)
    "rtcamp_batch": ("rtcamp_batch.cpp", []),
    # single-process N-GPU tile driver: HIP runtime types for RCCL's stream argument + librccl
    "tile_render": ("tile_render.cpp", ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]),
}


def build_apps(verbose=True):
    """g++ the C++ host applications that sit on the header-only mirrors (apps/) against libmvrt_hip.so.  Returns the path of rtcamp_batch."""
    root = os.path.dirname(HERE)
    deps = [os.path.join(root, "apps", "scene_io.hpp")] + [os.path.join(root, "include", "mvrt", f) for f in ("PathTracer.hpp", "IntersectorOctreeGPU.hpp")] + [os.path.join(root, "include", "mvrt.h")]
    for name, (src_name, extra) in APPS.items():
        out = os.path.join(root, "apps", name)
        src = os.path.join(root, "apps", src_name)
        if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(f) for f in [src] + deps):
            continue
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unused-result", "-pthread", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "apps"), src, "-o", out,
               "-L", HERE, "-l:libmvrt_hip.so", "-Wl,-rpath," + HERE, "-Wl,--allow-shlib-undefined"] + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return os.path.join(root, "apps", "rtcamp_batch")


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_apps()
