#!/usr/bin/env python3
"""Regenerate the DATA fixtures under tests/golden/ from the reference's data files.

Run in the build container only (reads /root/reference; the GPU box has no reference tree).
Fixtures are data, never source: the triangle soup of scenes/bunny.obj as raw float32, and a
byte copy of the 8 KB environment map bin/monks_forest_s.hdr.
"""
import hashlib, os, shutil, sys
import numpy as np

REF = os.environ.get("MVRT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_obj_triangles(path):
    verts, tris = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                for k in range(1, len(idx) - 1):  # fan
                    tris.append([idx[0] - 1, idx[k] - 1, idx[k + 1] - 1])
    v = np.asarray(verts, dtype=np.float32)
    t = np.asarray(tris, dtype=np.int64)
    return v[t].reshape(-1, 9)  # nTri x (v0 v1 v2)


def main():
    os.makedirs(OUT, exist_ok=True)
    tri = load_obj_triangles(os.path.join(REF, "scenes", "bunny.obj"))
    tri.astype("<f4").tofile(os.path.join(OUT, "bunny_tris.f32"))
    shutil.copyfile(os.path.join(REF, "bin", "monks_forest_s.hdr"), os.path.join(OUT, "monks_forest_s.hdr"))
    for name in ("bunny_tris.f32", "monks_forest_s.hdr"):
        p = os.path.join(OUT, name)
        print(name, os.path.getsize(p), hashlib.sha256(open(p, "rb").read()).hexdigest())
    print("triangles", tri.shape[0], "bbox", tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0))


if __name__ == "__main__":
    sys.exit(main())
