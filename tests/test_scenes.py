"""The procedural stand-in scenes (massivevoxelraytracing_amd/scenes.py) are the benchmark's workload: pin them.  The reference assets they stand in
for (xyzrgb_dragon, rtcamp9) are not in the reference tree.  sin/cos are float64 polynomials of IEEE +,-,* (no libm, no SIMD
kernels), so the triangle soups -- vertices, colours, emissions -- hash the same on every host."""
import hashlib

import numpy as np
import pytest

from massivevoxelraytracing_amd import scenes

PINS = {
    ("dragon", 1.0): (1102122, "4d14c6e417bac4cba77e2790ce4723305895135eb88732b6d9dc5a2509d31ffe"),
    ("dragon", 0.25): (67602, "5f278dec4bd4682562e1d252570c3a87048bdf1fa33181942ef814e9db35bb3f"),
    ("rtcamp", 1.0): (5943488, "9364ad2b5e50f98e3bed8c09052b2aa267804d8c76e952bc65a8d8f0af0d0273"),
    ("rtcamp", 0.25): (366056, "e7abec97fff2c6d34f79d60bbe4abcd6d186a6c51ed67bca5ace56e49df5c6f6"),
    ("cave", 1.0): (3174934, "9b86c0b1f4792a8f8853aeb5342b3dce44e78cb81c8c07686d215b86a786b2b1"),
    ("cave", 0.25): (193116, "818054457b6c0a8a4e103ec0ae680df3cbe5687a2577b96c79e42380be4c48a8"),
    ("tunnel", 1.0): (6691890, "237193df5dc1a3027e74531700268d06f07fe1b1bf4977fd7f1fa9fe87d4e29d"),
    ("tunnel", 0.25): (413604, "33d5774b825d0f6c676c1468a94d71b3495fd50388827f4a932f59d81e8a357a"),
}


def scene_digest(v, c, e):
    return hashlib.sha256(v.tobytes() + c.tobytes() + e.tobytes()).hexdigest()


@pytest.mark.parametrize("name,detail", sorted(PINS))
def test_standin_scene_is_pinned(name, detail):
    v, c, e = scenes.SCENES[name](detail)
    tris, digest = PINS[(name, detail)]
    assert v.dtype == c.dtype == e.dtype == np.float32 and len(v) == len(c) == len(e) == 3 * tris
    assert scene_digest(v, c, e) == digest


def test_deterministic_trig_matches_libm_closely():
    x = np.linspace(-60.0, 60.0, 200001)
    assert np.abs(scenes.dsin(x) - np.sin(x)).max() < 4e-16 and np.abs(scenes.dcos(x) - np.cos(x)).max() < 4e-16


def test_look_at_camera_is_pinned():
    cam = scenes.look_at_camera((1, 2, 3), (0, 0, 0), 40.0, 3.0, 0.02)
    assert hashlib.sha256(cam.tobytes()).hexdigest() == "9f75ed2d913fadbf4c86b6a60b243e1f7cd2a3d2c59e2a53caf980c04b49bdca"
    # orthonormal frame, front towards the target
    f, u, r = cam[3:6], cam[6:9], cam[9:12]
    assert abs(np.dot(f, u)) < 1e-6 and abs(np.dot(f, r)) < 1e-6 and abs(np.dot(u, r)) < 1e-6
    assert np.allclose(f, -np.array([1, 2, 3]) / np.sqrt(14.0), atol=1e-6)
