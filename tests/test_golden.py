"""Golden fixtures (tests/golden, written by make_golden.py from the oracle).

CPU: the oracle still reproduces them (regression pin; the reference itself has no
vectors).  GPU: the library reproduces them through the C ABI without the oracle.
"""
import os

import numpy as np
import pytest

from raytracer_amd import Camera, default_scene, scene_1k
from tests.conftest import synthetic_texture

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "default_cyan_96": (lambda: default_scene(), Camera(), 96, 96, 5, False),
    "default_textured_96": (lambda: default_scene(textured_wall=True), Camera(), 96, 96, 5, True),
    "default_cyan_160x90_depth0": (lambda: default_scene(), Camera(), 160, 90, 0, False),
    "spheres1k_64x36_depth4": (lambda: scene_1k(), Camera(), 64, 36, 4, False),
    "default_moved_camera_80x60": (lambda: default_scene(),
                                   Camera(position=(-1.0, 0.2, 1.0), direction=(0.6, 0.0, 0.8), fov=70.0),
                                   80, 60, 5, False),
}


@pytest.fixture(scope="module")
def frames():
    return np.load(os.path.join(GOLD, "frames.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden_frames(oracle, frames, name):
    make, cam, w, h, depth, tex = CASES[name]
    img, _ = oracle.render(make().flatten(), cam, w, h, max_depth=depth,
                           textures=[synthetic_texture()] if tex else None)
    assert np.array_equal(img, frames[name])


def test_oracle_reproduces_golden_ray_vectors(oracle):
    rows = np.load(os.path.join(GOLD, "rays_default_scene.npy"))
    fs = default_scene().flatten()
    for r in rows:
        w, h, x, y = (int(v) for v in r[:4])
        d = oracle.ray_direction(Camera(), x, y, w, h)
        assert d.view(np.uint32).tolist() == [int(v) for v in r[4:7]]
        prim, t, hit = oracle.intersect(fs, (0, 0, 0), d)
        assert prim == int(r[7])
        if prim >= 0:
            assert int(np.float32(t).view(np.uint32)) == int(r[8])
            assert hit.view(np.uint32).tolist() == [int(v) for v in r[9:12]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_reproduces_golden_frames(frames, name):
    from raytracer_amd import RenderContext
    make, cam, w, h, depth, tex = CASES[name]
    with RenderContext() as ctx:
        if tex:
            ctx.set_texture(0, synthetic_texture())
        ctx.set_scene(make().flatten())
        img = ctx.render(cam, w, h, depth)
    gold = frames[name]
    assert float((img == gold).mean()) >= 0.9999, name
    d = np.abs(((img[..., None] >> np.array([16, 8, 0])) & 255) - ((gold[..., None] >> np.array([16, 8, 0])) & 255))
    assert d.max() <= 1
