"""The C oracle against a second restatement written independently from the C# text
(tests/reference_restatement.py: object-oriented, recursive per light, scalar numpy arithmetic),
and known answers evaluated in double precision.  The reference ships no vectors, so this is the
strongest pin available: two restatements of different shape agreeing bit for bit."""
import numpy as np
import pytest

from raytracer_amd import Camera, Light, Material, Plane, Scene, Sphere, Triangle, default_scene, synthetic_scene
from tests.reference_restatement import Restatement

f32 = np.float32


def rgb(px):
    return (int(px) >> 16) & 255, (int(px) >> 8) & 255, int(px) & 255


def test_default_scene_frame_bit_for_bit(oracle):
    """64 x 64, the reference's depth bound, both wall variants (the mirror sphere, the reflecting
    ceiling, the checkerboard, the triangle and every wall are in view)."""
    s, cam = default_scene(), Camera()
    ref = Restatement(s, cam).render(64, 64)
    for mode in (oracle.MODE_MEMOISED, oracle.MODE_LITERAL):
        got, _ = oracle.render(s.flatten(), cam, 64, 64, max_depth=5, mode=mode)
        assert np.array_equal(got, ref), np.argwhere(got != ref)[:5]
    assert len(np.unique(ref)) > 500


def test_textured_wall_and_moved_camera_bit_for_bit(oracle, wall_texture):
    s = default_scene(textured_wall=True)
    cam = Camera(position=(0.4, -0.3, 0.2), direction=(-0.6, 0.0, -0.8), fov=75.0)   # looks at the z = -2 wall
    ref = Restatement(s, cam, textures=[wall_texture]).render(40, 30)
    got, _ = oracle.render(s.flatten(), cam, 40, 30, max_depth=5, textures=[wall_texture])
    assert np.array_equal(got, ref)
    assert len(np.unique(ref)) > 100


def test_synthetic_scene_three_lights_bit_for_bit(oracle):
    s = synthetic_scene(40, radius=(0.1, 0.4), seed=12)
    s.Add(Triangle((-1.0, 0.5, 2.0), (1.0, 0.5, 2.5), (0.0, -1.0, 3.0), Material(1.0, specular=0.3)))
    s.Add(Light(0.0, -3.0, 2.0))
    cam = Camera(position=(0.2, -0.5, -0.5), direction=(0.0, 0.1, 1.0), fov=80.0)
    ref = Restatement(s, cam, max_depth=3).render(32, 24)
    got, _ = oracle.render(s.flatten(), cam, 32, 24, max_depth=3)
    assert np.array_equal(got, ref)


def test_mirror_chain_through_every_level_in_double(oracle):
    """Two mirrors facing each other (objects.cs:80-94): the centre ray bounces until the depth
    bound cuts it.  The double-precision evaluation and the float oracle agree to a byte."""
    mirror = Material(0.2, specular=0.9, diffuseColor=(0.9, 0.8, 0.7, 1.0), ambientColor=(0.01, 0.02, 0.03, 1.0))
    s = Scene()
    s.Add(Plane(0.0, 0.0, -1.0, 3.0, mirror))      # z = 3, in front
    s.Add(Plane(0.0, 0.0, 1.0, 2.0, mirror))       # z = -2, behind
    s.Add(Light(0.7, -0.4, 0.5))
    s.Add(Light(-0.3, 0.6, 1.5))
    cam = Camera()
    w = h = 33
    ref64 = Restatement(s, cam, scalar=np.float64)
    got, cnt = oracle.render(s.flatten(), cam, w, h, max_depth=5)
    assert cnt["rays_secondary"] == 5 * w * h      # every level but the last reflects
    for (x, y) in [(16, 16), (3, 29), (30, 2), (10, 20)]:
        colour, _ = ref64.pixel(x, y, w, h)
        want = [int(c * 255.0) for c in colour[:3]]
        assert all(abs(a - b) <= 1 for a, b in zip(rgb(got[y, x]), want)), ((x, y), rgb(got[y, x]), want)
    assert len(np.unique(got)) > 50


def test_reflecting_ceiling_and_checkerboard_in_double(oracle):
    """Default scene at the template's 1080 x 1080: pixels on the specular-0.5 ceiling (top rows,
    MyApplication.cs:90-95,135) and a run of floor pixels across checkerboard edges
    (objects.cs:176-188) against the double-precision evaluation."""
    s, cam = default_scene(), Camera()
    ref64 = Restatement(s, cam, scalar=np.float64)
    w = h = 1080
    rows = {40: [100, 540, 900], 1000: list(range(300, 780, 8))}
    flips = 0
    for y, xs in rows.items():
        got, _ = oracle.render(s.flatten(), cam, w, h, max_depth=5, rows=(y, y + 1))
        prev = None
        for x in xs:
            colour, _ = ref64.pixel(x, y, w, h)
            want = [int(c * 255.0) for c in colour[:3]]
            assert all(abs(a - b) <= 1 for a, b in zip(rgb(got[y, x]), want)), ((x, y), rgb(got[y, x]), want)
            if prev is not None and abs(sum(want) - sum(prev)) > 150:
                flips += 1   # black <-> white square
            prev = want
    assert flips >= 2


def test_texels_of_the_reference_wall_texture(oracle):
    """Three texels of the reference's wall_texture.jpg as Pillow (libjpeg) decodes them — (0, 0) =
    (42, 62, 87), (959, 539) = (81, 127, 127), (1919, 1079) = (35, 59, 97) — placed in an otherwise
    black 1920 x 1080 image: hit points that Plane.GetTextureCoordinates (objects.cs:190-204) maps
    to them must be shaded with those colours (Texture.GetColor, objects.cs:348-353).  (The last
    column and row are addressed only when a coordinate rounds to 1.0f: (int)(u * (W - 1)) with
    u < 1 stops at W - 2, so the corner texel is looked up one short of it.)"""
    tex = np.zeros((1080, 1920, 4), dtype=np.uint8)
    tex[..., 3] = 255
    texels = {(0, 0): (42, 62, 87), (959, 539): (81, 127, 127), (1919, 1079): (35, 59, 97)}
    for (tx, ty), c in texels.items():
        tex[ty, tx, :3] = c
    s = default_scene(textured_wall=True)
    fs = s.flatten()
    wall = next(i for i, p in enumerate(s.primitives) if isinstance(p, Plane) and p.texture == 0)
    ref64 = Restatement(s, Camera(), scalar=np.float64, textures=[tex])
    tex[1078, 1918, :3] = texels[(1919, 1079)]
    # (hit points on the z = -2 wall inside the box, so that both lights reach them)
    hits = {(0, 0): (2.4999, -3.9999), (959, 539): (0.0, -1.5), (1918, 1078): (-2.49995, 0.99995)}
    for (tx, ty), (x, y) in hits.items():
        hit = (x, y, -2.0)
        assert oracle.texel(hit, 1920, 1080) == (tx, ty)
        view = (0.0, 0.0, -1.0)
        got = oracle.shade(fs, wall, hit, view, textures=[tex])
        want = ref64.shade(s.primitives[wall], ref64.A.vec(hit), ref64.A.vec(view), 0)
        assert np.allclose(got[:3], [float(c) for c in want[:3]], atol=2e-6), ((tx, ty), got, want)
    # and the colour really is the texel's: a lit point shows its ratios
    hit = (0.0, -1.5, -2.0)
    got = oracle.shade(fs, wall, hit, (0.0, 0.0, -1.0), textures=[tex])
    base = oracle.shade(fs, wall, (0.0, -1.4, -2.0), (0.0, 0.0, -1.0), textures=[tex])   # a black texel next to it
    d = got[:3] - base[:3]
    assert d[1] > 1e-3   # the point is lit
    assert abs(d[0] / d[1] - 81 / 127) < 1e-3 and abs(d[2] / d[1] - 1.0) < 1e-3
