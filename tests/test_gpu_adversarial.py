"""Adversarial parity: inputs chosen to sit on the conservative filters' margins (the sphere
filter, the bounding balls, the per-light direction tables) and on the reference's own edge
behaviour (NaN directions, zero distances, huge coordinates).  Every frame must equal the CPU
restatement exactly; a mismatch is a finding against a proof in rt_kernels.hip / rt_api.cpp, never
a tolerance to loosen."""
import numpy as np
import pytest

from raytracer_amd import (Camera, Light, Material, Plane, RenderContext, Scene, Sphere, Triangle,
                           default_scene, scene_1k, scene_64k, synthetic_scene)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = RenderContext()
    c.set_profiling(True)   # the Intersect-call counters are compared too
    yield c
    c.close()


def exact(ctx, oracle, scene, cam, w, h, depth, what, counters=True):
    fs = scene.flatten() if isinstance(scene, Scene) else scene
    ctx.set_scene(fs)
    gpu = ctx.render(cam, w, h, depth)
    ref, cnt = oracle.render(fs, cam, w, h, max_depth=depth)
    bad = np.argwhere(gpu != ref)
    assert bad.size == 0, f"{what}: {len(bad)} of {w * h} pixels differ, first at (y, x) = {tuple(bad[0])}: " \
                          f"{gpu[tuple(bad[0])]:#x} vs {ref[tuple(bad[0])]:#x}"
    if counters:
        st = ctx.stats()
        for k in ("rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests", "plane_tests", "triangle_tests"):
            assert st[k] == cnt[k], (what, k, st[k], cnt[k])
    return gpu


def moved(scene: Scene, scale: float, offset) -> Scene:
    """The scene under p -> scale * p + offset (planes: n.p + d = 0 keeps its normal)."""
    ox, oy, oz = offset
    out = Scene()
    for p in scene.primitives:
        if isinstance(p, Sphere):
            out.Add(Sphere(p.position[0] * scale + ox, p.position[1] * scale + oy, p.position[2] * scale + oz,
                           p.radius * scale, p.material))
        elif isinstance(p, Plane):
            n = p.normal
            out.Add(Plane(n[0], n[1], n[2], float(p.d) * scale - (n[0] * ox + n[1] * oy + n[2] * oz), p.material))
        else:
            tr = lambda v: (v[0] * scale + ox, v[1] * scale + oy, v[2] * scale + oz)
            out.Add(Triangle(tr(p.v0), tr(p.v1), tr(p.v2), p.material))
    for l in scene.lights:
        out.Add(Light(l.position[0] * scale + ox, l.position[1] * scale + oy, l.position[2] * scale + oz))
    return out


@pytest.mark.parametrize("scale,offset", [(1.0, (1e3, -2e3, 5e2)), (1.0, (1e4, 1e4, -1e4)), (1.0, (-1e5, 3e4, 1e5)),
                                          (100.0, (0.0, 0.0, 0.0)), (0.01, (3.0, 1.0, -2.0)), (0.02, (1e3, 1e3, 1e3))])
def test_scaled_and_translated_scenes(ctx, oracle, scale, offset):
    """Coordinates of 1e3 .. 1e5 with radii of 1e-3 .. 10: the margins of the filters scale with the
    coordinates, the reference's own rounding with them."""
    base = synthetic_scene(300, radius=(0.02, 0.15), seed=77)
    s = moved(base, scale, offset)
    cam = Camera(position=tuple(scale * c + o for c, o in zip((0.2, -0.3, 0.1), offset)), direction=(0.0, 0.0, 1.0))
    gpu = exact(ctx, oracle, s, cam, 120, 80, 3, f"scale {scale} offset {offset}")
    assert len(np.unique(gpu)) > 20   # a real picture, not a constant


def test_rays_tangent_to_spheres(ctx, oracle):
    """A pencil of almost parallel rays grazing spheres: discriminants within a few ulp of zero,
    both signs (the filter must keep every pair the exact evaluation could accept)."""
    s = Scene()
    white = Material(1.0, ambientColor=(0.3, 0.3, 0.3, 1.0))
    mirror = Material(0.0, specular=1.0, ambientColor=(0, 0, 0, 1), diffuseColor=(0, 0, 0, 1))
    s.Add(Sphere(0.0, 0.0, 5.0, 1.0, white))
    s.Add(Sphere(2.0, 0.0, 9.0, 1.0, mirror))
    s.Add(Sphere(1.0, 2.0, 14.0, 1.0, white))
    s.Add(Plane(0.0, 0.0, -1.0, 30.0, white))
    s.Add(Light(1.0, -5.0, 0.0))
    s.Add(Light(1.0, 0.0, -3.0))   # behind the camera, on the tangent line
    for fov in (1e-4, 1e-3, 0.05):
        exact(ctx, oracle, s, Camera(position=(1.0, 0.0, 0.0), direction=(0.0, 0.0, 1.0), fov=fov), 96, 96, 4,
              f"tangent pencil fov {fov}")
    # grazing from far away with a large offset between origin and sphere
    exact(ctx, oracle, s, Camera(position=(1.0, 0.0, -1e4), direction=(0.0, 0.0, 1.0), fov=0.01), 96, 64, 3,
          "tangent pencil from 1e4 away")


def test_camera_far_from_the_cluster(ctx, oracle):
    s = synthetic_scene(500, radius=(0.001, 0.2), seed=5)
    exact(ctx, oracle, s, Camera(position=(0.0, -1.0, -1e4), direction=(0.0, 0.0, 1.0), fov=0.05), 128, 72, 3,
          "camera 1e4 away")
    exact(ctx, oracle, s, Camera(position=(3e3, -4e3, 2e3), direction=(-0.557086, 0.742781, -0.371391), fov=0.1), 96, 64, 2,
          "camera 5e3 away, oblique")


def test_degenerate_spheres_in_a_valid_scene(ctx, oracle):
    """Zero radius, NaN and infinite data: such spheres can never be hit; the others must be
    rendered as if they were not there (never-pass filter entries, always-pass balls)."""
    s = synthetic_scene(120, radius=(0.05, 0.2), seed=9)
    m = Material(1.0)
    bad = [(0.0, 0.0, 2.0, 0.0), (float("nan"), 0.0, 2.0, 0.3), (0.0, float("inf"), 2.0, 0.3),
           (0.5, 0.5, 1.5, float("nan")), (0.5, 0.5, 1.5, float("inf")), (1e30, 1e30, 1e30, 1.0),
           (0.1, -0.2, 1.0, -0.5), (3e38, 0.0, 0.0, 3e38)]
    for i, (x, y, z, r) in enumerate(bad):
        s.primitives.insert(7 * i + 3, Sphere(x, y, z, r, m))
    with np.errstate(all="ignore"):
        exact(ctx, oracle, s, Camera(), 120, 90, 3, "degenerate spheres")


def test_lights_in_awkward_places(ctx, oracle):
    """A light on a plane, on a sphere's surface, inside a sphere, at a sphere's centre, far away,
    and at the camera: distances of zero give NaN directions, which must come out as the
    reference's (int)NaN packing and not as a crash."""
    s = synthetic_scene(150, radius=(0.05, 0.25), seed=21)
    s.lights.clear()
    first = next(p for p in s.primitives if isinstance(p, Sphere))
    c, r = first.position, float(first.radius)
    for pos in [(0.0, 1.0, 2.0),                      # on the floor plane y = 1
                (float(c[0]) + r, float(c[1]), float(c[2])),   # on a sphere
                (float(c[0]), float(c[1]), float(c[2])),       # at its centre
                (float(c[0]) + 0.5 * r, float(c[1]), float(c[2])),   # inside it
                (1e4, -1e4, 1e4),                     # far outside everything
                (0.0, 0.0, 0.0)]:                     # at the camera
        s.Add(Light(*pos))
    with np.errstate(all="ignore"):
        exact(ctx, oracle, s, Camera(), 128, 96, 3, "awkward lights")
        # a hit point exactly at a light: plane z = 2 seen head-on, light at the pixel's hit point
        t = Scene()
        t.Add(Plane(0.0, 0.0, -1.0, 2.0, Material(1.0)))
        t.Add(Light(0.0, 0.0, 2.0))
        exact(ctx, oracle, t, Camera(), 65, 65, 2, "light at a hit point")


def test_degenerate_cameras(ctx, oracle):
    s = default_scene()
    with np.errstate(all="ignore"):
        gpu = exact(ctx, oracle, s, Camera(direction=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0)), 64, 48, 5, "up parallel to direction")
        assert not gpu.any()   # Cross(up, direction) = 0 -> NaN rays -> nothing is hit
        exact(ctx, oracle, s, Camera(fov=0.0), 64, 48, 5, "fov 0")
        exact(ctx, oracle, s, Camera(fov=179.9), 64, 48, 5, "fov 179.9")
        exact(ctx, oracle, s, Camera(fov=360.0), 48, 48, 5, "fov 360")
        exact(ctx, oracle, s, Camera(direction=(0.0, 0.0, 0.0)), 32, 32, 5, "zero direction")
        exact(ctx, oracle, s, Camera(position=(float("nan"), 0.0, 0.0)), 32, 32, 5, "NaN position")


def test_open_scene_with_far_hits(ctx, oracle):
    """No box: a ground plane to the horizon, so hit points lie up to 1e6 away from the spheres —
    far outside the region the lights' direction tables serve (those origins fall back on every
    sphere) — with several lights, reflections off the ground and many levels."""
    s = Scene()
    ground = Material(1.0, diffuseColor=(0.8, 0.8, 0.7, 1.0), ambientColor=(0.05, 0.05, 0.05, 1.0), specular=0.4)
    s.Add(Plane(0.0, -1.0, 0.0, 1.0, ground))                     # y = 1, the "floor" (world up is -y on screen)
    rng = np.random.default_rng(3)
    mirror = Material(0.0, specular=1.0, ambientColor=(0, 0, 0, 1), diffuseColor=(0, 0, 0, 1))
    for i in range(260):
        x, z = rng.uniform(-4, 4), rng.uniform(1, 9)
        r = rng.uniform(0.05, 0.4)
        s.Add(Sphere(x, 1.0 - r - rng.uniform(0, 1.5), z, r, mirror if i % 3 == 0 else Material(1.0)))
    for pos in [(2.0, -3.0, 1.0), (-3.0, -2.0, 6.0), (0.0, -8.0, 4.0), (50.0, -20.0, 30.0), (0.5, 0.2, 0.5)]:
        s.Add(Light(*pos))
    gpu = exact(ctx, oracle, s, Camera(position=(0.0, 0.3, 0.0), direction=(0.0, 0.02, 1.0), fov=60.0), 160, 100, 5, "horizon")
    assert len(np.unique(gpu)) > 50 and (gpu == 0).any()   # sky (misses) and lit ground and spheres
    exact(ctx, oracle, s, Camera(position=(0.0, 0.999, -5.0), direction=(0.0, 0.0, 1.0), fov=40.0), 128, 64, 4,
          "grazing the ground")


def test_enclosing_sphere_and_many_lights(ctx, oracle):
    """A sphere around camera, lights and everything else (rays start inside it and never hit it;
    every light lies inside it), 70 lights."""
    s = synthetic_scene(90, radius=(0.05, 0.3), seed=31)
    s.primitives.insert(2, Sphere(0.0, -1.0, 1.5, 60.0, Material(1.0)))
    s.lights.clear()
    for i in range(70):
        s.Add(Light(-3.5 + 0.1 * i, -3.0 + 0.05 * i, 0.3 + 0.04 * i))
    exact(ctx, oracle, s, Camera(), 96, 64, 2, "enclosing sphere, 70 lights")


def test_thin_frames_over_64k_spheres(ctx, oracle):
    """1 x N and N x 1 frames on the 64-chunk scene: ragged chunks and ragged tiles together."""
    fs = scene_64k().flatten()
    for (w, h) in [(1, 96), (96, 1), (3, 67)]:
        exact(ctx, oracle, fs, Camera(), w, h, 8, f"64k spheres {w}x{h}")


@pytest.mark.parametrize("pipeline", ["levels", "fused"])
def test_both_forms_of_the_levels_stage(oracle, monkeypatch, pipeline):
    """Small frames normally take the single-launch form of the levels' work; large ones one launch
    per level and queue, where many-chunk scenes trace small levels as (tile, chunk) jobs,
    many-light scenes shade short queues with the lights spread over waves, and both reorder the
    records of deeper levels by position.  Both forms forced on
    the scenes that reach those paths, with the work counters on and off: same pixels, same counts."""
    monkeypatch.setenv("RT_HIP_PIPELINE", pipeline)
    monkeypatch.setenv("RT_HIP_BIN_MIN", "64")   # (levels: even these small levels are put into bin order)
    with RenderContext() as c:
        for profiling in (1, 0):
            c.set_profiling(profiling)
            fs = scene_64k().flatten()                      # 64 chunks, 16 lights, 8 levels
            exact(c, oracle, fs, Camera(), 80, 45, 8, f"64k spheres, {pipeline}", counters=profiling == 1)
            # two lights: the reflecting records' light terms are computed ahead of their children
            exact(c, oracle, scene_1k().flatten(), Camera(), 128, 72, 4, f"1k spheres, {pipeline}", counters=profiling == 1)
            s = synthetic_scene(300, radius=(0.05, 0.3), seed=5)   # one chunk, 5 and then 33 lights
            s.lights.clear()
            for i in range(5):
                s.Add(Light(-2.0 + i, -3.0 + 0.3 * i, 0.5 + 0.4 * i))
            exact(c, oracle, s, Camera(), 96, 64, 4, f"5 lights, {pipeline}", counters=profiling == 1)
            for i in range(28):
                s.Add(Light(-3.0 + 0.2 * i, -2.5, 0.4 + 0.1 * i))
            exact(c, oracle, s, Camera(position=(0.2, -0.1, 0.0)), 64, 48, 3, f"33 lights, {pipeline}",
                  counters=profiling == 1)


def test_chunk_jobs_and_tile_walks_agree_on_a_large_frame(oracle, monkeypatch):
    """The 64-chunk scene on a frame large enough for one launch per level by itself: level 0 has
    enough tiles for the tile-serial walk, deeper levels become (tile, chunk) jobs; with the switch
    forced either way every level takes one form.  All three frames are identical, and every 40th
    row equals the CPU restatement."""
    fs = scene_64k().flatten()
    w, h, depth = 640, 360, 6
    frames = []
    monkeypatch.setenv("RT_HIP_PIPELINE", "levels")
    with RenderContext() as c:
        c.set_scene(fs)
        for below in ("", "1", "1000000"):
            if below:
                monkeypatch.setenv("RT_HIP_SPLIT_BELOW", below)
                c.reload_tuning()
            frames.append(c.render(Camera(), w, h, depth))
            c.frame_status()
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    ref, _ = oracle.render(fs, Camera(), w, h, max_depth=depth, row_step=40)
    rows = list(range(0, h, 40))
    assert np.array_equal(frames[0][rows], ref[rows])
