"""The reflection rays' walk through the spheres' uniform grid (k_reflect_grid) against the CPU
restatement and against the kernels that walk the LDS-staged scene: the grid only selects which
spheres get the literal Sphere.Intersect, so every frame must be identical with it, without it, at
any resolution of the grid, and whatever part of the scene the grid's lists were built to serve."""
import numpy as np
import pytest

from raytracer_amd import Camera, Light, Material, Plane, RenderContext, Scene, Sphere, scene_1k, scene_64k, synthetic_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def grid_for_every_scene(monkeypatch):
    """By itself the library builds the grid only for scenes of more than 1024 spheres."""
    monkeypatch.setenv("RT_HIP_GRID", "2")


_oracle_frames = {}   # the CPU restatement's frames, rendered once per (scene name, frame)


def frames_equal_oracle(oracle, fs, cam, w, h, depth, what, row_step=1, key=None):
    with RenderContext() as c:
        c.set_profiling(True)   # the Intersect-call counters are compared too
        c.set_scene(fs)
        gpu = c.render(cam, w, h, depth)
        st = c.stats()
    if key is None or key not in _oracle_frames:
        _oracle_frames[key] = oracle.render(fs, cam, w, h, max_depth=depth, row_step=row_step)
    ref, cnt = _oracle_frames[key]
    rows = list(range(0, h, row_step))
    bad = np.argwhere(gpu[rows] != ref[rows])
    assert bad.size == 0, f"{what}: {len(bad)} pixels differ, first at {tuple(bad[0])}"
    if row_step == 1:
        for k in ("rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests", "plane_tests", "triangle_tests"):
            assert st[k] == cnt[k], (what, k, st[k], cnt[k])
    return gpu


@pytest.mark.parametrize("knob,value", [("RT_HIP_GRID", "0"), ("RT_HIP_GRID_DENSITY", "0.05"), ("RT_HIP_GRID_DENSITY", "64"),
                                        ("RT_HIP_GRID_REGION", "0"), ("RT_HIP_GRID_REGION", "8"),
                                        ("RT_HIP_GRID_TIER_D1", "0"), ("RT_HIP_GRID_TIER_D1", "0.3"), ("RT_HIP_GRID_TIER_D1", "1.5")])
def test_any_grid_gives_the_same_frame(oracle, monkeypatch, knob, value):
    """No grid at all, one cell per twenty spheres, 64 cells per sphere, a served region that ends at
    the spheres' own extent (most reflection rays then start outside it and evaluate every sphere, in
    the same waves as rays that walk the grid), one that reaches far beyond the room; without the second, shorter
    set of lists for the near part of a walk, with one that ends after 0.3 or 1.5 units of length."""
    cam = Camera(position=(0.3, -0.4, -0.5), direction=(0.1, -0.05, 1.0))
    fs1, fs64 = scene_1k().flatten(), scene_64k().flatten()
    base1 = frames_equal_oracle(oracle, fs1, cam, 256, 144, 4, "1k spheres, default grid", key="1k")
    base64 = frames_equal_oracle(oracle, fs64, Camera(), 96, 54, 8, "64k spheres, default grid", key="64k")
    monkeypatch.setenv(knob, value)
    assert np.array_equal(base1, frames_equal_oracle(oracle, fs1, cam, 256, 144, 4, f"1k spheres, {knob}={value}", key="1k"))
    assert np.array_equal(base64, frames_equal_oracle(oracle, fs64, Camera(), 96, 54, 8, f"64k spheres, {knob}={value}", key="64k"))


def test_axis_parallel_reflection_rays(oracle):
    """A mirror wall seen head-on: the centre column's and row's reflection rays have one or two
    direction components that are exactly zero (reciprocals of zero in the walk), and they come back
    through a lattice of spheres whose centres lie exactly on cell-aligned coordinates."""
    s = Scene()
    mirror = Material(0.0, specular=1.0, ambientColor=(0, 0, 0, 1), diffuseColor=(0, 0, 0, 1))
    s.Add(Plane(0.0, 0.0, -1.0, 6.0, mirror))                      # z = 6, faces the camera
    s.Add(Plane(0.0, 0.0, 1.0, 3.0, Material(1.0, ambientColor=(0.2, 0.1, 0.1, 1.0))))   # z = -3, behind it
    for k in range(5):
        for j in range(5):
            for i in range(5):
                if (i, j) == (2, 2) and k < 4:
                    continue   # a clear line of sight along the axis, up to the last layer
                s.Add(Sphere(-1.0 + 0.5 * i, -1.0 + 0.5 * j, 1.0 + 0.5 * k, 0.125 + 0.01 * ((i + j + k) % 3),
                             mirror if (i + j + k) % 4 == 0 else Material(1.0)))
    s.Add(Light(0.0, -3.0, 0.0))
    s.Add(Light(2.0, 1.0, -1.0))
    fs = s.flatten()
    for cam in (Camera(position=(0.0, 0.0, -2.0), direction=(0.0, 0.0, 1.0), fov=50.0),
                Camera(position=(0.0, 0.0, 8.5), direction=(0.0, 0.0, -1.0), fov=70.0),      # from behind the mirror's plane
                Camera(position=(-3.0, 0.0, 2.0), direction=(1.0, 0.0, 0.0), fov=40.0)):     # along x through the lattice
        frames_equal_oracle(oracle, fs, cam, 129, 97, 5, f"lattice from {cam.position}")


def test_grid_on_the_full_size_frame(oracle):
    """3840x2160: the first level holds more than a million reflection rays, which takes the tiles of 512
    records; every 120th row against the CPU restatement."""
    frames_equal_oracle(oracle, scene_1k().flatten(), Camera(), 3840, 2160, 4, "1k spheres at 4K", row_step=120)


def test_clustered_spheres_and_an_outlier(oracle):
    """Two tight clusters and one sphere far away: the grid's extent is set by the outlier, nearly all
    spheres share a handful of cells, and the rays cross long stretches of empty cells."""
    s = synthetic_scene(0)
    rng = np.random.default_rng(11)
    mirror = Material(0.0, specular=1.0, ambientColor=(0, 0, 0, 1), diffuseColor=(0, 0, 0, 1))
    for cx, cz in ((-1.5, 2.0), (1.8, 2.8)):
        for i in range(150):
            p = rng.normal(0.0, 0.25, 3)
            s.Add(Sphere(cx + p[0], -1.0 + p[1], cz + p[2], float(rng.uniform(0.01, 0.06)), mirror if i % 2 else Material(1.0)))
    s.Add(Sphere(300.0, -200.0, 900.0, 40.0, Material(1.0)))
    frames_equal_oracle(oracle, s.flatten(), Camera(), 192, 108, 5, "clusters and an outlier")


@pytest.mark.parametrize("below", ["0", "3000", "200000"])
def test_light_terms_ahead_with_many_lights(oracle, monkeypatch, below):
    """The reflecting records' light terms computed ahead of their children on scenes of more than four lights:
    for every level, for the levels of fewer than 3000 reflecting records, of fewer than 200000 (by itself the
    library takes 65536); long queues one record per lane, short ones with the lights spread over waves."""
    monkeypatch.setenv("RT_HIP_EARLY_LIGHTS_BELOW", below)
    monkeypatch.setenv("RT_HIP_PIPELINE", "levels")
    for profiling in (1, 0):
        with RenderContext() as c:
            c.set_profiling(profiling)
            fs, cam, w, h, depth, key = scene_64k().flatten(), Camera(), 96, 54, 8, "64k"
            c.set_scene(fs)
            gpu = c.render(cam, w, h, depth)
            if key not in _oracle_frames:
                _oracle_frames[key] = oracle.render(fs, cam, w, h, max_depth=depth)
            assert np.array_equal(gpu, _oracle_frames[key][0]), f"64k spheres, profiling {profiling}"
            s = synthetic_scene(300, radius=(0.05, 0.3), seed=5)
            s.lights.clear()
            for i in range(9):
                s.Add(Light(-2.0 + 0.5 * i, -3.0 + 0.3 * i, 0.5 + 0.3 * i))
            fs = s.flatten()
            c.set_scene(fs)
            gpu = c.render(Camera(), 640, 360, 4)     # level 0 is a long queue, the deep levels short ones
            if "9 lights" not in _oracle_frames:
                _oracle_frames["9 lights"] = oracle.render(fs, Camera(), 640, 360, max_depth=4, row_step=24)
            rows = list(range(0, 360, 24))
            assert np.array_equal(gpu[rows], _oracle_frames["9 lights"][0][rows]), f"9 lights, profiling {profiling}"


@pytest.mark.parametrize("knobs", [{"RT_HIP_LIGHT_TIERS": "1"}, {"RT_HIP_LIGHT_TIER_D1": "0.4", "RT_HIP_LIGHT_TIER_RATIO": "2.2"},
                                   {"RT_HIP_LIGHT_TIER_D1": "2.5", "RT_HIP_LIGHT_TIER_RATIO": "1.15"}])
def test_any_tiering_of_the_lights_tables_gives_the_same_frame(oracle, monkeypatch, knobs):
    """The lights' tables are built in tiers by the light ray's length (a short ray's table lists the spheres with
    a smaller rounding margin): one tier only, tiers so short that most rays fall into the later ones (0.4, 0.88,
    1.9), tiers close together (2.5, 2.9, 3.3: neighbouring lanes keep changing tier)."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    cam = Camera(position=(0.3, -0.4, -0.5), direction=(0.1, -0.05, 1.0))
    frames_equal_oracle(oracle, scene_1k().flatten(), cam, 256, 144, 4, f"1k spheres, {knobs}", key="1k")
    frames_equal_oracle(oracle, scene_64k().flatten(), Camera(), 96, 54, 8, f"64k spheres, {knobs}", key="64k")


def test_tiers_change_nothing_on_the_full_size_frames(monkeypatch):
    """The 64k-sphere, 16-light scene at 1920x1080 with 8 bounces (52 M rays) and three other views of it, as the
    library builds its tables (four tiers of the lights' tables, two sets of grid lists) and with one tier and one
    set: the same frames, pixel for pixel, and the same ray counts.  (The untiered tables are the ones the CPU
    restatement pins at the sizes it can render.)"""
    fs = scene_64k().flatten()
    cams = [Camera(), Camera(position=(2.5, -2.0, 0.2), direction=(-0.6, 0.1, 1.0)),
            Camera(position=(0.0, -3.6, 1.9), direction=(0.0, 1.0, 0.05), fov=70.0),
            Camera(position=(-3.4, 0.4, 3.3), direction=(1.0, -0.3, -0.6), fov=110.0)]

    def frames():
        out = []
        with RenderContext() as c:
            c.set_scene(fs)
            for cam in cams:
                out.append((c.render(cam, 1920, 1080, 8).copy(), c.stats()["rays_shadow"], c.stats()["rays_secondary"]))
        return out

    tiered = frames()
    monkeypatch.setenv("RT_HIP_LIGHT_TIERS", "1")
    monkeypatch.setenv("RT_HIP_GRID_TIER_D1", "0")
    plain = frames()
    for k, ((a, sa, ra), (b, sb, rb)) in enumerate(zip(tiered, plain)):
        assert (sa, ra) == (sb, rb), f"view {k}: ray counts differ"
        assert np.array_equal(a, b), f"view {k}: {np.count_nonzero(a != b)} pixels differ"
        assert len(np.unique(a)) > 1000   # a real picture


@pytest.mark.parametrize("knobs", [{}, {"RT_HIP_LIGHT_TIER_D1": "0.5", "RT_HIP_LIGHT_TIER_RATIO": "2", "RT_HIP_GRID_TIER_D1": "0.7"}])
def test_adversarial_scenes_with_the_grid(oracle, monkeypatch, knobs):
    """The scenes of test_gpu_adversarial.py (a few hundred spheres each: by itself the library would not
    build a grid for them) with the grid forced: large offsets and scales, tangent pencils, a far camera,
    degenerate spheres, lights and cameras, hit points far outside the served region.  Once as the library
    builds its tables for them, once with tiers of the lights' tables at ray lengths of 0.5, 1 and 2 and the
    grid's short lists up to 0.7, whatever the scene's scale (a hundred times larger: nearly every ray is
    longer than every tier; fifty times smaller: the region's own bound is the tighter one)."""
    import importlib.util
    import os
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    spec = importlib.util.spec_from_file_location("adversarial_scenes", os.path.join(os.path.dirname(__file__), "test_gpu_adversarial.py"))
    adv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(adv)
    with RenderContext() as c:
        c.set_profiling(True)
        for scale, offset in [(1.0, (1e3, -2e3, 5e2)), (1.0, (-1e5, 3e4, 1e5)), (100.0, (0.0, 0.0, 0.0)), (0.02, (1e3, 1e3, 1e3))]:
            adv.test_scaled_and_translated_scenes(c, oracle, scale, offset)
        adv.test_camera_far_from_the_cluster(c, oracle)
        adv.test_degenerate_spheres_in_a_valid_scene(c, oracle)
        adv.test_lights_in_awkward_places(c, oracle)
        adv.test_open_scene_with_far_hits(c, oracle)
        adv.test_enclosing_sphere_and_many_lights(c, oracle)


def test_grid_through_the_entry_point_that_allocates_nothing(oracle):
    """The grid walk with the ray queues in the library's fixed scratch array (rt_render_rows_device_noalloc):
    the same pixels as through the allocating entry point, on the 64k-sphere scene and a 1k-sphere one."""
    import torch
    stream = torch.cuda.Stream()
    for fs, w, h, depth in ((scene_64k().flatten(), 128, 72, 4), (scene_1k().flatten(), 200, 120, 3)):
        with RenderContext() as c:
            c.set_scene(fs)
            want = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
            got = torch.full((h, w), -2, dtype=torch.int32, device="cuda")
            c.render_rows_device(Camera(), w, h, 0, h, depth, want.data_ptr(), stream.cuda_stream)
            c.frame_status()
            c.render_rows_device_noalloc(Camera(), w, h, 0, h, depth, got.data_ptr(), stream.cuda_stream)
            c.frame_status()
            assert torch.equal(got, want)
