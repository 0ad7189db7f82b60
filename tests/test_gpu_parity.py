"""GPU parity: librt_hip.so (through the C ABI) against the CPU oracle, same inputs."""
import numpy as np
import pytest

from raytracer_amd import (Camera, Light, Material, Plane, Raytracer, RenderContext, RtError, Scene, Sphere,
                           Surface, Triangle, default_scene, scene_1k, synthetic_scene)
from tests.parity import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = RenderContext()
    c.set_profiling(True)   # the Intersect-call counters are compared with the restatement's
    yield c
    c.close()


def both(ctx, oracle, scene, cam, w, h, depth, textures=None):
    fs = scene.flatten() if isinstance(scene, Scene) else scene
    if textures:
        for i, t in enumerate(textures):
            ctx.set_texture(i, t)
    ctx.set_scene(fs)
    gpu = ctx.render(cam, w, h, depth)
    ref, counters = oracle.render(fs, cam, w, h, max_depth=depth, textures=textures)
    return gpu, ref, counters


def test_default_scene_template_resolution(ctx, oracle):
    """Config 1: the reference's default scene (cyan wall) at 1080x1080, depth 5."""
    gpu, ref, cnt = both(ctx, oracle, default_scene(), Camera(), 1080, 1080, 5)
    r = assert_parity(gpu, ref, "default 1080x1080")
    assert gpu[540, 540] == ref[540, 540] == 0x3E5A72
    st = ctx.stats()
    assert st["rays_primary"] == cnt["rays_primary"] == 1080 * 1080
    assert st["rays_shadow"] == cnt["rays_shadow"]
    assert st["rays_secondary"] == cnt["rays_secondary"]
    assert st["shaded_hits"] == cnt["shaded_hits"]
    assert st["sphere_tests"] == cnt["sphere_tests"]
    assert st["plane_tests"] == cnt["plane_tests"]
    assert st["triangle_tests"] == cnt["triangle_tests"]
    assert r["max_diff"] <= 1


def test_default_scene_textured_wall(ctx, oracle, wall_texture):
    gpu, ref, _ = both(ctx, oracle, default_scene(textured_wall=True), Camera(), 1080, 1080, 5,
                       textures=[wall_texture])
    assert_parity(gpu, ref, "textured 1080x1080")


def test_config2_1920x1080_primary_and_shadow_only(ctx, oracle):
    """Config 2: 16:9 (horizontal half-angle fov*aspect/2), recursion disabled."""
    gpu, ref, cnt = both(ctx, oracle, default_scene(), Camera(), 1920, 1080, 0)
    assert_parity(gpu, ref, "default 1920x1080 depth 0")
    assert cnt["rays_secondary"] == 0 and ctx.stats()["rays_secondary"] == 0


@pytest.mark.parametrize("depth", [0, 1, 2, 4, 8])
def test_1k_spheres_depths(ctx, oracle, depth):
    """Config 3's scene (1000 spheres + box) at a size the oracle finishes in seconds."""
    gpu, ref, cnt = both(ctx, oracle, scene_1k(), Camera(), 480, 270, depth)
    assert_parity(gpu, ref, f"1k spheres depth {depth}")
    st = ctx.stats()
    for k in ("rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests", "plane_tests"):
        assert st[k] == cnt[k], k


def test_ragged_sizes_and_single_pixel(ctx, oracle):
    for (w, h) in [(1, 1), (17, 9), (333, 127), (64, 1), (1, 64)]:
        gpu, ref, _ = both(ctx, oracle, default_scene(), Camera(), w, h, 5)
        assert np.array_equal(gpu, ref), (w, h)


def test_moved_and_rotated_camera(ctx, oracle):
    cams = [
        Camera(position=(0.3, -0.2, 0.5), direction=(0.0, 0.0, 1.0)),
        Camera(position=(-1.0, 0.2, 1.0), direction=(0.6, 0.0, 0.8), fov=70.0),
        Camera(position=(0.0, 0.0, 2.0), direction=(0.0, 0.0, 1.0)),  # inside the mirror sphere
        Camera(position=(1.0, -2.0, 3.0), direction=(-0.36, 0.48, -0.8), up=(0.0, 1.0, 0.0), fov=120.0),
    ]
    for cam in cams:
        gpu, ref, _ = both(ctx, oracle, default_scene(), cam, 320, 200, 5)
        assert_parity(gpu, ref, f"camera {cam}")


def test_many_lights_use_several_mask_words(ctx, oracle):
    s = synthetic_scene(40, seed=7)
    s.lights.clear()
    for i in range(37):
        s.Add(Light(-3.0 + 0.16 * i, -1.0 + 0.01 * i, 0.4 + 0.07 * i))
    gpu, ref, cnt = both(ctx, oracle, s, Camera(), 160, 120, 3)
    assert_parity(gpu, ref, "37 lights")
    assert ctx.stats()["rays_shadow"] == cnt["rays_shadow"]


@pytest.mark.parametrize("nl", [0, 1, 3, 5, 16])
def test_light_counts(ctx, oracle, nl):
    s = synthetic_scene(30, seed=11)
    s.lights.clear()
    for i in range(nl):
        s.Add(Light(-2.0 + 0.5 * i, -1.0, 0.5 + 0.2 * i))
    gpu, ref, _ = both(ctx, oracle, s, Camera(), 128, 96, 4)
    assert_parity(gpu, ref, f"{nl} lights")


def test_empty_scene_and_miss(ctx, oracle):
    s = Scene()
    s.Add(Light(0, 0, 0))
    gpu, ref, _ = both(ctx, oracle, s, Camera(), 40, 30, 5)
    assert not gpu.any() and np.array_equal(gpu, ref)
    # one small sphere: most primary rays miss and must write 0
    s.Add(Sphere(0.0, 0.0, 3.0, 0.4, Material(1.0)))
    gpu, ref, _ = both(ctx, oracle, s, Camera(), 200, 200, 5)
    assert np.array_equal(gpu, ref) and gpu.any() and (gpu == 0).any()


def test_ties_lowest_index_wins(ctx, oracle):
    """Coincident primitives of different classes and colours: the first one added wins."""
    red = Material(1.0, diffuseColor=(1, 0, 0, 1), ambientColor=(0.2, 0, 0, 1))
    blue = Material(1.0, diffuseColor=(0, 0, 1, 1), ambientColor=(0, 0, 0.2, 1))
    for order in (0, 1):
        s = Scene()
        tri = Triangle((-5.0, -5.0, 3.0), (5.0, -5.0, 3.0), (0.0, 5.0, 3.0), red)
        pl = Plane(0.0, 0.0, -1.0, 3.0, blue)
        for p in ((tri, pl) if order == 0 else (pl, tri)):
            s.Add(p)
        s.Add(Sphere(0.0, 0.0, 2.0, 0.25, blue if order == 0 else red))
        s.Add(Sphere(0.0, 0.0, 2.0, 0.25, red if order == 0 else blue))
        s.Add(Light(0.0, 0.0, 0.0))
        gpu, ref, _ = both(ctx, oracle, s, Camera(), 96, 96, 2)
        assert np.array_equal(gpu, ref), order


def test_row_ranges_tile_the_frame(ctx, oracle):
    import torch
    fs = scene_1k().flatten()
    ctx.set_scene(fs)
    cam, w, h, depth = Camera(), 200, 120, 3
    full = ctx.render(cam, w, h, depth)
    pieces = []
    for (y0, y1) in [(0, 37), (37, 38), (38, 120)]:
        buf = torch.empty((y1 - y0, w), dtype=torch.int32, device="cuda")
        ctx.render_rows_device(cam, w, h, y0, y1, depth, buf.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        pieces.append(buf.cpu().numpy())
    assert np.array_equal(np.concatenate(pieces, axis=0), full)
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth)
    assert_parity(full, ref, "row ranges")


def test_interleaved_bands_tile_the_frame(ctx):
    """rt_render_bands_device: the parts of an interleaved partition, put back, are the frame."""
    import torch
    from raytracer_amd import abi
    from raytracer_amd.distributed import band_rows_of
    lib = abi.load_library()
    fs = scene_1k().flatten()
    ctx.set_scene(fs)
    cam, depth = Camera(), 3
    for (w, h, band, parts) in [(200, 120, 16, 3), (97, 53, 16, 8), (64, 40, 4, 2), (33, 5, 16, 4)]:
        full = ctx.render(cam, w, h, depth)
        got = np.zeros_like(full)
        for part in range(parts):
            rows = band_rows_of(h, part, parts, band)
            assert lib.rt_band_rows(h, band, part, parts) == len(rows)
            buf = torch.full((max(len(rows), 1), w), -1, dtype=torch.int32, device="cuda")
            ctx.render_bands_device(cam, w, h, band, part, parts, depth, buf.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if rows:
                got[rows] = buf.cpu().numpy()[: len(rows)]
        assert np.array_equal(got, full), (w, h, band, parts)
    with pytest.raises(RtError):
        ctx.render_bands_device(cam, 8, 8, 0, 0, 1, depth, 0, 0)
    with pytest.raises(RtError):
        ctx.render_bands_device(cam, 8, 8, 16, 2, 2, depth, 0, 0)


def test_library_owned_framebuffer_takes_the_direct_copy(oracle):
    """rt_alloc_pixels: a page-locked framebuffer of the library's own (works where user memory cannot be
    page-locked): frames rendered into it - whole, a sub-range, a large frame rendered in row blocks, a
    frame from two device states - are the frames the staged path delivers; the buffer is replaced by the
    next one and released by rt_unpin_pixels."""
    fs, cam = default_scene().flatten(), Camera()
    with RenderContext() as c:
        c.set_scene(fs)
        plain = c.render(cam, 320, 200, 5)
        fb = c.alloc_pixels(2 * 200 + 1, 320)
        fb[:] = -1
        whole = c.render(cam, 320, 200, 5, out=fb[:200])
        assert np.array_equal(whole.reshape(200, 320), plain)
        part = c.render(cam, 320, 200, 5, out=fb[200:400])
        assert np.array_equal(part.reshape(200, 320), plain) and (fb[400] == -1).all()
        big = c.alloc_pixels(1600, 2048)           # replaces the first buffer; 3.3 M pixels: rendered in row blocks
        staged = c.render(cam, 2048, 1600, 3)
        direct = c.render(cam, 2048, 1600, 3, out=big)
        assert np.array_equal(direct.reshape(1600, 2048), staged)
        c.unpin_pixels()
        assert np.array_equal(c.render(cam, 320, 200, 5), plain)
        ref, _ = oracle.render(fs, cam, 320, 200, max_depth=5)
        assert np.array_equal(plain, ref)
    with RenderContext([0, 0]) as c2:               # two device states on the one device: the gather path
        c2.set_scene(fs)
        fb = c2.alloc_pixels(200, 320)
        assert np.array_equal(c2.render(cam, 320, 200, 5, out=fb).reshape(200, 320), plain)


def test_pinned_framebuffer_takes_the_direct_copy(oracle):
    """rt_pin_pixels: frames rendered into the registered array (whole or a sub-range) and into
    an unregistered one are the same frames."""
    fs, cam = default_scene().flatten(), Camera()
    with RenderContext() as c:
        c.set_scene(fs)
        plain = c.render(cam, 320, 200, 5)
        big = np.full(2 * 320 * 200 + 7, -1, dtype=np.int32)
        try:
            c.pin_pixels(big)
        except RtError as e:   # this system does not page-lock user memory: the staged path must go on working
            assert e.status == -4 and "hipHostRegister" in e.message
        whole = c.render(cam, 320, 200, 5, out=big[: 320 * 200])
        assert np.array_equal(whole.reshape(200, 320), plain)
        part = c.render(cam, 320, 200, 5, out=big[320 * 200 + 3: 2 * 320 * 200 + 3])
        assert np.array_equal(part.reshape(200, 320), plain) and (big[-4:] == -1).all()
        other = c.render(cam, 320, 200, 5)          # not inside the range: staged path
        assert np.array_equal(other, plain)
        c.unpin_pixels()
        assert np.array_equal(c.render(cam, 320, 200, 5, out=big[: 320 * 200]).reshape(200, 320), plain)
        ref, _ = oracle.render(fs, cam, 320, 200, max_depth=5)
        assert np.array_equal(plain, ref)
    screen = Surface(96, 64)
    rt = Raytracer(default_scene(), cam, screen)   # pins screen.pixels
    rt.Render()
    ref, _ = oracle.render(fs, cam, 96, 64, max_depth=5)
    assert np.array_equal(screen.image(), ref)
    rt.context.close()


def test_debug_rays_match_the_restatement(ctx, oracle):
    """rt_debug_rays: the rays of Raytracer.Debug (row 320, every 10th pixel of the 1080-wide
    template window, MyApplication.cs:459-478) and a few other rows / scenes, bit for bit."""
    for scene, cam, (w, h), row, step in [
        (default_scene(), Camera(), (1080, 1080), 320, 10),
        (default_scene(), Camera(position=(-1.0, 0.2, 1.0), direction=(0.6, 0.0, 0.8), fov=70.0), (640, 360), 7, 3),
        (scene_1k(), Camera(), (480, 270), 135, 1),
    ]:
        fs = scene.flatten()
        ctx.set_scene(fs)
        rays = ctx.debug_rays(cam, w, h, row, step)
        assert len(rays) == (w + step - 1) // step
        for i, r in enumerate(rays):
            p, d, t, hit, n, refl = oracle.debug_ray(fs, cam, i * step, row, w, h)
            assert r["prim"] == p, (i, r["prim"], p)
            assert np.array_equal(r["direction"], d)
            if p >= 0:
                assert r["t"] == t and np.array_equal(r["hit"], hit)
                assert np.array_equal(r["normal"], n) and np.array_equal(r["reflected"], refl)
    s = Scene()
    s.Add(Sphere(0.0, 0.0, 3.0, 0.4, Material(1.0)))
    ctx.set_scene(s.flatten())
    rays = ctx.debug_rays(Camera(), 100, 100, 50, 10)
    assert (rays["prim"] == -1).any() and (rays["prim"] == 0).any()


def test_fxaa_post_pass_matches_the_restatement(ctx, oracle):
    """rt_fxaa_device against the CPU restatement of screen_fxaa_fs.glsl:80-140, bit for bit,
    on rendered frames (edges, mirror, checkerboard), ragged sizes and a constant frame."""
    import torch
    stream = torch.cuda.Stream()
    frames = []
    ctx.set_scene(default_scene().flatten())
    frames.append(ctx.render(Camera(), 640, 360, 5))
    frames.append(ctx.render(Camera(position=(-1.0, 0.2, 1.0), direction=(0.6, 0.0, 0.8), fov=70.0), 333, 127, 5))
    ctx.set_scene(scene_1k().flatten())
    frames.append(ctx.render(Camera(), 960, 540, 4))
    frames.append(np.full((9, 5), 0x336699, dtype=np.int32))
    frames.append(np.random.default_rng(5).integers(0, 1 << 24, size=(64, 1), dtype=np.int32))
    for img in frames:
        h, w = img.shape
        src = torch.from_numpy(img).cuda()
        dst = torch.full_like(src, -1)
        ctx.fxaa_device(src.data_ptr(), dst.data_ptr(), w, h, stream.cuda_stream)
        stream.synchronize()
        ref = oracle.fxaa(img)
        assert np.array_equal(dst.cpu().numpy(), ref), (w, h)
    assert np.array_equal(oracle.fxaa(frames[3]), frames[3])   # nothing to smooth in a constant frame
    assert (oracle.fxaa(frames[0]) != frames[0]).any()
    with pytest.raises(RtError):
        ctx.fxaa_device(src.data_ptr(), src.data_ptr(), w, h, 0)


def test_errors_follow_the_convention(ctx):
    fresh = RenderContext()
    with pytest.raises(RtError) as e:
        fresh.render(Camera(), 8, 8, 5)
    assert e.value.status == -5 and "rt_set_scene" in e.value.message
    fresh.set_scene(default_scene(textured_wall=True).flatten())
    with pytest.raises(RtError) as e:
        fresh.render(Camera(), 8, 8, 5)
    assert e.value.status == -6
    fresh.set_scene(default_scene().flatten())
    with pytest.raises(RtError):
        fresh.render(Camera(), 0, 8, 5)
    with pytest.raises(RtError):
        fresh.render(Camera(), 8, 8, 99)
    fresh.close()


def test_config3_full_size_sampled_rows(ctx, oracle):
    """Config 3 at its full 3840x2160, 4 bounces: every 45th row against the oracle, and the
    frame-level invariants (every pixel written, counters consistent)."""
    fs = scene_1k().flatten()
    ctx.set_scene(fs)
    cam, w, h, depth = Camera(), 3840, 2160, 4
    gpu = ctx.render(cam, w, h, depth)
    st = ctx.stats()
    assert st["rays_primary"] == w * h
    assert st["rays_shadow"] == st["shaded_hits"] * fs.num_lights
    assert st["shaded_hits"] >= w * h  # closed box: every primary ray hits something
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth, row_step=45)
    rows = np.arange(0, h, 45)
    assert_parity(gpu[rows], ref[rows], "config 3 sampled rows")
    # idempotence: a second frame on the same context is identical
    assert np.array_equal(ctx.render(cam, w, h, depth), gpu)


def test_config4_resolution_row_blocks(ctx, oracle):
    """Config 4's 7680x4320 frame as eight 540-row blocks (the 8-GPU partition, rendered one
    after the other here): blocks tile the frame and sampled rows match the oracle."""
    import torch
    fs = scene_1k().flatten()
    ctx.set_scene(fs)
    cam, w, h, depth = Camera(), 7680, 4320, 4
    frame = torch.empty((h, w), dtype=torch.int32, device="cuda")
    for r in range(8):
        y0, y1 = 540 * r, 540 * (r + 1)
        ctx.render_rows_device(cam, w, h, y0, y1, depth, frame[y0:y1].data_ptr(), 0)
        torch.cuda.synchronize()
    gpu = frame.cpu().numpy()
    rows = np.arange(7, h, 540)  # one row inside every block
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth, rows=(7, h), row_step=540)
    assert_parity(gpu[rows], ref[rows], "config 4 row blocks")


def test_config5_64k_spheres_16_lights_depth8(ctx, oracle):
    """Config 5's scene (65536 spheres in 64 LDS chunks, 16 lights, 8 bounces) at a size the
    oracle finishes in seconds."""
    from raytracer_amd import scene_64k
    fs = scene_64k().flatten()
    gpu, ref, cnt = both(ctx, oracle, fs, Camera(), 64, 36, 8)
    assert_parity(gpu, ref, "64k spheres")
    st = ctx.stats()
    for k in ("rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests", "plane_tests"):
        assert st[k] == cnt[k], k


@pytest.mark.parametrize("scene_kind", ["grid16", "plain8"])
def test_binning_launches_left_out_on_the_last_frames_counts(monkeypatch, scene_kind):
    """Scenes whose deeper levels are put into bin order: the frame's last launch leaves the levels' record counts in
    page-locked memory, and the next frame of the same scene, size and depth leaves the binning launches of its
    short levels out (grid scenes: pushes those levels' records straight to their slots).  Only speed may depend
    on that: frames of one context in a row - sizes alternating, so that hints come and go and a level is binned in
    one frame and not in the next - are those of contexts that never use the hint (RT_HIP_BIN_HINT=0), and with the
    binning bound lowered so that every size has levels on both sides of it."""
    if scene_kind == "grid16":
        fs = synthetic_scene(3000, radius=(0.02, 0.08), seed=91, lights="grid16").flatten()   # grid walk, staged records
    else:
        sc = synthetic_scene(1000, seed=5)                                                     # filter loops, 8 lights
        sc.lights.clear()
        for i in range(8):
            sc.Add(Light(-2.0 + 0.6 * i, -1.0, 0.5 + 0.3 * i))
        fs = sc.flatten()
    cam, depth = Camera(), 5
    sizes = [(320, 180), (320, 180), (96, 54), (320, 180), (320, 180), (640, 360), (640, 360), (320, 180)]
    monkeypatch.setenv("RT_HIP_BIN_MIN", "2000")
    monkeypatch.setenv("RT_HIP_BIN_HINT", "0")
    with RenderContext() as c:
        c.set_scene(fs)
        plain = {size: c.render(cam, size[0], size[1], depth) for size in set(sizes)}
    monkeypatch.delenv("RT_HIP_BIN_HINT")
    with RenderContext() as c:
        c.set_scene(fs)
        for i, (w, h) in enumerate(sizes):
            assert np.array_equal(c.render(cam, w, h, depth), plain[(w, h)]), (i, w, h)
        c.set_scene(fs)   # (a new scene: the counts of the old one's frames say nothing)
        assert np.array_equal(c.render(cam, 320, 180, depth), plain[(320, 180)])


def test_a_wrong_hint_costs_speed_only(monkeypatch):
    """The 64k-sphere scene under 16 lights, 1280x720: looking at the back wall no record reflects, so the counts the
    frame leaves behind promise short queues everywhere; the next frame looks at the spheres, 300 000 reflecting
    records at level 0 - a queue that the launch with the lights spread over waves would take, had the host issued
    it.  The one launch it did issue shades the queue all the same: every frame of the sequence is the frame of a
    context that uses no hints."""
    from raytracer_amd import scene_64k
    fs = scene_64k().flatten()
    back, front = Camera(direction=(0.0, 0.0, -1.0)), Camera()
    w, h, depth = 1280, 720, 8
    monkeypatch.setenv("RT_HIP_BIN_HINT", "0")
    with RenderContext() as c:
        c.set_scene(fs)
        plain = {id(cam): c.render(cam, w, h, depth) for cam in (back, front)}
    monkeypatch.delenv("RT_HIP_BIN_HINT")
    assert plain[id(front)].any() and not np.array_equal(plain[id(front)], plain[id(back)])
    with RenderContext() as c:
        c.set_scene(fs)
        for i, cam in enumerate((back, back, front, front, back, front)):
            assert np.array_equal(c.render(cam, w, h, depth), plain[id(cam)]), i


@pytest.mark.parametrize("nspheres,nl,depth,size", [(1500, 2, 3, (160, 90)), (1025, 5, 2, (96, 54)),
                                                    (2048, 0, 2, (64, 36)), (132000, 2, 2, (40, 24))])
def test_chunked_scenes(ctx, oracle, nspheres, nl, depth, size):
    """Scenes of more than one LDS chunk of spheres: a ragged last chunk, exactly two chunks,
    and more chunks than one piece of chunk balls holds (128)."""
    s = synthetic_scene(nspheres, radius=(0.004, 0.05) if nspheres > 10000 else (0.02, 0.1), seed=23 + nspheres)
    s.lights.clear()
    for i in range(nl):
        s.Add(Light(-2.5 + 1.1 * i, -1.0 - 0.1 * i, 0.6 + 0.5 * i))
    w, h = size
    gpu, ref, cnt = both(ctx, oracle, s, Camera(), w, h, depth)
    assert_parity(gpu, ref, f"{nspheres} spheres, {nl} lights")
    st = ctx.stats()
    for k in ("rays_shadow", "rays_secondary", "shaded_hits", "sphere_tests", "plane_tests"):
        assert st[k] == cnt[k], k


@pytest.mark.parametrize("nspheres,nl,depth,size", [(3000, 16, 3, (96, 54)), (1500, 2, 3, (160, 90)),
                                                    (70000, 5, 2, (48, 27))])
def test_light_rays_may_stop_at_any_occluder(oracle, nspheres, nl, depth, size):
    """Without profiling a light ray of a multi-chunk scene stops looking once it has met an
    occluder: the pixels must not notice."""
    s = synthetic_scene(nspheres, radius=(0.01, 0.06), seed=41 + nspheres)
    s.lights.clear()
    for i in range(nl):
        s.Add(Light(-3.0 + 0.4 * i, -1.0 - 0.05 * i, 0.5 + 0.17 * i))
    fs, cam, (w, h) = s.flatten(), Camera(), size
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth)
    with RenderContext() as c:
        c.set_scene(fs)
        quick = c.render(cam, w, h, depth)          # profiling off: any occluder will do
        c.set_profiling(True)
        counted = c.render(cam, w, h, depth)
    assert_parity(quick, ref, "any-hit light rays")
    assert np.array_equal(quick, counted)


def test_large_frame_rendered_in_parts_with_overlapped_copy(oracle):
    """Without profiling rt_render renders a frame of Config 3's size as row blocks and copies
    each to the host while the next renders: same pixels and the same ray counts as the frame in
    one piece (profiling on), sampled rows against the oracle, and a second call identical."""
    fs = scene_1k().flatten()
    cam, w, h, depth = Camera(), 3840, 2160, 4
    with RenderContext() as c:
        c.set_scene(fs)
        parts = c.render(cam, w, h, depth)
        st_parts = c.stats()
        again = c.render(cam, w, h, depth)
        c.set_profiling(True)
        whole = c.render(cam, w, h, depth)
        st_whole = c.stats()
        small = c.render(cam, 320, 180, depth)   # back to a frame in one piece on the same context
        assert c.stats()["rays_primary"] == 320 * 180
    assert np.array_equal(parts, whole)
    assert np.array_equal(again, whole)
    for k in ("rays_primary", "rays_shadow", "rays_secondary", "shaded_hits"):
        assert st_parts[k] == st_whole[k], k
    rows = np.arange(0, h, 120)
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth, row_step=120)
    assert_parity(parts[rows], ref[rows], "frame rendered in parts")
    ref_small, _ = oracle.render(fs, cam, 320, 180, max_depth=depth)
    assert_parity(small, ref_small, "small frame after a frame in parts")


@pytest.mark.parametrize("parts,bands", [("2", ""), ("3", ""), ("5", ""), ("3", "0"), ("8", "")])
def test_row_blocks_of_any_number_and_form(monkeypatch, parts, bands):
    """rt_render's row blocks forced on a frame whose height is no multiple of the band (a short last band that
    falls to block 1 of 2, 1 of 3, ...): interleaved bands put in place by strided copies, contiguous blocks
    (RT_HIP_PART_BANDS=0), into pageable memory (staged, 2 MiB pieces) and into the library's page-locked
    framebuffer (one strided copy per block): the frame rendered in one piece, every time."""
    fs = scene_1k().flatten()
    cam, w, h, depth = Camera(position=(0.1, -0.2, -0.1)), 1280, 1000 + 9, 3
    with RenderContext() as c:
        c.set_scene(fs)
        c.set_profiling(2)
        whole = c.render(cam, w, h, depth)           # (profiling on: one piece)
        c.set_profiling(0)
        monkeypatch.setenv("RT_HIP_FRAME_PARTS", parts)
        if bands:
            monkeypatch.setenv("RT_HIP_PART_BANDS", bands)
        c.reload_tuning()
        staged = c.render(cam, w, h, depth)
        st = c.stats()
        assert np.array_equal(staged, whole)
        assert st["rays_primary"] == w * h
        fb = c.alloc_pixels(h + 2, w)
        fb[:] = -1
        direct = c.render(cam, w, h, depth, out=fb[1:h + 1])
        assert np.array_equal(direct.reshape(h, w), whole)
        assert (fb[0] == -1).all() and (fb[h + 1] == -1).all()


@pytest.mark.parametrize("size,depth", [((64, 48), 6), ((640, 400), 6), ((640, 400), 2), ((320, 200), 3)])
def test_reflecting_records_summed_path_by_path(oracle, monkeypatch, size, depth):
    """Frames whose light terms are all computed ahead sum their reflecting records in one launch, every pixel's
    path from its last reflecting record up (the default), instead of one launch per level
    (RT_HIP_PATH_SHADE=0), or both (=2: the paths down to level 1, then level 0's own launch).  An open scene -
    mirrors over a mirror floor, no walls - so that paths end in every way: a reflection ray that meets nothing,
    a child that does not reflect, the depth limit.  The same frame each way, and the restatement's."""
    mirror = Material(1.0, specular=0.8, diffuseColor=(0.9, 0.9, 1.0, 1.0))
    half = Material(1.0, specular=0.5, diffuseColor=(1.0, 0.6, 0.2, 1.0))
    matte = Material(1.0, diffuseColor=(0.3, 0.8, 0.4, 1.0))
    s = Scene()
    s.Add(Plane(0.0, -1.0, 0.0, 1.0, mirror))
    k = 0
    for ix in range(-3, 4):
        for iz in range(4):
            s.Add(Sphere(0.55 * ix, 0.55 - 0.25 * ((ix + iz) % 3), 1.6 + 0.6 * iz, 0.22, (mirror, half, matte)[k % 3]))
            k += 1
    s.Add(Light(1.0, -2.0, 0.5))
    s.Add(Light(-1.5, -1.0, 2.5))
    fs = s.flatten()
    cam, (w, h) = Camera(position=(0.0, -0.3, 0.0)), size
    frames = {}
    with RenderContext() as c:
        c.set_scene(fs)
        for knob in ("", "0", "2"):
            if knob:
                monkeypatch.setenv("RT_HIP_PATH_SHADE", knob)
            c.reload_tuning()
            frames[knob] = c.render(cam, w, h, depth)
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth)
    assert (ref == 0).any() and ref.any()
    assert np.array_equal(frames[""], frames["0"])
    assert np.array_equal(frames["2"], frames["0"])
    assert_parity(frames[""], ref, "paths summed in one launch")


@pytest.mark.parametrize("size,offset", [((1024, 768), 1), ((1024, 768), 3), ((3840, 2160), 1), ((333, 217), 2)])
def test_framebuffer_at_any_alignment(size, offset):
    """pixels_out is whatever address the host's array has (a managed int[] is 8-byte aligned at
    best): the chunked copy with its 16-byte streaming stores must fill exactly the frame, at
    any 4-byte alignment, for one chunk, several chunks and a frame rendered in parts."""
    fs = scene_1k().flatten()
    cam, (w, h), depth = Camera(), size, 2
    with RenderContext() as c:
        c.set_scene(fs)
        aligned = c.render(cam, w, h, depth)
        guard = np.full(w * h + 8, 0x5A5A5A5A, dtype=np.int32)
        out = guard[offset:offset + w * h]
        c.render(cam, w, h, depth, out=out)
    assert np.array_equal(out.reshape(h, w), aligned)
    assert np.all(guard[:offset] == 0x5A5A5A5A) and np.all(guard[offset + w * h:] == 0x5A5A5A5A)


def test_profiling_modes(oracle):
    """Mode 0: no counters, no times.  Mode 2: the same kernels bracketed by events.  Mode 1: the
    counting kernels, whose Intersect-call counts equal the restatement's.  Same pixels in all."""
    fs = scene_1k().flatten()
    cam, (w, h), depth = Camera(), (320, 200), 3
    ref, cnt = oracle.render(fs, cam, w, h, max_depth=depth)
    with RenderContext() as c:
        c.set_scene(fs)
        plain = c.render(cam, w, h, depth)
        st0 = c.stats()
        c.set_profiling(2)
        timed = c.render(cam, w, h, depth)
        st2 = c.stats()
        c.set_profiling(1)
        counted = c.render(cam, w, h, depth)
        st1 = c.stats()
        with pytest.raises(RtError):
            c.set_profiling(3)
    assert np.array_equal(plain, ref) and np.array_equal(timed, ref) and np.array_equal(counted, ref)
    for st in (st0, st1, st2):   # the ray counts never depend on the mode
        for k in ("rays_primary", "rays_shadow", "rays_secondary", "shaded_hits"):
            assert st[k] == cnt[k], k
    assert st0["kernel_ms"] == 0.0 and st2["kernel_ms"] > 0.0 and st2["trace_ms"] > 0.0 and st1["kernel_ms"] > 0.0
    assert st1["sphere_tests"] == cnt["sphere_tests"] and st1["plane_tests"] == cnt["plane_tests"]
    # without the counting kernels only the primary rays' share is known
    ns = int((fs.prim_types == 0).sum())
    assert st0["sphere_tests"] == st2["sphere_tests"] == w * h * ns


def test_pipeline_slots_alternate_streams_on_one_context(ctx, oracle):
    """FramePipeline on one rank: consecutive frames go through two slots (two streams, two
    frame buffers) of ONE render context; each slot's stream waits for the previous frame's
    kernels, so every frame is complete and a slot's frame survives the next frame."""
    import torch
    from raytracer_amd.distributed import FramePipeline
    fs = scene_1k().flatten()
    ctx.set_scene(fs)
    w, h, depth = 640, 360, 3
    pipe = FramePipeline(w, h, 0, 1, torch.device("cuda", 0), depth=2)
    cams = [Camera(position=(0.05 * i, 0.0, 0.0)) for i in range(6)]
    frames = [pipe.render_frame(ctx, cam, depth) for cam in cams]
    pipe.synchronize()
    assert frames[4].data_ptr() != frames[5].data_ptr() and frames[3].data_ptr() == frames[5].data_ptr()
    for i in (4, 5):   # the last frame of either slot
        ref, _ = oracle.render(fs, cams[i], w, h, max_depth=depth)
        assert_parity(frames[i].cpu().numpy(), ref, f"pipelined frame {i}")


@pytest.mark.parametrize("host_gather", ["1", "0"])
def test_two_device_states_split_rows_and_gather(oracle, monkeypatch, host_gather):
    """rt_create with several device entries (the same GPU several times on a one-GPU box): tall
    frames are split into interleaved bands of 16 rows and put in place with one strided copy per
    device (a short last band, exact bands, three devices), short frames into contiguous blocks.  Every
    device copies its rows to the host itself, into pageable memory through the staging frame and straight
    into a page-locked framebuffer; with RT_HIP_HOST_GATHER=0 the rows first meet in the first device's frame."""
    monkeypatch.setenv("RT_HIP_HOST_GATHER", host_gather)
    fs = default_scene().flatten()
    cam = Camera()
    with RenderContext([0, 0]) as two:
        two.set_profiling(True)
        two.set_scene(fs)
        for (w, h) in [(200, 150), (96, 128), (64, 33)]:
            gpu = two.render(cam, w, h, 5)
            ref, cnt = oracle.render(fs, cam, w, h, max_depth=5)
            assert np.array_equal(gpu, ref), (w, h)
            st = two.stats()
            assert st["rays_shadow"] == cnt["rays_shadow"] and st["sphere_tests"] == cnt["sphere_tests"]
        assert two.stats()["num_devices"] == 2
        assert two.stats()["rays_primary"] == 64 * 33
    with RenderContext([0, 0, 0]) as three:
        three.set_scene(fs)
        for (w, h) in [(120, 100), (120, 161)]:
            gpu = three.render(cam, w, h, 5)
            ref, _ = oracle.render(fs, cam, w, h, max_depth=5)
            assert np.array_equal(gpu, ref), (w, h)
            fb = three.alloc_pixels(h + 1, w)       # the direct path: every device's copy lands in the caller's frame
            fb[:] = -1
            assert np.array_equal(three.render(cam, w, h, 5, out=fb[:h]).reshape(h, w), ref), (w, h, "page-locked")
            assert (fb[h] == -1).all()
            three.unpin_pixels()
        # a frame tall enough for several strided copies per device on the staged path (2 MiB pieces)
        w, h = 1024, 1500
        gpu = three.render(cam, w, h, 2)
        ref, _ = oracle.render(fs, cam, w, h, max_depth=2, row_step=25)
        rows = list(range(0, h, 25))
        assert np.array_equal(gpu[rows], ref[rows])
    with RenderContext() as one:
        one.set_scene(fs)
        assert np.array_equal(one.render(cam, w, h, 2), gpu)


def test_devices_without_peer_access_hand_their_rows_over_through_the_host(oracle, monkeypatch):
    """The gather of a context whose other devices cannot write into the first one's memory: their
    rows travel through the page-locked staging frame (RT_HIP_NO_PEER=1 makes rt_create treat every
    other device that way).  Interleaved bands and the contiguous fallback, two and three devices."""
    monkeypatch.setenv("RT_HIP_NO_PEER", "1")
    monkeypatch.setenv("RT_HIP_HOST_GATHER", "0")   # (by default every device copies its rows to the host itself)
    fs = default_scene().flatten()
    cam = Camera(position=(0.3, 0.1, -0.2))
    with RenderContext([0, 0]) as two:
        two.set_scene(fs)
        for (w, h) in [(200, 150), (64, 33)]:
            gpu = two.render(cam, w, h, 5)
            ref, _ = oracle.render(fs, cam, w, h, max_depth=5)
            assert np.array_equal(gpu, ref), (w, h)
        assert two.stats()["num_devices"] == 2
    with RenderContext([0, 0, 0]) as three:
        three.set_scene(fs)
        gpu = three.render(cam, 120, 161, 5)
        ref, _ = oracle.render(fs, cam, 120, 161, max_depth=5)
        assert np.array_equal(gpu, ref)


def test_pipeline_with_a_context_per_slot_renders_frames_side_by_side(oracle):
    """FramePipeline built with one render context per slot: the slots' frames render at the same
    time on streams of their own (each context sized for a third of the device) and every frame
    is the frame its camera asks for."""
    import torch
    from raytracer_amd.distributed import FramePipeline
    fs = scene_1k().flatten()
    w, h, depth = 480, 270, 3
    ctxs = [RenderContext() for _ in range(3)]
    for c in ctxs:
        c.set_scene(fs)
        c.set_device_share(3)
    pipe = FramePipeline(w, h, 0, 1, torch.device("cuda", 0), depth=3, contexts=ctxs)
    cams = [Camera(position=(0.04 * i, -0.02 * i, 0.0)) for i in range(9)]
    frames = [pipe.render_frame(None, cam, depth) for cam in cams]
    pipe.synchronize()
    for c in ctxs:
        c.frame_status()
    assert len({f.data_ptr() for f in frames}) == 3
    for i in (6, 7, 8):   # the last frame of every slot
        ref, _ = oracle.render(fs, cams[i], w, h, max_depth=depth)
        assert_parity(frames[i].cpu().numpy(), ref, f"frame {i} of three in flight")
    with pytest.raises(RtError):
        ctxs[0].set_device_share(0)
    for c in ctxs:
        c.close()
