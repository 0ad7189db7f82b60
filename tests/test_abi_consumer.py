"""include/rt_hip.h as a C99 host sees it: tests/abi_consumer.c is compiled with
`gcc -std=c99 -Wall -Werror -pedantic`, linked against librt_hip.so, and renders a frame."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from raytracer_amd import Camera, default_scene
from raytracer_amd.abi import library_path
from raytracer_amd.scene import FlatScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_consumer(tmp_path) -> str:
    exe = str(tmp_path / "abi_consumer")
    libdir = os.path.dirname(library_path())
    rocm_lib = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_consumer.c"), "-o", exe, "-L", libdir, "-lrt_hip",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm_lib}", f"-Wl,-rpath-link,{rocm_lib}"]
    subprocess.run(cmd, check=True)
    return exe


def write_scene(path, fs: FlatScene) -> None:
    fs.as_struct()   # makes every array contiguous and of the ABI's element type
    with open(path, "wb") as f:
        np.array([fs.num_prims, fs.num_lights], dtype=np.int32).tofile(f)
        for name in ("prim_types", "prim_positions", "prim_normals", "prim_diffuse", "prim_glossy", "prim_ambient",
                     "prim_specular", "prim_radius", "prim_d", "prim_v0", "prim_v1", "prim_v2", "prim_checkerboard",
                     "prim_texture", "light_positions", "light_colors"):
            getattr(fs, name).tofile(f)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_header_compiles_as_strict_c99_and_layouts_hold(tmp_path):
    """Compile and link only: the layout checks in abi_consumer.c are compile-time."""
    assert os.path.exists(build_consumer(tmp_path))


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_c_host_renders_the_default_scene(tmp_path, oracle):
    exe = build_consumer(tmp_path)
    fs = default_scene().flatten()
    w, h, depth = 96, 96, 5
    ref, _ = oracle.render(fs, Camera(), w, h, max_depth=depth)
    golden = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))
    assert np.array_equal(golden["default_cyan_96"], ref)   # the committed frame of this very configuration
    write_scene(tmp_path / "scene.bin", fs)
    ref.astype(np.int32).tofile(tmp_path / "expected.raw")
    out = subprocess.run([exe, str(tmp_path / "scene.bin"), str(w), str(h), str(depth), str(tmp_path / "expected.raw")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 of 9216 pixels differ" in out.stdout
