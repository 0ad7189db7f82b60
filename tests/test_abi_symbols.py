"""The C-ABI library loads and exports exactly what include/rt_hip.h declares (no GPU)."""
import ctypes
import os
import re

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert declared_functions() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    for name in declared_functions():
        assert getattr(lib, name) is not None, name
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    assert lib.rt_api_version() == int(re.search(r"#define RT_API_VERSION (\d+)", header).group(1)) == 7


def test_struct_sizes_match_the_header_layout():
    assert ctypes.sizeof(abi.rt_camera) == 40
    assert ctypes.sizeof(abi.rt_scene) == 8 + 16 * 8
    assert ctypes.sizeof(abi.rt_stats) == 13 * 8 + 6 * 8 + 8


def test_no_device_is_an_error_not_a_crash():
    lib = abi.load_library()
    if lib.rt_device_count() > 0:
        return
    ctx = ctypes.c_void_p()
    assert lib.rt_create(ctypes.byref(ctx), None, 0) == -2
    assert b"HIP device" in lib.rt_last_error(None)
    lib.rt_destroy(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"


def test_csharp_binding_declares_exactly_the_exports():
    """host/RtHip.cs (the P/Invoke file the C# host adds): every DllImport names an export of librt_hip.so and
    takes as many parameters as the header's declaration, and no export is missing from it."""
    text = open(os.path.join(ROOT, "host", "RtHip.cs")).read()
    decls = re.findall(r"\[DllImport\(Lib\)\]\s*public static extern\s+[\w\*]+\s+(rt_\w+)\((.*?)\);", text, flags=re.S)
    found = {}
    for name, params in decls:
        params = params.strip()
        found[name] = 0 if not params else len([p for p in params.split(",") if p.strip()])
    assert sorted(found) == sorted(abi.EXPORTS)
    lib = abi.load_library()
    for name, count in found.items():
        assert getattr(lib, name) is not None, name
        assert count == len(abi.EXPORTS[name][1]), f"{name}: {count} parameters in RtHip.cs, {len(abi.EXPORTS[name][1])} in the header"
    # the struct layouts the C# side states
    assert "rt_camera (40 bytes)" in text and ctypes.sizeof(abi.rt_camera) == 40
    assert "rt_scene (136 bytes)" in text and ctypes.sizeof(abi.rt_scene) == 136
    assert "rt_stats (160 bytes)" in text and ctypes.sizeof(abi.rt_stats) == 160
    assert "rt_debug_ray (56 bytes)" in text and ctypes.sizeof(abi.rt_debug_ray) == 56


def test_host_patches_are_unified_diffs_of_the_files_they_name():
    """host/*.patch: the edits to the reference's host files (apply with `patch --binary -p1`: the files have
    CRLF line ends).  Every hunk header is well formed and the Render() hunk replaces the per-pixel loop."""
    for name in ("MyApplication", "objects", "template"):
        lines = open(os.path.join(ROOT, "host", f"{name}.patch"), newline="").read().split("\n")
        assert lines[0].startswith(f"--- a/{name}.cs") and lines[1].startswith(f"+++ b/{name}.cs")
        assert any(re.match(r"@@ -\d+(,\d+)? \+\d+(,\d+)? @@", ln) for ln in lines)
    body = open(os.path.join(ROOT, "host", "MyApplication.patch"), newline="").read()
    assert "-            Parallel.For(0, h, y =>" in body and "+            gpuRenderer.Render(camera, screen);" in body
