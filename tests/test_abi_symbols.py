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
