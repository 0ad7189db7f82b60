"""ctypes view of include/rt_hip.h and the loader of librt_hip.so.

The structs mirror the header field for field.  The loader fails loudly when the
HIP library has not been built: there is no CPU fallback behind this package.
"""
from __future__ import annotations

import ctypes as C
import os

RT_PRIM_SPHERE = 0
RT_PRIM_PLANE = 1
RT_PRIM_TRIANGLE = 2
RT_REFERENCE_MAX_DEPTH = 5
RT_MAX_DEPTH = 16
RT_MAX_TEXTURES = 8
RT_NOALLOC_SCRATCH_BYTES = 32 << 20

RT_OK = 0
STATUS_NAMES = {
    0: "RT_OK",
    -1: "RT_ERR_INVALID_ARG",
    -2: "RT_ERR_NO_DEVICE",
    -3: "RT_ERR_OUT_OF_MEMORY",
    -4: "RT_ERR_HIP",
    -5: "RT_ERR_NO_SCENE",
    -6: "RT_ERR_BAD_TEXTURE",
    -7: "RT_ERR_LIMIT",
    -8: "RT_ERR_STALLED",
}

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)


class rt_camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("direction", C.c_float * 3),
        ("up", C.c_float * 3),
        ("fov_deg", C.c_float),
    ]


class rt_scene(C.Structure):
    _fields_ = [
        ("num_prims", C.c_int32),
        ("num_lights", C.c_int32),
        ("prim_types", _i),
        ("prim_positions", _f),
        ("prim_normals", _f),
        ("prim_diffuse", _f),
        ("prim_glossy", _f),
        ("prim_ambient", _f),
        ("prim_specular", _f),
        ("prim_radius", _f),
        ("prim_d", _f),
        ("prim_v0", _f),
        ("prim_v1", _f),
        ("prim_v2", _f),
        ("prim_checkerboard", _i),
        ("prim_texture", _i),
        ("light_positions", _f),
        ("light_colors", _f),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("rays_secondary", C.c_uint64),
        ("shaded_hits", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("plane_tests", C.c_uint64),
        ("triangle_tests", C.c_uint64),
        ("exact_candidates", C.c_uint64),
        ("filter_ray_tests", C.c_uint64),
        ("filter_origin_terms", C.c_uint64),
        ("primary_filter_ray_tests", C.c_uint64),
        ("primary_exact_candidates", C.c_uint64),
        ("shadow_candidates", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("primary_ms", C.c_double),
        ("trace_ms", C.c_double),
        ("shade_ms", C.c_double),
        ("gather_ms", C.c_double),
        ("copy_ms", C.c_double),
        ("num_devices", C.c_int32),
        ("reserved", C.c_int32),
    ]


class rt_debug_ray(C.Structure):
    _fields_ = [
        ("direction", C.c_float * 3),
        ("prim", C.c_int32),
        ("hit", C.c_float * 3),
        ("t", C.c_float),
        ("normal", C.c_float * 3),
        ("reflected", C.c_float * 3),
    ]


# Every symbol include/rt_hip.h declares: name -> (restype, argtypes).
EXPORTS = {
    "rt_api_version": (C.c_int, []),
    "rt_device_count": (C.c_int, []),
    "rt_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_set_scene": (C.c_int, [C.c_void_p, C.POINTER(rt_scene)]),
    "rt_set_texture": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "rt_render": (
        C.c_int,
        [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "rt_render_rows_device": (
        C.c_int,
        [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_void_p, C.c_void_p],
    ),
    "rt_render_rows_device_noalloc": (
        C.c_int,
        [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_void_p, C.c_void_p],
    ),
    "rt_render_bands_device": (
        C.c_int,
        [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
         C.c_void_p, C.c_void_p],
    ),
    "rt_band_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rt_frame_status": (C.c_int, [C.c_void_p]),
    "rt_set_device_share": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_reload_tuning": (C.c_int, [C.c_void_p]),
    "rt_debug_rays": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(rt_debug_ray), C.POINTER(C.c_int)]),
    "rt_fxaa_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rt_pin_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "rt_unpin_pixels": (C.c_int, [C.c_void_p]),
    "rt_alloc_pixels": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "rt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(rt_stats)]),
    "rt_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
}

LIB_NAME = "librt_hip.so"


def library_path() -> str:
    """The in-tree library; RT_HIP_LIBRARY names another build of it (A/B timing of variants)."""
    override = os.environ.get("RT_HIP_LIBRARY")
    if override:
        return os.path.abspath(override)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)


_lib = None


def _preload_torch_hip_runtime() -> None:
    """Keep ONE HIP runtime in the process.

    PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as the system one).
    If librt_hip.so pulled in /opt/rocm's copy first and torch loaded its own afterwards,
    the process would hold two HIP/HSA runtimes and the second finds no GPU.  Loading
    torch's copy (when torch is installed; torch itself is not imported) before
    librt_hip.so makes the dynamic loader resolve both to the same runtime.  Hosts
    without torch (the C# application) simply use the system runtime; RT_HIP_SYSTEM_RUNTIME=1
    asks for the same in a Python process that does not import torch (the two runtimes differ: the
    wheel's copies device memory to the host with a shader kernel, the system's with the DMA engines).
    """
    import importlib.util
    import sys
    if os.environ.get("RT_HIP_SYSTEM_RUNTIME") and "torch" not in sys.modules:
        return   # a process that never imports torch (like the C# host): the system's runtime, as librt_hip.so is linked
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load_library() -> C.CDLL:
    """Load librt_hip.so and bind every exported symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
        )
    _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (restype, argtypes) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
