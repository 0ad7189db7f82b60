"""ctypes wrapper of the CPU parity oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package raytracer_amd never does.  Parity unpinned: see rt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

from raytracer_amd import abi
from raytracer_amd.scene import Camera, FlatScene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")

MODE_MEMOISED = 0
MODE_LITERAL = 1


class orc_texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class orc_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_shadow", "rays_secondary", "rays_secondary_dead", "shaded_hits",
        "sphere_tests", "plane_tests", "triangle_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_f3 = C.c_float * 3
_f4 = C.c_float * 4
_lib = None


def build(force: bool = False) -> str:
    """Compile librt_oracle.so with gcc through oracle/Makefile."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("rt_oracle.c", "rt_oracle.h")):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        S, Cam, T = C.POINTER(abi.rt_scene), C.POINTER(abi.rt_camera), C.POINTER(orc_texture)
        L.orc_ray_direction.argtypes = [Cam, C.c_int, C.c_int, C.c_int, C.c_int, _f3]
        L.orc_ray_direction.restype = None
        L.orc_intersect.argtypes = [S, _f3, _f3, C.POINTER(C.c_float), _f3]
        L.orc_intersect.restype = C.c_int
        L.orc_in_shadow.argtypes = [S, _f3, _f3, C.c_float, C.c_float]
        L.orc_in_shadow.restype = C.c_int
        L.orc_prim_intersect.argtypes = [S, C.c_int, _f3, _f3, C.POINTER(C.c_float), _f3]
        L.orc_prim_intersect.restype = C.c_int
        L.orc_shade.argtypes = [S, T, C.c_int, C.c_int, _f3, _f3, C.c_int, C.c_int, C.c_int, _f4]
        L.orc_shade.restype = None
        L.orc_mix_color.argtypes = [_f4]
        L.orc_mix_color.restype = C.c_int32
        L.orc_texture_coords.argtypes = [_f3, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_texture_coords.restype = None
        L.orc_texel.argtypes = [_f3, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_texel.restype = None
        L.orc_checker_white.argtypes = [_f3]
        L.orc_checker_white.restype = C.c_int
        L.orc_render.argtypes = [S, T, C.c_int, Cam, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(orc_counters)]
        L.orc_render.restype = C.c_int
        L.orc_debug_ray.argtypes = [S, Cam, C.c_int, C.c_int, C.c_int, C.c_int, _f3, C.POINTER(C.c_float), _f3, _f3, _f3]
        L.orc_debug_ray.restype = C.c_int
        L.orc_fxaa.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_fxaa.restype = None
        L.orc_num_threads.argtypes = []
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _tex_array(textures: Optional[Sequence[np.ndarray]]):
    if not textures:
        return None, 0, []
    keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
    arr = (orc_texture * len(keep))()
    for i, t in enumerate(keep):
        assert t.ndim == 3 and t.shape[2] == 4, "texture must be HxWx4 uint8"
        arr[i].rgba8 = t.ctypes.data
        arr[i].height, arr[i].width = t.shape[0], t.shape[1]
    return arr, len(keep), keep


def ray_direction(cam: Camera, x: int, y: int, w: int, h: int) -> np.ndarray:
    out = _f3()
    lib().orc_ray_direction(C.byref(cam.as_struct()), x, y, w, h, out)
    return np.array(out[:], dtype=np.float32)


def intersect(fs: FlatScene, o, d) -> Tuple[int, float, np.ndarray]:
    t, hit = C.c_float(0), _f3()
    p = lib().orc_intersect(C.byref(fs.as_struct()), _f3(*map(float, o)), _f3(*map(float, d)),
                            C.byref(t), hit)
    return p, np.float32(t.value), np.array(hit[:], dtype=np.float32)


def debug_ray(fs: FlatScene, cam: Camera, x: int, y: int, w: int, h: int):
    """orc_debug_ray: (prim, direction, t, hit, normal, reflected); prim = -1 on a miss."""
    d, hit, n, r, t = _f3(), _f3(), _f3(), _f3(), C.c_float(0)
    p = lib().orc_debug_ray(C.byref(fs.as_struct()), C.byref(cam.as_struct()), x, y, w, h, d, C.byref(t), hit, n, r)
    f = lambda v: np.array(v[:], dtype=np.float32)
    return p, f(d), np.float32(t.value), f(hit), f(n), f(r)


def prim_intersect(fs: FlatScene, prim: int, o, d):
    t, hit = C.c_float(0), _f3()
    r = lib().orc_prim_intersect(C.byref(fs.as_struct()), prim, _f3(*map(float, o)),
                                 _f3(*map(float, d)), C.byref(t), hit)
    return bool(r), np.float32(t.value), np.array(hit[:], dtype=np.float32)


def in_shadow(fs: FlatScene, o, d, eps: float, dist: float) -> bool:
    return bool(lib().orc_in_shadow(C.byref(fs.as_struct()), _f3(*map(float, o)),
                                    _f3(*map(float, d)), eps, dist))


def shade(fs: FlatScene, prim: int, hit, view, depth: int = 0, max_depth: int = 5,
          mode: int = MODE_MEMOISED, textures=None) -> np.ndarray:
    arr, n, _keep = _tex_array(textures)
    out = _f4()
    lib().orc_shade(C.byref(fs.as_struct()), arr, n, prim, _f3(*map(float, hit)),
                    _f3(*map(float, view)), depth, max_depth, mode, out)
    return np.array(out[:], dtype=np.float32)


def mix_color(rgba) -> int:
    return int(lib().orc_mix_color(_f4(*map(float, rgba))))


def texture_coords(hit) -> Tuple[np.float32, np.float32]:
    u, v = C.c_float(0), C.c_float(0)
    lib().orc_texture_coords(_f3(*map(float, hit)), C.byref(u), C.byref(v))
    return np.float32(u.value), np.float32(v.value)


def texel(hit, tex_w: int, tex_h: int) -> Tuple[int, int]:
    tx, ty = C.c_int(0), C.c_int(0)
    lib().orc_texel(_f3(*map(float, hit)), tex_w, tex_h, C.byref(tx), C.byref(ty))
    return tx.value, ty.value


def checker_white(hit) -> bool:
    return bool(lib().orc_checker_white(_f3(*map(float, hit))))


def fxaa(frame: np.ndarray) -> np.ndarray:
    """orc_fxaa over an int32 [h, w] frame of 0x00RRGGBB pixels."""
    src = np.ascontiguousarray(frame, dtype=np.int32)
    out = np.empty_like(src)
    lib().orc_fxaa(src.ctypes.data, src.shape[1], src.shape[0], out.ctypes.data)
    return out


def num_threads() -> int:
    return int(lib().orc_num_threads())


def render(fs: FlatScene, cam: Camera, w: int, h: int, *, max_depth: int = 5,
           mode: int = MODE_MEMOISED, textures=None, rows: Tuple[int, int] = None,
           row_step: int = 1, nthreads: int = 0, out: Optional[np.ndarray] = None):
    """Render rows [rows[0], rows[1]) (every row_step-th) of a w x h frame.

    Returns (pixels int32 [h, w], counters dict); rows that were not rendered are zero
    (or keep the contents of ``out``).
    """
    arr, n, _keep = _tex_array(textures)
    if out is None:
        out = np.zeros((h, w), dtype=np.int32)
    assert out.dtype == np.int32 and out.shape == (h, w) and out.flags.c_contiguous
    r0, r1 = (0, h) if rows is None else rows
    cnt = orc_counters()
    rc = lib().orc_render(C.byref(fs.as_struct()), arr, n, C.byref(cam.as_struct()), w, h,
                          r0, r1, row_step, max_depth, mode, nthreads, out.ctypes.data,
                          C.byref(cnt))
    if rc != 0:
        raise ValueError("orc_render: invalid arguments")
    return out, cnt.as_dict()
