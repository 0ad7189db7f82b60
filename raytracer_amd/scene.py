"""Host-side mirror of the reference's scene classes and the scene flatten.

The classes carry the names, fields and constructor arithmetic of the reference
(/root/reference/objects.cs:9-27,99-107,142-153,223-234,288-297 and
MyApplication.cs:259-275,339-350) so that code written against the reference reads
the same here.  ``Scene.flatten()`` is the dynamic-capacity form of the flatten in
MyApplication.Init (MyApplication.cs:142-176) and produces the arrays ``rt_scene``
points at.  All arithmetic that the reference does in FP32 is done with numpy
float32 scalars, operation by operation.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import abi

f32 = np.float32


def _v(*xs) -> np.ndarray:
    return np.array(xs, dtype=np.float32)


def _dot(a, b) -> np.float32:
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b) -> np.ndarray:
    return _v(
        f32(f32(a[1] * b[2]) - f32(a[2] * b[1])),
        f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
        f32(f32(a[0] * b[1]) - f32(a[1] * b[0])),
    )


def _normalized(a) -> np.ndarray:
    """OpenTK Vector3.Normalized(): scale = 1/Length, then three multiplies."""
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = f32(1.0) / np.sqrt(_dot(a, a), dtype=np.float32)
        return _v(f32(a[0] * scale), f32(a[1] * scale), f32(a[2] * scale))


class Material:
    """objects.cs:9-27."""

    def __init__(self, diffuse: float, *, diffuseColor=None, glossyColor=None, ambientColor=None,
                 specular: float = 0.0, useCheckerboard: bool = False):
        self.diffuse = f32(diffuse)
        self.specular = f32(specular)
        self.glossyColor = _v(1, 1, 1, 1) if glossyColor is None else _v(*glossyColor)
        self.diffuseColor = _v(1, 1, 1, 1) if diffuseColor is None else _v(*diffuseColor)
        if ambientColor is None:
            # ambientColor = diffuseColor * 0.1f * diffuse, evaluated with the ctor's
            # default diffuseColor (1,1,1,1) before any object initialiser runs.
            base = _v(1, 1, 1, 1)
            self.ambientColor = (base * f32(0.1)).astype(np.float32) * self.diffuse
            self.ambientColor = self.ambientColor.astype(np.float32)
        else:
            self.ambientColor = _v(*ambientColor)
        self.useCheckerboard = bool(useCheckerboard)


class Primitive:
    material: Material

    def __init__(self):
        self.position = _v(0, 0, 0)
        self.material = Material(1.0)


class Sphere(Primitive):
    """objects.cs:99-107."""

    def __init__(self, x, y, z, radius, material: Optional[Material] = None):
        super().__init__()
        self.position = _v(x, y, z)
        self.radius = f32(radius)
        if material is not None:
            self.material = material


class Plane(Primitive):
    """objects.cs:142-153: the constructor normalises (nx, ny, nz); plane n.p + d = 0."""

    def __init__(self, nx, ny, nz, d, material: Optional[Material] = None, texture: int = -1):
        super().__init__()
        self.normal = _normalized(_v(nx, ny, nz))
        self.d = f32(d)
        self.texture = int(texture)  # texture slot, -1 = none (objects.cs:146 holds a Texture)
        if material is not None:
            self.material = material


class Triangle(Primitive):
    """objects.cs:223-234."""

    def __init__(self, v0, v1, v2, material: Optional[Material] = None):
        super().__init__()
        self.v0, self.v1, self.v2 = _v(*v0), _v(*v1), _v(*v2)
        self.normal = _normalized(_cross(self.v1 - self.v0, self.v2 - self.v0))
        if material is not None:
            self.material = material


class Light:
    """objects.cs:288-297: colour is fixed at (2,2,2,1)."""

    def __init__(self, x, y, z, color=(2.0, 2.0, 2.0, 1.0)):
        self.position = _v(x, y, z)
        self.color = _v(*color)


@dataclass
class Camera:
    """MyApplication.cs:259-275 (fields read by the render loop only)."""

    position: Sequence[float] = (0.0, 0.0, 0.0)
    direction: Sequence[float] = (0.0, 0.0, 1.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fov: float = 90.0

    def as_struct(self) -> abi.rt_camera:
        c = abi.rt_camera()
        c.position[:] = [float(f32(v)) for v in self.position]
        c.direction[:] = [float(f32(v)) for v in self.direction]
        c.up[:] = [float(f32(v)) for v in self.up]
        c.fov_deg = float(f32(self.fov))
        return c


@dataclass
class FlatScene:
    """The arrays of rt_scene plus the ctypes struct that points into them."""

    prim_types: np.ndarray
    prim_positions: np.ndarray
    prim_normals: np.ndarray
    prim_diffuse: np.ndarray
    prim_glossy: np.ndarray
    prim_ambient: np.ndarray
    prim_specular: np.ndarray
    prim_radius: np.ndarray
    prim_d: np.ndarray
    prim_v0: np.ndarray
    prim_v1: np.ndarray
    prim_v2: np.ndarray
    prim_checkerboard: np.ndarray
    prim_texture: np.ndarray
    light_positions: np.ndarray
    light_colors: np.ndarray
    _struct: Optional[abi.rt_scene] = field(default=None, repr=False)

    FLOAT_FIELDS = ("prim_positions", "prim_normals", "prim_diffuse", "prim_glossy", "prim_ambient",
                    "prim_specular", "prim_radius", "prim_d", "prim_v0", "prim_v1", "prim_v2",
                    "light_positions", "light_colors")
    INT_FIELDS = ("prim_types", "prim_checkerboard", "prim_texture")

    @property
    def num_prims(self) -> int:
        return int(self.prim_types.shape[0])

    @property
    def num_lights(self) -> int:
        return int(self.light_positions.shape[0])

    def as_struct(self) -> abi.rt_scene:
        """ctypes rt_scene pointing at this object's arrays (kept alive by self)."""
        s = abi.rt_scene()
        s.num_prims = self.num_prims
        s.num_lights = self.num_lights
        for name in self.FLOAT_FIELDS:
            arr = np.ascontiguousarray(getattr(self, name), dtype=np.float32)
            setattr(self, name, arr)
            setattr(s, name, arr.ctypes.data_as(C.POINTER(C.c_float)))
        for name in self.INT_FIELDS:
            arr = np.ascontiguousarray(getattr(self, name), dtype=np.int32)
            setattr(self, name, arr)
            setattr(s, name, arr.ctypes.data_as(C.POINTER(C.c_int32)))
        self._struct = s
        return s

    def checksum(self) -> int:
        """FNV-1a over the raw bytes of every array, in field order."""
        h = 0xCBF29CE484222325
        for name in self.INT_FIELDS + self.FLOAT_FIELDS:
            for b in np.ascontiguousarray(getattr(self, name)).tobytes():
                h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h


class Scene:
    """MyApplication.cs:339-350."""

    def __init__(self):
        self.primitives: List[Primitive] = []
        self.lights: List[Light] = []

    def Add(self, obj) -> None:
        (self.lights if isinstance(obj, Light) else self.primitives).append(obj)

    def flatten(self) -> FlatScene:
        """MyApplication.cs:142-176 without the capacity of 10, plus the texture slot."""
        n, nl = len(self.primitives), len(self.lights)
        z3 = lambda: np.zeros((n, 3), np.float32)
        fs = FlatScene(
            prim_types=np.zeros(n, np.int32), prim_positions=z3(), prim_normals=z3(),
            prim_diffuse=np.zeros((n, 4), np.float32), prim_glossy=np.zeros((n, 4), np.float32),
            prim_ambient=np.zeros((n, 4), np.float32), prim_specular=np.zeros(n, np.float32),
            prim_radius=np.zeros(n, np.float32), prim_d=np.zeros(n, np.float32),
            prim_v0=z3(), prim_v1=z3(), prim_v2=z3(),
            prim_checkerboard=np.zeros(n, np.int32), prim_texture=np.full(n, -1, np.int32),
            light_positions=np.zeros((nl, 3), np.float32), light_colors=np.zeros((nl, 4), np.float32),
        )
        for i, l in enumerate(self.lights):
            fs.light_positions[i] = l.position
            fs.light_colors[i] = l.color
        for i, p in enumerate(self.primitives):
            m = p.material
            fs.prim_diffuse[i] = m.diffuseColor
            fs.prim_glossy[i] = m.glossyColor
            fs.prim_ambient[i] = m.ambientColor
            fs.prim_specular[i] = m.specular
            fs.prim_checkerboard[i] = 1 if m.useCheckerboard else 0
            if isinstance(p, Sphere):
                fs.prim_types[i] = abi.RT_PRIM_SPHERE
                fs.prim_positions[i] = p.position
                fs.prim_radius[i] = p.radius
            elif isinstance(p, Plane):
                fs.prim_types[i] = abi.RT_PRIM_PLANE
                fs.prim_normals[i] = p.normal
                fs.prim_d[i] = p.d
                fs.prim_texture[i] = p.texture
            elif isinstance(p, Triangle):
                fs.prim_types[i] = abi.RT_PRIM_TRIANGLE
                fs.prim_normals[i] = p.normal
                fs.prim_v0[i], fs.prim_v1[i], fs.prim_v2[i] = p.v0, p.v1, p.v2
            else:
                raise TypeError(f"unknown primitive {type(p).__name__}")
        return fs


# --------------------------------------------------------------------------------------
# Fixture: the reference's default scene, MyApplication.cs:73-140.
# --------------------------------------------------------------------------------------

def _materials():
    cyan = Material(1.0, diffuseColor=(0.75, 0.95, 1.0, 1.0), ambientColor=(0.02, 0.05, 0.1, 1.0))
    green = Material(1.0, diffuseColor=(0.8, 1.0, 0.8, 1.0), ambientColor=(0.06, 0.08, 0.06, 1.0))
    mirror = Material(0.0, specular=1.0, glossyColor=(1, 1, 1, 1), ambientColor=(0, 0, 0, 1),
                      diffuseColor=(0, 0, 0, 1))
    orange = Material(1.0, diffuseColor=(1.0, 0.85, 0.7, 1.0), ambientColor=(0.07, 0.05, 0.05, 1.0),
                      specular=0.5)
    pink = Material(1.0, diffuseColor=(1.0, 0.8, 0.95, 1.0), ambientColor=(0.07, 0.05, 0.05, 1.0))
    checker = Material(1.0, diffuseColor=(1, 1, 1, 1), ambientColor=(0.1, 0.1, 0.1, 1.0),
                       useCheckerboard=True)
    return dict(cyan=cyan, green=green, mirror=mirror, orange=orange, pink=pink, checker=checker)


def _box_planes(m, textured_wall: bool) -> List[Primitive]:
    wall = (Plane(0.0, 0.0, 1.0, 2.0, Material(1.0), texture=0) if textured_wall
            else Plane(0.0, 0.0, 1.0, 2.0, m["cyan"]))
    return [
        Plane(0.0, -1.0, 0.0, 1.0, m["checker"]),
        wall,
        Plane(0.0, 0.0, -1.0, 3.5, m["cyan"]),
        Plane(0.0, 1.0, 0.0, 4.0, m["orange"]),
        Plane(1.0, 0.0, 0.0, 4.0, m["green"]),
        Plane(-1.0, 0.0, 0.0, 4.0, m["pink"]),
    ]


def default_scene(textured_wall: bool = False) -> Scene:
    """MyApplication.cs:109-140.  ``textured_wall`` selects the Windows branch (:117-127),
    where the z = -2 wall is a default Material(1.0) plane with the wall texture in slot 0."""
    m = _materials()
    s = Scene()
    s.Add(Sphere(0.0, 0.0, 2.0, 0.5, m["mirror"]))
    s.Add(Sphere(0.5, 0.6, 2.2, 0.3, m["green"]))
    s.Add(Triangle((-2.0, 0.0, 1.0), (-1.5, 0.0, 2.0), (-2.0, -1.5, 2.0), m["cyan"]))
    for p in _box_planes(m, textured_wall):
        s.Add(p)
    s.Add(Light(1.0, -1.0, 1.5))
    s.Add(Light(-1.0, -1.0, 0.5))
    return s


# --------------------------------------------------------------------------------------
# Synthetic throughput scenes (this build's definition; the reference has no generator).
# --------------------------------------------------------------------------------------

class SplitMix64:
    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self) -> np.float32:
        """24 random bits scaled by 2^-24: exact in float32, in [0, 1)."""
        return f32((self.next() >> 40) * (1.0 / 16777216.0))


SEED_BASE = 0x5EEDC0DE


def _lerp(lo: float, hi: float, u: np.float32) -> np.float32:
    return f32(f32(lo) + f32(f32(f32(hi) - f32(lo)) * u))


def synthetic_scene(num_spheres: int, *, radius=(0.03, 0.12), seed: int = SEED_BASE + 3,
                    lights: str = "default") -> Scene:
    """The six box planes of the default scene, then ``num_spheres`` random spheres.

    Per sphere the generator draws, in this order, cx, cy, cz, r, material class, dr, dg, db
    (eight draws even when some are unused); a sphere that contains the camera origin is
    redrawn.  Classes: [0,.25) mirror, [.25,.5) specular-0.5 orange, [.5,1) diffuse with
    diffuseColor = .5 + .5*(dr,dg,db) and ambient = .07 * diffuseColor.
    ``lights``: "default" = the default scene's two lights; "grid16" = 16 lights on a 4x4
    grid at y = -1, x in {-3,-1,1,3}, z in {0.3,1.3,2.3,3.3}.
    """
    m = _materials()
    rng = SplitMix64(seed)
    s = Scene()
    for p in _box_planes(m, textured_wall=False):
        s.Add(p)
    made = 0
    while made < num_spheres:
        cx = _lerp(-3.5, 3.5, rng.uniform())
        cy = _lerp(-3.5, 0.7, rng.uniform())
        cz = _lerp(0.6, 3.2, rng.uniform())
        r = _lerp(radius[0], radius[1], rng.uniform())
        cls = rng.uniform()
        dr, dg, db = rng.uniform(), rng.uniform(), rng.uniform()
        c = _v(cx, cy, cz)
        if _dot(c, c) <= f32(r * r):
            continue
        if cls < f32(0.25):
            mat = m["mirror"]
        elif cls < f32(0.5):
            mat = m["orange"]
        else:
            col = [_lerp(0.5, 1.0, u) for u in (dr, dg, db)]
            amb = [f32(f32(0.07) * cc) for cc in col]
            mat = Material(1.0, diffuseColor=(*col, 1.0), ambientColor=(*amb, 1.0))
        s.Add(Sphere(cx, cy, cz, r, mat))
        made += 1
    if lights == "default":
        s.Add(Light(1.0, -1.0, 1.5))
        s.Add(Light(-1.0, -1.0, 0.5))
    elif lights == "grid16":
        for j in range(4):
            for i in range(4):
                s.Add(Light(-3.0 + 2.0 * i, -1.0, float(f32(0.3) + f32(j))))
    else:
        raise ValueError(lights)
    return s


def scene_1k() -> Scene:
    """1000 spheres + 6 planes, 2 lights: the 3840x2160 / 4-bounce throughput scene."""
    return synthetic_scene(1000, radius=(0.03, 0.12), seed=SEED_BASE + 3)


def scene_64k() -> Scene:
    """65536 small spheres + 6 planes, 16 lights: the divergence / occupancy stress scene."""
    return synthetic_scene(65536, radius=(0.005, 0.03), seed=SEED_BASE + 5, lights="grid16")
