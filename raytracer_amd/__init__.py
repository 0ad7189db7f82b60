"""raytracer_amd — MI355X (gfx950) render loop behind the reference host's interface.

The product is librt_hip.so (raytracer_amd/csrc, C ABI in include/rt_hip.h); this package
is the Python host layer above it: the scene classes and flatten (scene.py), the renderer
objects (raytracer.py), the one-process-per-GPU row partition (distributed.py) and the
in-tree build (build.py).
"""
from .abi import RT_REFERENCE_MAX_DEPTH, load_library  # noqa: F401
from .scene import (Camera, FlatScene, Light, Material, Plane, Scene, Sphere, Triangle,  # noqa: F401
                    default_scene, scene_1k, scene_64k, synthetic_scene)
from .raytracer import Raytracer, RenderContext, RtError, Surface  # noqa: F401
