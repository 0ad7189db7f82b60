"""Host-side mirror of the reference's renderer classes on top of the C ABI.

``Surface`` and ``Raytracer`` carry the names and meaning of the reference's classes
(/root/reference/surface.cs:7-19, MyApplication.cs:383-426): ``Raytracer(scene, camera,
screen).Render()`` fills ``screen.pixels`` with 0x00RRGGBB ints, row 0 at the top.  The
work is done by librt_hip.so through ``RenderContext``, a thin object wrapper over the
entry points of include/rt_hip.h; errors surface as ``RtError`` carrying the library's
message, the way the reference throws ``Exception(message)`` (template.cs:102-106).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import abi
from .scene import Camera, FlatScene, Scene


class RtError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{abi.STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


class RenderContext:
    """One rt_ctx.  Not thread-safe: one caller thread per context, like the reference."""

    def __init__(self, device_ids: Optional[Sequence[int]] = None):
        self._lib = abi.load_library()
        self._ctx = C.c_void_p()
        ids = None
        n = 0
        if device_ids:
            n = len(device_ids)
            ids = (C.c_int * n)(*device_ids)
        rc = self._lib.rt_create(C.byref(self._ctx), ids, n)
        if rc != abi.RT_OK:
            raise RtError(rc, self._lib.rt_last_error(None).decode())
        self._scene_keepalive = None

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != abi.RT_OK:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ABI calls --------------------------------------------------------------------
    def set_scene(self, flat: FlatScene) -> None:
        self._check(self._lib.rt_set_scene(self._ctx, C.byref(flat.as_struct())))

    def set_texture(self, slot: int, rgba8: np.ndarray) -> None:
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 4:
            raise ValueError("texture must be an HxWx4 uint8 array (R,G,B,A)")
        self._check(self._lib.rt_set_texture(self._ctx, slot, img.ctypes.data, img.shape[1],
                                             img.shape[0]))

    def pin_pixels(self, pixels: np.ndarray) -> None:
        """rt_pin_pixels: page-lock a C-contiguous int32 array that later render(out=...) calls
        write into (the caller keeps it alive until unpin_pixels() or close())."""
        if pixels.dtype != np.int32 or not pixels.flags.c_contiguous:
            raise ValueError("pixels must be a C-contiguous int32 array")
        self._check(self._lib.rt_pin_pixels(self._ctx, pixels.ctypes.data, pixels.size))
        self._pinned_keepalive = pixels

    def unpin_pixels(self) -> None:
        self._check(self._lib.rt_unpin_pixels(self._ctx))
        self._pinned_keepalive = None

    def alloc_pixels(self, height: int, width: int) -> np.ndarray:
        """rt_alloc_pixels: a height x width int32 framebuffer in page-locked memory of the library's own
        (a view, valid until unpin_pixels(), the next alloc_pixels() / pin_pixels(), or close());
        render(out=...) into it, or into a row range of it, is written by the DMA itself."""
        import ctypes as C
        p = C.c_void_p()
        self._check(self._lib.rt_alloc_pixels(self._ctx, int(height) * int(width), C.byref(p)))
        buf = (C.c_int32 * (int(height) * int(width))).from_address(p.value)
        self._pinned_keepalive = None
        return np.frombuffer(buf, dtype=np.int32).reshape(height, width)

    def set_profiling(self, enable) -> None:
        """rt_set_profiling: False / 0 off, True / 1 counters and kernel timing, 2 kernel timing only."""
        self._check(self._lib.rt_set_profiling(self._ctx, int(enable)))

    def render(self, camera: Camera, width: int, height: int,
               max_depth: int = abi.RT_REFERENCE_MAX_DEPTH,
               out: Optional[np.ndarray] = None) -> np.ndarray:
        """rt_render: returns (or fills) an int32 [height, width] array in host memory."""
        if out is None:
            out = np.empty((height, width), dtype=np.int32)
        if out.dtype != np.int32 or out.size != width * height or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous int32 array of width*height elements")
        self._check(self._lib.rt_render(self._ctx, C.byref(camera.as_struct()), width, height,
                                        max_depth, out.ctypes.data))
        return out

    def render_rows_device(self, camera: Camera, width: int, height: int, row_begin: int,
                           row_end: int, max_depth: int, dev_ptr: int, stream: int = 0) -> None:
        """rt_render_rows_device: dev_ptr is a device address of (row_end-row_begin)*width
        int32 (e.g. ``tensor.data_ptr()``), stream a hipStream_t handle or 0."""
        self._check(self._lib.rt_render_rows_device(
            self._ctx, C.byref(camera.as_struct()), width, height, row_begin, row_end, max_depth,
            C.c_void_p(dev_ptr), C.c_void_p(stream)))

    def render_rows_device_noalloc(self, camera: Camera, width: int, height: int, row_begin: int,
                                   row_end: int, max_depth: int, dev_ptr: int, stream: int = 0) -> None:
        """rt_render_rows_device_noalloc: render_rows_device with the ray queues in the library's
        fixed scratch array; raises RtError (RT_ERR_LIMIT) when the rows do not fit it."""
        self._check(self._lib.rt_render_rows_device_noalloc(
            self._ctx, C.byref(camera.as_struct()), width, height, row_begin, row_end, max_depth,
            C.c_void_p(dev_ptr), C.c_void_p(stream)))

    def render_bands_device(self, camera: Camera, width: int, height: int, band_rows: int,
                            part: int, parts: int, max_depth: int, dev_ptr: int, stream: int = 0) -> None:
        """rt_render_bands_device: every parts-th band of band_rows rows, starting with band
        `part`, written one after the other to dev_ptr."""
        self._check(self._lib.rt_render_bands_device(
            self._ctx, C.byref(camera.as_struct()), width, height, band_rows, part, parts, max_depth,
            C.c_void_p(dev_ptr), C.c_void_p(stream)))

    def set_device_share(self, contexts: int) -> None:
        """rt_set_device_share: this many contexts render on the device at the same time."""
        self._check(self._lib.rt_set_device_share(self._ctx, int(contexts)))

    def reload_tuning(self) -> None:
        """rt_reload_tuning: read the RT_HIP_* tuning aids and test knobs from the environment again
        (the library reads them when a context is created and when it gets a scene, not per frame)."""
        self._check(self._lib.rt_reload_tuning(self._ctx))

    def frame_status(self) -> None:
        """rt_frame_status: waits for the frames enqueued with render_*_device so far and raises
        RtError (RT_ERR_STALLED) if the device gave up on one of them."""
        self._check(self._lib.rt_frame_status(self._ctx))

    def debug_rays(self, camera: Camera, width: int, height: int, row: int = 320, x_step: int = 10) -> np.ndarray:
        """rt_debug_rays as a structured array (direction, prim, hit, t, normal, reflected): the
        rays Raytracer.Debug draws (MyApplication.cs:459-478: row 320, every 10th pixel)."""
        n = (width + x_step - 1) // x_step
        buf = (abi.rt_debug_ray * max(n, 1))()
        count = C.c_int(0)
        self._check(self._lib.rt_debug_rays(self._ctx, C.byref(camera.as_struct()), width, height, row, x_step,
                                            n, buf, C.byref(count)))
        dt = np.dtype([("direction", np.float32, 3), ("prim", np.int32), ("hit", np.float32, 3), ("t", np.float32),
                       ("normal", np.float32, 3), ("reflected", np.float32, 3)])
        return np.frombuffer(buf, dtype=dt, count=count.value).copy()

    def fxaa_device(self, dev_in: int, dev_out: int, width: int, height: int, stream: int = 0) -> None:
        """rt_fxaa_device: the present path's FXAA filter from one device frame to another."""
        self._check(self._lib.rt_fxaa_device(self._ctx, C.c_void_p(dev_in), C.c_void_p(dev_out), width, height,
                                             C.c_void_p(stream)))

    def stats(self) -> dict:
        st = abi.rt_stats()
        self._check(self._lib.rt_get_stats(self._ctx, C.byref(st)))
        return {name: getattr(st, name) for name, _ in st._fields_ if name != "reserved"}


class Surface:
    """surface.cs:7-19: a width x height int framebuffer, allocated once."""

    def __init__(self, width: int, height: int):
        self.width, self.height = int(width), int(height)
        self.pixels = np.zeros(self.width * self.height, dtype=np.int32)

    def Clear(self, c: int) -> None:  # surface.cs:42-45
        self.pixels[:] = c

    def image(self) -> np.ndarray:
        return self.pixels.reshape(self.height, self.width)


class Raytracer:
    """MyApplication.cs:383-426 with Render() running on the GPU."""

    def __init__(self, scene: Scene, camera: Camera, screen: Surface,
                 device_ids: Optional[Sequence[int]] = None, textures: Sequence[np.ndarray] = (),
                 max_depth: int = abi.RT_REFERENCE_MAX_DEPTH):
        self.scene, self.camera, self.screen = scene, camera, screen
        self.max_depth = max_depth
        self.context = RenderContext(device_ids)
        try:   # Surface.pixels is never reallocated (surface.cs:14-19); an optimisation only
            self.context.pin_pixels(self.screen.pixels)
        except RtError:
            pass
        for slot, tex in enumerate(textures):
            self.context.set_texture(slot, tex)
        self.UpdateScene()

    def UpdateScene(self) -> None:
        """Re-flatten the scene after the host edited it (the reference reads its object
        graph live; the library works on a copy)."""
        self.context.set_scene(self.scene.flatten())

    def Render(self) -> None:
        self.context.render(self.camera, self.screen.width, self.screen.height, self.max_depth,
                            out=self.screen.pixels)
