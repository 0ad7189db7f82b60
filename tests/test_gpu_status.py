"""The device-entry path reports a frame the device gave up on (rt_frame_status)."""
import os

import numpy as np
import pytest

from raytracer_amd import Camera, RenderContext, RtError, scene_1k

pytestmark = pytest.mark.gpu


def test_stalled_schedule_surfaces_through_every_entry(oracle, monkeypatch):
    """A test knob loses the completion count of one reflecting tile of level 0, so level 1 never
    opens: the workgroups give up after the stall limit, the frame is incomplete, and both the
    asynchronous entry (through rt_frame_status) and rt_render say so.  The next frame is whole."""
    import torch
    fs, cam, (w, h), depth = scene_1k().flatten(), Camera(), (320, 200), 3
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth)
    stream = torch.cuda.Stream()
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    with RenderContext() as c:
        c.set_scene(fs)
        c.render_rows_device(cam, w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
        c.frame_status()                       # a healthy frame: nothing to report
        assert np.array_equal(buf.cpu().numpy(), ref)
        monkeypatch.setenv("RT_HIP_PIPELINE", "fused")   # the form with a device-side schedule
        monkeypatch.setenv("RT_HIP_STALL_MS", "150")
        monkeypatch.setenv("RT_HIP_DEBUG_WITHHOLD_LEVEL", "0")
        c.reload_tuning()
        c.render_rows_device(cam, w, h, 0, h, depth, buf.data_ptr(), stream.cuda_stream)
        with pytest.raises(RtError) as e:
            c.frame_status()
        assert e.value.status == -8 and "level 1" in e.value.message
        c.frame_status()                       # the report was consumed
        with pytest.raises(RtError) as e:
            c.render(cam, w, h, depth)
        assert e.value.status == -8
        monkeypatch.delenv("RT_HIP_DEBUG_WITHHOLD_LEVEL")
        c.reload_tuning()
        c.render_bands_device(cam, w, h, 16, 0, 1, depth, buf.data_ptr(), stream.cuda_stream)
        c.frame_status()
        assert np.array_equal(buf.cpu().numpy(), ref)
        assert np.array_equal(c.render(cam, w, h, depth), ref)


def test_frames_on_many_caller_streams_are_the_same_frame(oracle):
    """A host may hand over any stream, and a new one every frame.  The runtime keeps a process's streams on a
    few hardware queues, so sooner or later a caller's stream shares one with the context's second stream; the
    library checks the pair the first time it sees a stream and takes another second stream if need be (a
    hazard for the frame's time, never for its pixels).  Eight streams, two scenes: every frame is the frame."""
    import torch
    fs, cam, (w, h), depth = scene_1k().flatten(), Camera(), (320, 200), 3
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth)
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(8)]
    with RenderContext() as c:
        c.set_scene(fs)
        for rounds in range(2):
            for s in streams:
                buf.zero_()
                torch.cuda.synchronize()
                c.render_rows_device(cam, w, h, 0, h, depth, buf.data_ptr(), s.cuda_stream)
                c.frame_status()
                assert np.array_equal(buf.cpu().numpy(), ref)
