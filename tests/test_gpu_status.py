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


def test_host_side_deadline_reports_where_the_device_was(oracle, monkeypatch):
    """No wait on the device is unbounded: with a deadline shorter than the frame (the frame is made long by the knob
    that stalls the single-launch form's schedule for 400 ms) rt_render comes back with RT_ERR_STALLED and a
    description of what the device had reached; the context then refuses further frames and is destroyed without
    waiting for the device; a new context renders the frame."""
    import time
    fs, cam, (w, h), depth = scene_1k().flatten(), Camera(), (320, 200), 3
    ref, _ = oracle.render(fs, cam, w, h, max_depth=depth)
    monkeypatch.setenv("RT_HIP_PIPELINE", "fused")
    monkeypatch.setenv("RT_HIP_STALL_MS", "400")
    monkeypatch.setenv("RT_HIP_DEBUG_WITHHOLD_LEVEL", "0")
    monkeypatch.setenv("RT_HIP_FRAME_TIMEOUT_MS", "50")
    c = RenderContext()
    c.set_scene(fs)
    t = time.perf_counter()
    with pytest.raises(RtError) as e:
        c.render(cam, w, h, depth)
    waited = time.perf_counter() - t
    assert e.value.status == -8 and "time limit" in e.value.message and "second stream" in e.value.message
    assert 0.04 < waited < 0.35
    with pytest.raises(RtError) as e:
        c.render(cam, w, h, depth)
    assert e.value.status == -8 and "destroy the context" in e.value.message
    c.close()
    time.sleep(0.6)                        # (the abandoned launch gives up by itself after its own limit)
    for k in ("RT_HIP_PIPELINE", "RT_HIP_STALL_MS", "RT_HIP_DEBUG_WITHHOLD_LEVEL", "RT_HIP_FRAME_TIMEOUT_MS"):
        monkeypatch.delenv(k)
    with RenderContext() as c2:
        c2.set_scene(fs)
        assert np.array_equal(c2.render(cam, w, h, depth), ref)
