"""FrameGatherer with DEVICE blocks on a gloo group (ranks rehearsed on fewer GPUs than ranks): the blocks must
travel through host tensors on the gatherer's own stream — gloo handed a device tensor copies it on a
high-priority side stream, which stalls a GPU that two processes share.  All ranks' gatherers live in this one
process; torch.distributed's gather is replaced by a stand-in that moves the tensors it is given."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("interleave,height,world", [(True, 150, 3), (True, 64, 2), (False, 101, 3), (False, 100, 2)])
def test_gloo_gather_of_device_blocks_goes_through_host_tensors(monkeypatch, interleave, height, world):
    import torch
    from raytracer_amd import distributed as D

    width = 40
    device = torch.device("cuda", 0)
    frame = torch.arange(height * width, dtype=torch.int32).reshape(height, width)
    gatherers = [D.FrameGatherer(width, height, r, world, device, interleave=interleave) for r in range(world)]
    for r, g in enumerate(gatherers):
        rows = D.band_rows_of(height, r, world) if g.interleave else list(range(*D.row_block(height, r, world)))
        assert g.rows == len(rows)
        g.block.copy_(frame[rows].to(device))

    sent, seen_devices = {}, []

    def fake_gather(tensor, gather_list=None, dst=0, group=None):
        seen_devices.append(tensor.device.type)
        rank = len(sent) + 1 if gather_list is None else 0      # ranks 1 .. world - 1 call first, rank 0 last
        sent[rank] = tensor.clone()
        if gather_list is not None:
            assert len(gather_list) == world
            for r, out in enumerate(gather_list):
                seen_devices.append(out.device.type)
                out.copy_(sent[r])

    monkeypatch.setattr(D.dist, "get_backend", lambda group=None: "gloo")
    monkeypatch.setattr(D.dist, "gather", fake_gather)
    for r in range(1, world):
        assert gatherers[r].gather() is None
    out = gatherers[0].gather()
    gatherers[0].stream.synchronize()
    assert set(seen_devices) == {"cpu"}, "gloo must only ever see host tensors"
    assert np.array_equal(out.cpu().numpy(), frame.numpy())
