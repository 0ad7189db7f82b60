"""The one-process-per-GPU row partition and gather, rehearsed with gloo on CPU.

Two (and three) CPU processes each fill their block with the oracle's rows and gather
them on rank 0; the result must equal the single-process frame.  The same FrameGatherer
drives RCCL on the GPUs (bench.py).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, w: int, h: int, interleave: bool, outdir: str):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from raytracer_amd import Camera, default_scene
    from raytracer_amd.distributed import FrameGatherer, band_rows_of, row_block
    fs, cam = default_scene().flatten(), Camera()
    g = FrameGatherer(w, h, rank, world, torch.device("cpu"), interleave=interleave, band_rows=4)
    img, _ = O.render(fs, cam, w, h, nthreads=1)
    mine = band_rows_of(h, rank, world, 4) if interleave else list(range(*row_block(h, rank, world)))
    assert g.rows == len(mine)
    for _ in range(2):  # two frames in a row reuse the buffers
        g.block.copy_(torch.from_numpy(img[mine]))
        frame = g.gather()
    if rank == 0:
        np.save(os.path.join(outdir, "frame.npy"), frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank: int, world: int, port: int, w: int, h: int, outdir: str):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from raytracer_amd import Camera, default_scene
    from raytracer_amd.distributed import FramePipeline, band_rows_of
    fs = default_scene().flatten()
    pipe = FramePipeline(w, h, rank, world, torch.device("cpu"), band_rows=4, depth=2)
    mine = band_rows_of(h, rank, world, 4)
    frames = []
    for i in range(5):   # five different frames through two slots
        img, _ = O.render(fs, Camera(position=(0.1 * i, 0.0, 0.0)), w, h, nthreads=1)
        slot = pipe.begin()
        slot.block.copy_(torch.from_numpy(img[mine]))
        frame = pipe.finish(slot)
        if rank == 0:
            frames.append(frame.numpy().copy())
        else:
            assert frame is None
    pipe.synchronize()
    if rank == 0:
        np.save(os.path.join(outdir, "frames.npy"), np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


def test_pipeline_of_two_frames_in_flight(tmp_path, oracle):
    from raytracer_amd import Camera, default_scene
    w, h, world = 48, 30, 2
    mp.spawn(_pipeline_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "frames.npy")
    fs = default_scene().flatten()
    for i in range(5):
        ref, _ = oracle.render(fs, Camera(position=(0.1 * i, 0.0, 0.0)), w, h)
        assert np.array_equal(got[i], ref), i


@pytest.mark.parametrize("world,h,interleave", [(2, 48, False), (2, 37, False), (3, 50, False),
                                                (2, 48, True), (2, 37, True), (3, 50, True), (3, 7, True)])
def test_gather_reassembles_the_frame(tmp_path, oracle, world, h, interleave):
    from raytracer_amd import Camera, default_scene
    w = 64
    mp.spawn(_worker, args=(world, _free_port(), w, h, interleave, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "frame.npy")
    ref, _ = oracle.render(default_scene().flatten(), Camera(), w, h)
    assert np.array_equal(got, ref)


def test_row_blocks_tile_the_frame():
    from raytracer_amd.distributed import row_block
    for h in (1, 7, 1080, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            blocks = [row_block(h, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        row_block(10, 2, 2)


def test_bands_tile_the_frame():
    from raytracer_amd.distributed import band_rows_of
    for h in (1, 7, 37, 1080, 2160):
        for world in (1, 2, 3, 8):
            for band in (1, 4, 16):
                rows = [band_rows_of(h, r, world, band) for r in range(world)]
                assert sorted(y for m in rows for y in m) == list(range(h))
                sizes = [len(m) for m in rows]
                assert max(sizes) - min(sizes) <= band
