"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- block-cyclic scanline ownership + one
gather to rank 0 + de-interleave (simple_raytracer_amd/tiling.py).  The tiles are produced by the CPU
oracle here (no GPU in this container); on the GPU box the same FrameGather carries the HIP output."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, block_rows, W, H, L, q, block_cols=0):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_util as gu
        from oracle import pyoracle as po
        from simple_raytracer_amd import abi, tiling
        g = gu.GoldenScene("cubes4_a0")
        lights = abi.light_staircase(g.light, L)
        p = tiling.split_params(W, H, lights, rank, world, block_rows, block_cols)
        o = po.render(g.flat, p, n_threads=1)
        probe = tiling.FrameGather(W, H, block_rows, rank, world, torch.device("cpu"), block_cols=block_cols)
        assert (probe.rows, probe.cols) == o["rgb8"].shape[:2]
        fg = tiling.FrameGather(W, H, block_rows, rank, world, torch.device("cpu"), frames=2, block_cols=block_cols)
        fg.tile[0, : fg.rows, : fg.cols].copy_(torch.from_numpy(o["rgb8"]))
        fg.tile[1, : fg.rows, : fg.cols].copy_(torch.from_numpy(255 - o["rgb8"]))
        frame = fg.gather()
        # two slots, gathers in flight while the next slot is filled (the overlap scheme of bench.py)
        hg = tiling.FrameGather(W, H, block_rows, rank, world, torch.device("cpu"), channels=1, dtype=torch.int32, slots=2, block_cols=block_cols)
        hg.tiles[0][0, : hg.rows, : hg.cols, 0].copy_(torch.from_numpy(o["hit_id"]))
        hg.start(0)
        hg.tiles[1][0, : hg.rows, : hg.cols, 0].copy_(torch.from_numpy(o["hit_id"] + 7))
        hg.start(1)
        hits = hg.finish(0)
        if rank == 0:
            hits = hits.clone()
        shifted = hg.finish(1)
        if rank == 0:
            assert torch.equal(shifted, hits + 7)
        assert hg.finish_all() is None
        if rank == 0:
            assert torch.equal(frame[1], 255 - frame[0])
            q.put((frame[0].numpy().copy(), hits[0].numpy()[..., 0].copy()))
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,block_rows,block_cols", [(2, 16, 0), (2, 7, 0), (3, 5, 0), (2, 16, 32), (3, 8, 16), (4, 8, 24), (8, 8, 8)])
def test_gather_reassembles_reference_frame(world, block_rows, block_cols):
    """Scanline blocks (block_cols = 0) and tiles dealt in two dimensions (the column term of srt_params): tiles rendered per rank,
    one gather, de-interleave on rank 0 = the reference's image.  (8 ranks: the world size of the target node.)"""
    import golden_util as gu
    W, H, L = 128, 96, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, block_rows, W, H, L, q, block_cols)) for r in range(world)]
    for p in procs:
        p.start()
    frame, hits = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = gu.GoldenScene("cubes4_a0")
    assert np.array_equal(frame, g.out(W, H, L, "rgb8")), "tiled + gathered frame differs from the reference image"
    assert np.array_equal(hits, g.out(W, H, L, "hit_id"))


def _group_worker(rank, world, port, groups, block_rows, W, H, frames, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_util as gu
        from oracle import pyoracle as po
        from simple_raytracer_amd import abi, tiling
        g = gu.GoldenScene("cubes4_a0")
        fg = tiling.FrameGather(W, H, block_rows, rank, world, torch.device("cpu"), frames=frames, frame_groups=groups)
        assert fg.frames == frames // groups and fg.per_group == world // groups
        # frame f of the step is the scene lit by f + 1 light samples; this rank renders frames group, group + F, ...
        # and of each only the scanline blocks of its place inside the group
        for k in range(fg.frames):
            f = k * groups + fg.group
            p = tiling.split_params(W, H, abi.light_staircase(g.light, f + 1), fg.member, fg.per_group, block_rows)
            o = po.render(g.flat, p, n_threads=1)
            assert o["rgb8"].shape[0] == fg.rows
            fg.tile[k, : fg.rows].copy_(torch.from_numpy(o["rgb8"]))
        out = fg.gather()
        if rank == 0:
            q.put(out.numpy().copy())
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,groups,block_rows", [(2, 2, 16), (4, 2, 8), (3, 1, 5)])
def test_frame_groups_reassemble_every_frame(world, groups, block_rows):
    """bench.py deals the frames of a step to groups of ranks and splits each frame by scanline blocks inside a group:
    rank 0 must end up with every frame, each identical to the whole-frame render."""
    import golden_util as gu
    from oracle import pyoracle as po
    from simple_raytracer_amd import abi
    W, H, frames = 96, 64, 4 if groups > 1 else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, groups, block_rows, W, H, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    g = gu.GoldenScene("cubes4_a0")
    assert got.shape == (frames, H, W, 3)
    for f in range(frames):
        want = po.render(g.flat, abi.make_params(W, H, abi.light_staircase(g.light, f + 1)), n_threads=2)["rgb8"]
        assert np.array_equal(got[f], want), f"frame {f}"
