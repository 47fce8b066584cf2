"""Multi-GPU image tiling for the ray-trace path: block-cyclic scanline ownership + one gather.

Pixels are independent (sendRaysAndIntersectPointsColors, simple_raytracer.cpp:511-517) and the scene
is read-only, so the frame shards with no data-path exchange except the final assembly of the
framebuffer on rank 0.  Scanline block b (BLOCK_ROWS rows) belongs to rank b mod world: sky rows and
object rows are dealt round-robin, which keeps the ranks' work balanced where contiguous bands
would not be.  One process per GPU; the collective is torch.distributed (backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" in the CPU tests).  torch is plumbing here: device memory and the collective.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import abi


class FrameGather:
    """Owns this rank's padded output tile and assembles the whole frame on rank `dst`.

    Every rank's tile is padded to the same number of rows so that the exchange is ONE equal-size
    gather per frame (7 peers -> rank 0, each over its own xGMI link)."""

    def __init__(self, width, height, block_rows, rank, world, device, channels=3, dtype=torch.uint8, dst=0, frames=1, stage_through_host=False):
        self.W, self.H, self.block_rows, self.rank, self.world, self.dst = width, height, block_rows, rank, world, dst
        self.rows_of = [abi.rows_owned(height, block_rows, r, world) for r in range(world)]
        self.rows = len(self.rows_of[rank])
        self.max_rows = max(len(r) for r in self.rows_of)
        self.frames = frames
        self.stage = stage_through_host            # gloo rehearsal on a GPU box: collectives on host copies
        # [frames, rows, W, C]: a step's frames travel in ONE collective (few, large messages suit the
        # point-to-point xGMI links: 7 peers -> rank 0, each over its own link)
        self.tile = torch.zeros((frames, self.max_rows, width, channels), dtype=dtype, device=device)
        if rank == dst:
            self.recv = [torch.empty_like(self.tile) for _ in range(world)] if world > 1 else None
            self.frame = torch.empty((frames, height, width, channels), dtype=dtype, device=device)
            self.index = [torch.as_tensor(np.asarray(r), dtype=torch.long, device=device) for r in self.rows_of]
        else:
            self.recv, self.frame, self.index = None, None, None

    def gather(self):
        """Gather every rank's `tile` to rank dst and de-interleave into `frame` [frames, H, W, C] (returned on
        dst, else None)."""
        if self.world == 1:
            return self.tile[:, : self.rows]
        if self.stage:
            host = self.tile.cpu()
            recv = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(host, recv, dst=self.dst)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.recv[r].copy_(recv[r])
        else:
            dist.gather(self.tile, self.recv, dst=self.dst)
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            n = len(self.rows_of[r])
            if n:
                self.frame.index_copy_(1, self.index[r], self.recv[r][:, :n])
        return self.frame


def split_params(width, height, lights, rank, world, block_rows, **kw):
    """srt_params of rank `rank` for a frame tiled over `world` ranks."""
    if world == 1:
        return abi.make_params(width, height, lights, **kw)
    return abi.make_params(width, height, lights, block_rows=block_rows, block_first=rank, block_stride=world, **kw)
