"""Multi-GPU image tiling for the ray-trace path: block-cyclic ownership of scanline blocks (or of tiles) + one gather.

Pixels are independent (sendRaysAndIntersectPointsColors, simple_raytracer.cpp:511-517) and the scene
is read-only, so the frame shards with no data-path exchange except the final assembly of the
framebuffer on rank 0.  Scanline block b (BLOCK_ROWS rows) belongs to rank b mod world: sky rows and
object rows are dealt round-robin, which keeps the ranks' work balanced where contiguous bands
would not be.  With block_cols > 0 the blocks are cut into tiles of BLOCK_ROWS x block_cols pixels and tile (bx, by) belongs
to rank (bx + by) mod world (include/srt.h srt_params.block_cols): expensive pixels cluster in both directions (a tree crown, a
bunny), and whole-width rows spread a cluster over 8 ranks only coarsely.  One process per GPU; the collective is torch.distributed (backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" in the CPU tests).  torch is plumbing here: device memory and the collective.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import abi


class FrameGather:
    """Owns this rank's padded output tiles and assembles whole frames on rank `dst`.

    Every rank's tile is padded to the same number of rows so that the exchange is ONE equal-size gather per
    step (7 peers -> rank 0, each over its own xGMI link).  `slots` > 1 gives that many independent tile buffers so
    that the gather of one step overlaps the rendering of the next (RCCL runs the collective on its own stream):

        g = FrameGather(..., frames=B, slots=2)
        for step in range(K):
            k = step % 2
            g.finish(k)                  # the gather that last used slot k is done; rank dst de-interleaves it
            render into g.tiles[k] ...
            g.start(k)                   # asynchronous gather of slot k
        g.finish_all()
    """

    def __init__(self, width, height, block_rows, rank, world, device, channels=3, dtype=torch.uint8, dst=0, frames=1,
                 stage_through_host=False, slots=1, frame_groups=1, block_cols=0):
        """frame_groups = F: the `frames` of a step are dealt to F groups of R = world / F ranks (frame f belongs to group
        f % F); inside a group every frame is split into scanline blocks over the group's R ranks.  Rank r is member
        r % R of group r // R.  F = 1 is the pure scanline split, F = world whole frames per rank."""
        assert world % frame_groups == 0 and frames % frame_groups == 0, "frame groups must divide the ranks and the frames"
        self.W, self.H, self.block_rows, self.rank, self.world, self.dst = width, height, block_rows, rank, world, dst
        self.groups, self.per_group = frame_groups, world // frame_groups
        self.group, self.member = rank // self.per_group, rank % self.per_group
        self.block_cols = block_cols if self.per_group > 1 else 0
        # per rank: image pixel (flat index) of every local output pixel, -1 = padding of a tile deal
        self.pix_of = [abi.owned_pixels(width, height, block_rows, r % self.per_group, self.per_group, self.block_cols) for r in range(world)]
        self.rows, self.cols = self.pix_of[rank].shape
        self.max_rows = max(p.shape[0] for p in self.pix_of)
        self.max_cols = max(p.shape[1] for p in self.pix_of)
        self.frames_total = frames
        frames = frames // frame_groups              # frames this rank renders per step
        self.frames = frames
        self.stage = stage_through_host            # gloo rehearsal on a GPU box: collectives on host copies
        # [frames, rows, W, C]: a step's frames travel in ONE collective (few, large messages suit the
        # point-to-point xGMI links: 7 peers -> rank 0, each over its own link)
        self.tiles = [torch.zeros((frames, self.max_rows, self.max_cols, channels), dtype=dtype, device=device) for _ in range(slots)]
        self.tile = self.tiles[0]
        self.work = [None] * slots
        self._host = [None] * slots
        # rank dst puts the gathered tiles together on a stream of its own: 8 x 28 MB of copies per step at N = 8 (a fifth of the
        # step) that would otherwise sit between two of its own renders -- rank dst is the slowest rank then, and the step is the MAX
        self.asm_stream = torch.cuda.Stream(device=device) if (torch.device(device).type == "cuda" and world > 1 and rank == dst) else None
        self.asm_done = [None] * slots
        if rank == dst:
            self.recv = [[torch.empty_like(self.tile) for _ in range(world)] for _ in range(slots)] if world > 1 else None
            self.frame = torch.empty((self.frames_total, height, width, channels), dtype=dtype, device=device)
            # sender r: positions of its real pixels inside the padded [max_rows, max_cols] tile, and where they go in the frame.
            # Whole-width blocks move as whole rows (one contiguous W x C run each); tiles dealt in two dimensions pixel by pixel.
            self.src, self.index = [], []
            for pm in self.pix_of:
                if not self.block_cols:
                    self.src.append(None)
                    self.index.append(torch.as_tensor(pm[:, 0] // width, dtype=torch.long, device=device))      # image row of each local row
                    continue
                pos = (np.arange(pm.shape[0], dtype=np.int64)[:, None] * self.max_cols + np.arange(pm.shape[1], dtype=np.int64)[None, :])[pm >= 0]
                self.src.append(torch.as_tensor(pos, dtype=torch.long, device=device))
                self.index.append(torch.as_tensor(pm[pm >= 0], dtype=torch.long, device=device))
        else:
            self.recv, self.frame, self.index, self.src = None, None, None, None

    def start(self, k=0):
        """Begin the gather of slot k (asynchronous on the collective's stream)."""
        if self.world == 1:
            return
        if self.asm_done[k] is not None:                     # the previous round's assembly still reads recv[k]
            torch.cuda.current_stream().wait_event(self.asm_done[k])
        if self.stage:
            host = self.tiles[k].cpu()
            recv = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            self.work[k] = dist.gather(host, recv, dst=self.dst, async_op=True)
            self._host[k] = (host, recv)
        else:
            self.work[k] = dist.gather(self.tiles[k], self.recv[k] if self.rank == self.dst else None, dst=self.dst, async_op=True)

    def finish(self, k=0):
        """Wait (stream-wise) for slot k's gather and de-interleave it into `frame` on rank dst.  Returns `frame`
        [frames, H, W, C] on dst (None elsewhere, or if slot k has no gather in flight)."""
        if self.world == 1:
            return self.tiles[k][:, : self.rows, : self.cols]
        if self.work[k] is None:
            return None
        work = self.work[k]
        work.wait()                          # the calling stream may render into tiles[k] again
        self.work[k] = None
        if self.rank != self.dst:
            return None
        if self.asm_stream is not None:      # assembled on its own stream: synchronise (finish_all + torch.cuda.synchronize) before reading `frame`
            self.asm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.asm_stream):
                work.wait()
                self._assemble(k)
                self.asm_done[k] = torch.cuda.Event()
                self.asm_done[k].record(self.asm_stream)
        else:
            self._assemble(k)
        return self.frame

    def _assemble(self, k):
        if self.stage:
            _, recv = self._host[k]
            for r in range(self.world):
                self.recv[k][r].copy_(recv[r], non_blocking=True)
        C = self.frame.shape[-1]
        for r in range(self.world):
            if self.index[r].numel():       # sender r holds its pixels of the frames of its group: frames group, group + F, ...
                mine = self.frame[r // self.per_group :: self.groups]
                if self.src[r] is None:     # whole rows
                    mine.index_copy_(1, self.index[r], self.recv[k][r][:, : self.index[r].numel()])
                else:
                    flat = self.recv[k][r].reshape(mine.shape[0], -1, C).index_select(1, self.src[r])
                    mine.view(mine.shape[0], -1, C).index_copy_(1, self.index[r], flat)

    def wait_collective(self, k=0):
        """Wait for slot k's collective only (the assembly stays to be done by finish): for per-phase clocks."""
        if self.world > 1 and self.work[k] is not None:
            self.work[k].wait()

    def wait_assembly(self):
        """Block the host until rank dst's assembly stream has drained (no-op elsewhere)."""
        if self.asm_stream is not None:
            self.asm_stream.synchronize()

    def finish_all(self):
        out = None
        for k in range(len(self.tiles)):
            f = self.finish(k)
            out = f if f is not None else out
        return out

    def gather(self, k=0):
        """Synchronous form: gather slot k and return the assembled frames on rank dst (None elsewhere)."""
        self.start(k)
        return self.finish(k)


def split_params(width, height, lights, rank, world, block_rows, block_cols=0, **kw):
    """srt_params of rank `rank` for a frame tiled over `world` ranks (block_cols > 0: tiles dealt in two dimensions)."""
    if world == 1:
        return abi.make_params(width, height, lights, **kw)
    return abi.make_params(width, height, lights, block_rows=block_rows, block_first=rank, block_stride=world, block_cols=block_cols, **kw)
