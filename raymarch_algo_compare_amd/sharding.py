"""Row sharding of one frame across the GPUs of a node (one process per GPU).

The frame is embarrassingly parallel over rays, so there is no exchange during the render; the
only communication is the optional final gather of the three output maps (RCCL all-gather over
xGMI through torch.distributed, worthwhile only for >= 4K frames -- SURVEY.md section 5i / 8e).

Rows are dealt band-cyclically (bands of 4 rows: rank r renders bands r, r+N, r+2N, ...) so every
rank sees the same mix of sky rows and object rows; contiguous 1/N blocks would leave the ranks
holding the object's rows as stragglers.  Bands are 4 rows high so the reference's 8x4 divergence
blocks (core/types.py:125-133) never straddle two ranks.  When height is not a multiple of
4 * world_size the plan falls back to contiguous, 4-aligned row blocks.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

BAND_ROWS = 4


@dataclass(frozen=True)
class ShardPlan:
    height: int
    world_size: int
    rank: int
    cyclic: bool
    row0: int          # contiguous: first image row; cyclic: 0
    rows: int          # local row count
    band_rows: int = 0
    band_stride: int = 0
    band_offset: int = 0

    def desc_kwargs(self) -> dict:
        """Keyword arguments for _native.make_desc describing this rank's slice."""
        if self.cyclic:
            return dict(row0=0, rows=self.rows, band_rows=self.band_rows, band_stride=self.band_stride,
                        band_offset=self.band_offset)
        return dict(row0=self.row0, rows=self.rows)

    def image_rows(self) -> np.ndarray:
        """Image row of every local row (length `rows`)."""
        y = np.arange(self.rows)
        if self.cyclic:
            return ((y // self.band_rows) * self.band_stride + self.band_offset) * self.band_rows + y % self.band_rows
        return self.row0 + y


def plan_rows(height: int, world_size: int, rank: int) -> ShardPlan:
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    if world_size == 1:
        return ShardPlan(height, 1, 0, False, 0, height)
    if height % (BAND_ROWS * world_size) == 0:
        return ShardPlan(height, world_size, rank, True, 0, height // world_size, BAND_ROWS, world_size, rank)
    # contiguous blocks with 4-aligned boundaries; the last rank takes the remainder
    nblk = (height + BAND_ROWS - 1) // BAND_ROWS
    per = (nblk + world_size - 1) // world_size
    r0 = min(rank * per * BAND_ROWS, height)
    r1 = min((rank + 1) * per * BAND_ROWS, height)
    return ShardPlan(height, world_size, rank, False, r0, r1 - r0)


def assemble(parts, plans) -> np.ndarray:
    """Host-side reassembly of per-rank (rows, W) arrays into the (H, W) image."""
    H = plans[0].height
    W = parts[0].shape[1]
    out = np.empty((H, W), dtype=parts[0].dtype)
    for a, p in zip(parts, plans):
        out[p.image_rows()] = a
    return out


def all_gather_frame(local, plan: ShardPlan, group=None):
    """RCCL (backend 'nccl') / gloo all-gather of one output map and un-permutation into image
    order.  `local` is a torch tensor (rows, W) on this rank's device; returns (H, W) on every rank.
    Cyclic plans have equal shards; uneven contiguous plans are padded for the collective."""
    import torch
    import torch.distributed as dist
    N = plan.world_size
    if N == 1:
        return local
    if not plan.cyclic:
        sizes = [plan_rows(plan.height, N, r).rows for r in range(N)]
        if len(set(sizes)) != 1:
            # uneven contiguous blocks: pad every shard to the largest, gather, trim
            mx, W = max(sizes), local.shape[1]
            padded = torch.zeros((mx, W), dtype=local.dtype, device=local.device)
            padded[: local.shape[0]] = local
            buf = torch.empty((N * mx, W), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(buf, padded, group=group)
            return torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(N)], dim=0)
    # output laid out as the concatenation of the rank shards along dim 0 (the form every backend accepts)
    gathered = torch.empty((N * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    if not plan.cyclic:
        return gathered.reshape(plan.height, local.shape[1])
    nb = plan.rows // plan.band_rows
    W = local.shape[1]
    # gathered[r, b*band + i] is image row (b*N + r)*band + i
    return gathered.reshape(N, nb, plan.band_rows, W).permute(1, 0, 2, 3).reshape(plan.height, W)
