"""Multi-GPU plumbing for independent blocks: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Blocks are independent LZS streams (each lzs_compress() call starts with empty history and
ends with its own end marker: reference lzs-compression.c:291-299, 449-466), so the compute
path needs no exchange at all: rank r compresses blocks [r*N/G, (r+1)*N/G).  Collectives
appear only where data really moves between GPUs:

* ``scatter_blocks``  root -> all: input shards (BASELINE.json configs[4] "input scatter");
* ``gather_streams``  all -> root: every rank's dense compressed bytes, variable length
  ("compressed-output gather"), preceded by an all_gather of the per-rank byte counts.

The concatenation of independently compressed blocks is a stream the reference's file
decompressor accepts (it realigns after each end marker, lzs-decompression.c:564-576).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(nblocks: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block range [lo, hi) of ``rank``; remainders go to the low ranks."""
    base, rem = divmod(nblocks, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def scatter_blocks(blocks_on_root: Optional[torch.Tensor], nblocks: int, block_len: int,
                   device: torch.device, src: int = 0, group=None) -> torch.Tensor:
    """Root holds uint8 [nblocks, block_len]; every rank receives its shard_range rows.
    Point-to-point sends issued as one batch, so the root drives all its links at once."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(nblocks, rank, world)
    mine = torch.empty((hi - lo, block_len), dtype=torch.uint8, device=device)
    ops = []
    if rank == src:
        assert blocks_on_root is not None and tuple(blocks_on_root.shape) == (nblocks, block_len)
        for r in range(world):
            rlo, rhi = shard_range(nblocks, r, world)
            if r == src:
                mine.copy_(blocks_on_root[rlo:rhi])
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.isend, blocks_on_root[rlo:rhi].contiguous(), r, group))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, mine, src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return mine


def gather_streams(dense: torch.Tensor, nbytes: int, dst: int = 0, group=None
                   ) -> Tuple[Optional[torch.Tensor], List[int]]:
    """Gather-v of compressed bytes: ``dense[:nbytes]`` of every rank lands on ``dst``,
    concatenated in rank order.  Returns (bytes on dst | None elsewhere, per-rank counts)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = torch.tensor([nbytes], dtype=torch.int64, device=dense.device)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=dense.device) for _ in range(world)]
    dist.all_gather(counts_t, mine, group=group)
    counts = [int(t.item()) for t in counts_t]
    out = None
    ops = []
    if rank == dst:
        out = torch.empty(sum(counts), dtype=torch.uint8, device=dense.device)
        at = 0
        for r in range(world):
            if r == dst:
                out[at:at + counts[r]].copy_(dense[:counts[r]])
            elif counts[r]:
                ops.append(dist.P2POp(dist.irecv, out[at:at + counts[r]], r, group))
            at += counts[r]
    elif nbytes:
        ops.append(dist.P2POp(dist.isend, dense[:nbytes].contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


def gather_lengths(lens: torch.Tensor, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """All ranks' per-block lengths (int32 [blocks_of_rank]) concatenated on ``dst``."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = torch.tensor([lens.numel()], dtype=torch.int64, device=lens.device)
    ns = [torch.zeros(1, dtype=torch.int64, device=lens.device) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(t.item()) for t in ns]
    out = None
    ops = []
    if rank == dst:
        out = torch.empty(sum(ns), dtype=lens.dtype, device=lens.device)
        at = 0
        for r in range(world):
            if r == dst:
                out[at:at + ns[r]].copy_(lens)
            elif ns[r]:
                ops.append(dist.P2POp(dist.irecv, out[at:at + ns[r]], r, group))
            at += ns[r]
    elif lens.numel():
        ops.append(dist.P2POp(dist.isend, lens.contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
