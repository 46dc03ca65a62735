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


P2P_PIECE = 1 << 30      # bytes per point-to-point operation: shards and streams of several GiB go as pieces


def _pieces(flat: torch.Tensor, piece: int):
    """``flat`` (1-D uint8) cut into views of at most ``piece`` bytes."""
    n = flat.numel()
    return [flat[at:min(at + piece, n)] for at in range(0, n, piece)] if n else []


def scatter_blocks(blocks_on_root, nblocks: int, block_len: int,
                   device: torch.device, src: int = 0, group=None, out: Optional[torch.Tensor] = None,
                   piece: int = P2P_PIECE) -> torch.Tensor:
    """Root holds the blocks -- one uint8 tensor [nblocks, block_len], or a list with one tensor per
    rank (rank r's rows: how bench.py generates 64 GiB in 8 GiB pieces) -- and every rank receives
    its shard_range rows (into ``out`` if given).  All point-to-point sends of the root are issued
    as ONE batch (one ncclGroup), so the root drives all its xGMI links at once; shards of several
    GiB go as pieces of ``piece`` bytes inside that batch."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(nblocks, rank, world)
    mine = out if out is not None else torch.empty((hi - lo, block_len), dtype=torch.uint8, device=device)
    assert tuple(mine.shape) == (hi - lo, block_len) and mine.dtype == torch.uint8 and mine.is_contiguous()
    ops = []
    if rank == src:
        assert blocks_on_root is not None
        per_rank = isinstance(blocks_on_root, (list, tuple))
        if per_rank:
            assert len(blocks_on_root) == world
        else:
            assert tuple(blocks_on_root.shape) == (nblocks, block_len)
        for r in range(world):
            rlo, rhi = shard_range(nblocks, r, world)
            rows = blocks_on_root[r] if per_rank else blocks_on_root[rlo:rhi]
            assert tuple(rows.shape) == (rhi - rlo, block_len)
            if r == src:
                if rows.data_ptr() != mine.data_ptr():
                    mine.copy_(rows)
            else:
                for v in _pieces(rows.contiguous().view(-1), piece):
                    ops.append(dist.P2POp(dist.isend, v, r, group))
    else:
        for v in _pieces(mine.view(-1), piece):
            ops.append(dist.P2POp(dist.irecv, v, src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return mine


def gather_counts(nbytes: int, device: torch.device, group=None) -> List[int]:
    """Every rank's byte count, on every rank (one all_gather of int64)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([nbytes], dtype=torch.int64, device=device)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts_t, mine, group=group)
    return [int(t.item()) for t in counts_t]


def gather_streams(dense: torch.Tensor, nbytes: int, dst: int = 0, group=None,
                   out: Optional[torch.Tensor] = None, piece: int = P2P_PIECE,
                   counts: Optional[List[int]] = None) -> Tuple[Optional[torch.Tensor], List[int]]:
    """Gather-v of compressed bytes: ``dense[:nbytes]`` of every rank lands on ``dst``,
    concatenated in rank order (into ``out`` if given: uint8, at least the total).  Returns
    (bytes on dst | None elsewhere, per-rank counts).  An all_gather of the byte counts (unless
    the caller already has them: ``counts``), then one batch of point-to-point operations
    (pieces of ``piece`` bytes)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if counts is None:
        counts = gather_counts(nbytes, dense.device, group)
    assert len(counts) == world and counts[rank] == nbytes
    res = None
    ops = []
    if rank == dst:
        total = sum(counts)
        if out is None:
            out = torch.empty(total, dtype=torch.uint8, device=dense.device)
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= total, "gather buffer too small"
        res = out.view(-1)[:total]
        at = 0
        for r in range(world):
            if r == dst:
                res[at:at + counts[r]].copy_(dense[:counts[r]])
            else:
                for v in _pieces(res[at:at + counts[r]], piece):
                    ops.append(dist.P2POp(dist.irecv, v, r, group))
            at += counts[r]
    else:
        for v in _pieces(dense[:nbytes], piece):
            ops.append(dist.P2POp(dist.isend, v, dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return res, counts


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
