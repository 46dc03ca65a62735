"""BASELINE.json configs[4] as a job: blocks generated at the root GPU, scattered over xGMI,
compressed by every rank, compacted and gathered back.

    root:   pieces[r] = rank r's blocks, generated in HBM (workload.fill_device)
    step(): SCATTER   one batched group of point-to-point sends, root -> 7 peers at once
            COMPRESS  lzs_compress_batch_device on the rank's shard (no collective: blocks are
                      independent streams, reference lzs-compression.c:291-299, 449-466)
            GATHER    lzs_compact_device, all_gather of byte counts, gather-v of the dense
                      streams and of the per-block lengths to the root

The three phases run one after the other and are timed separately (a device synchronize ends
each); the compressor and the compaction are passed in, so the same control flow runs under
``gloo`` on CPU tensors in tests/test_sharding.py with the oracle standing in for the kernel.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from . import sharding


class ShardedCompressJob:
    def __init__(self, blocks_per_rank: int, block_len: int, slot_stride: int, device: torch.device,
                 compress: Callable, compact: Callable, sync: Callable[[], None], group=None, root: int = 0):
        """``compress(x, slots, lens)`` fills slots [nb, slot_stride] / lens [nb] int32 from x [nb, block_len];
        ``compact(slots, lens, dense, offsets)`` packs them and returns the byte total (a host int);
        ``sync()`` waits for the device (no-op on CPU)."""
        self.nb, self.block_len, self.slot_stride = blocks_per_rank, block_len, slot_stride
        self.device, self.group, self.root = device, group, root
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.compress, self.compact, self.sync = compress, compact, sync
        self.total_blocks = self.nb * self.world
        u8 = dict(dtype=torch.uint8, device=device)
        # the root compresses its own piece in place: only peers need a landing buffer
        self.mine = None if self.rank == root else torch.empty((self.nb, block_len), **u8)
        self.slots = torch.empty((self.nb, slot_stride), **u8)
        self.lens = torch.empty(self.nb, dtype=torch.int32, device=device)
        self.dense = torch.empty(self.nb * slot_stride, **u8)
        self.offsets = torch.empty(self.nb + 1, dtype=torch.int64, device=device)
        self.gathered = None            # root: allocated at the first gather, grown when needed
        self.all_lens = None
        self.counts: List[int] = []
        self.nbytes = 0
        self.out = None                 # root: the gathered streams of the last step (a view of `gathered`)

    def step(self, pieces_on_root: Optional[list]) -> Dict[str, float]:
        """One pass of the job.  ``pieces_on_root``: on the root, one tensor [nb, block_len] per rank."""
        t0 = time.perf_counter()
        # ---- SCATTER
        if self.rank == self.root:
            x = sharding.scatter_blocks(pieces_on_root, self.total_blocks, self.block_len, self.device,
                                        src=self.root, group=self.group, out=pieces_on_root[self.root])
        else:
            x = sharding.scatter_blocks(None, self.total_blocks, self.block_len, self.device,
                                        src=self.root, group=self.group, out=self.mine)
        self.sync()
        t1 = time.perf_counter()
        # ---- COMPRESS
        self.compress(x, self.slots, self.lens)
        self.sync()
        t2 = time.perf_counter()
        # ---- GATHER
        self.nbytes = int(self.compact(self.slots, self.lens, self.dense, self.offsets))
        self.counts = sharding.gather_counts(self.nbytes, self.device, self.group)
        if self.rank == self.root and (self.gathered is None or self.gathered.numel() < sum(self.counts)):
            self.gathered = None                       # (free first: tens of GB at full size)
            self.gathered = torch.empty(sum(self.counts) + sum(self.counts) // 64 + 4096, dtype=torch.uint8, device=self.device)
        self.out, _ = sharding.gather_streams(self.dense, self.nbytes, dst=self.root, group=self.group,
                                              out=self.gathered, counts=self.counts)
        self.all_lens = sharding.gather_lengths(self.lens, dst=self.root, group=self.group)
        self.sync()
        t3 = time.perf_counter()
        return {"scatter": t1 - t0, "compress": t2 - t1, "gather": t3 - t2, "total": t3 - t0}
