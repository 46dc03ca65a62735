"""BASELINE.json configs[4] as a job: blocks generated at the root GPU, scattered over xGMI,
compressed by every rank, compacted and gathered back.

Blocks are independent streams (reference lzs-compression.c:291-299, 449-466), so the compute
needs no collective; data moves only where north_star says it does: input scatter from the root,
compressed-output gather to the root.  The job is PIPELINED over chunks of blocks, so a step is
bounded by its slowest phase and not by the sum of the three:

    chunk j of every rank:   scatter(j)  ->  compress(j) + compact(j)  ->  gather(j)
    stage s = 0 .. K+1:      C(s) = ONE batch of point-to-point operations (one ncclGroup):
                                    scatter(s) root -> peers  and  gather(s-2) peers -> root,
                                    both directions of every xGMI link of the root at once
                             X(s-1) = compress + compact of chunk s-1, on the compute lane,
                                    running beside C(s)

    comm lane    : C(0) C(1) q0 C(2) q1 C(3) q2 ...        (q_j: all_gather of the byte counts of
    compute lanes:      X(0)    X(1)    X(2)    ...         chunk j, one int64 per rank)
                                                            (two lanes, chunks alternate: X(j+1) starts in X(j)'s tail)

Every rank issues the same collectives in the same order on ONE communicator: there is nothing
to deadlock.  Dependencies on the device are events (X(j) waits for C(j), q_j for X(j)); the host
only waits where it needs a number: the byte counts of chunk s-2 before it can post the gather of
that chunk in C(s) -- by then C(s-1) and X(s-1) are already queued, so neither lane runs dry
(one host round trip per stage on the comm lane, none on the compute lane).

What lands where.  A rank compacts chunk j into a region of its own (`dense`, worst case per
chunk, so no destination depends on a count).  The root keeps one worst-case region per rank in
`gathered`; chunk j of rank r is received behind that rank's earlier chunks.  After the last stage
the root moves the regions of ranks 1.. down against each other (device copies in pieces that do
not overlap), so `out` is the plain concatenation of all blocks' streams in block order --
a file the reference's decompressor reads (lzs-decompression.c:564-576) -- and `all_lens` its index.

``chunk_blocks >= blocks_per_rank`` gives the un-overlapped job (scatter, then compress, then
gather): what round 2 measured, and what `serial_phases()` times phase by phase.

The compressor and the compaction are passed in, so the same control flow runs under ``gloo`` on
CPU tensors in tests/test_sharding.py with the oracle standing in for the kernel.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

from . import sharding


class _Lane:
    """A timeline of the device.  On a GPU: a side stream; whatever is issued inside ``with lane:``
    is ordered on it (torch.distributed queues NCCL work behind the current stream, and
    ``work.wait()`` makes the current stream wait for that work without blocking the host).
    On the CPU (gloo) calls complete before they return and a lane is nothing."""

    def __init__(self, device: torch.device):
        self.cuda = device.type == "cuda"
        self.stream = torch.cuda.Stream(device) if self.cuda else None
        self._ctx = None

    def __enter__(self):
        if self.cuda:
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.cuda:
            self._ctx.__exit__(*exc)
            self._ctx = None
        return False

    def mark(self):
        """An event at the lane's present end (None on the CPU)."""
        if not self.cuda:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.stream)
        return ev

    def after(self, ev) -> None:
        """What is issued on the lane from now on waits for ``ev``."""
        if self.cuda and ev is not None:
            self.stream.wait_event(ev)

    def after_current(self) -> None:
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())


def _ms(a, b) -> float:
    return float(a.elapsed_time(b)) if a is not None and b is not None else 0.0


class ShardedCompressJob:
    def __init__(self, blocks_per_rank: int, block_len: int, slot_stride: int, device: torch.device,
                 compress: Callable, compact: Callable, sync: Callable[[], None], group=None, root: int = 0,
                 chunk_blocks: Optional[int] = None):
        """``compress(x, slots, lens)`` fills slots [n, slot_stride] / lens [n] int32 from x [n, block_len];
        ``compact(slots, lens, dense, offsets)`` packs the first lens[b] bytes of every slot into
        ``dense`` and writes the exclusive prefix sums into ``offsets`` (int64 [n + 1]; offsets[n] =
        the byte total) -- both asynchronous on the current stream, neither returns anything the
        host must wait for; ``sync()`` waits for the device (no-op on CPU).
        ``chunk_blocks``: blocks per pipeline chunk (default: the whole shard = no overlap)."""
        self.nb, self.block_len, self.slot_stride = blocks_per_rank, block_len, slot_stride
        self.device, self.group, self.root = device, group, root
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.compress, self.compact, self.sync = compress, compact, sync
        self.total_blocks = self.nb * self.world
        self.cb = max(1, min(self.nb, chunk_blocks or self.nb))
        self.K = (self.nb + self.cb - 1) // self.cb
        self.overlap = self.K > 1
        u8 = dict(dtype=torch.uint8, device=device)
        need = self.memory_needed(self.nb, block_len, slot_stride, self.world, self.cb, self.rank == root)
        if device.type == "cuda":
            free, _total = torch.cuda.mem_get_info(device)
            if free < need + (1 << 30):
                raise RuntimeError(
                    f"rank {self.rank}: the sharded job needs {need / 1e9:.1f} GB of HBM on this GPU "
                    f"({'root: one worst-case region per rank for the gathered streams' if self.rank == root else 'landing buffer + slots + streams of one shard'}) "
                    f"but only {free / 1e9:.1f} GB are free; use fewer --blocks per GPU")
        # the root compresses its own piece where it was generated: only peers need a landing buffer
        self.mine = None if self.rank == root else torch.empty((self.nb, block_len), **u8)
        # two sets of slots, two compute lanes: chunk j on lane j % 2, so the first workgroups of
        # compress(j+1) fill the CUs that the last ones of compress(j) leave idle (a launch of 8192
        # blocks on 1280 resident workgroups ends with a tail: 61.4 -> 63+ GB/s at world 1)
        self.slots = [torch.empty((self.cb, slot_stride), **u8) for _ in range(2 if self.K > 1 else 1)]
        self.lens = torch.empty(self.nb, dtype=torch.int32, device=device)
        self.dense = torch.empty(self.K * self.cb * slot_stride, **u8)         # chunk j at j * cb * slot_stride
        self.offsets = torch.zeros((self.K, self.cb + 1), dtype=torch.int64, device=device)
        self.q = torch.zeros((self.K, self.world), dtype=torch.int64, device=device)   # byte counts per chunk and rank
        self.region = self.nb * slot_stride
        if self.rank == root:
            self.gathered = torch.empty(self.world * self.region, **u8)        # allocated once, outside any timed region
            self.all_lens = torch.empty(self.total_blocks, dtype=torch.int32, device=device)
        else:
            self.gathered = self.all_lens = None
        self.counts: List[int] = []         # bytes per rank, last step
        self.chunk_counts: List[List[int]] = []
        self.nbytes = 0
        self.out = None                     # root: the gathered streams of the last step (a view of `gathered`)
        self.comm = _Lane(device)
        self.comp = [_Lane(device) for _ in self.slots]
        self.last_stage_ms: Dict[str, List[float]] = {}

    @staticmethod
    def memory_needed(nb: int, block_len: int, slot_stride: int, world: int, cb: int, is_root: bool) -> int:
        K = (nb + cb - 1) // cb
        n = (2 if K > 1 else 1) * cb * slot_stride + K * cb * slot_stride + nb * 4 + K * (cb + 1) * 8
        return n + (world * nb * slot_stride + world * nb * 4 if is_root else nb * block_len)

    # ------------------------------------------------------------------ pieces of a step
    def _chunk(self, j: int):
        lo = j * self.cb
        return lo, min(self.nb, lo + self.cb)

    def _dense_chunk(self, j: int) -> torch.Tensor:
        base = j * self.cb * self.slot_stride
        return self.dense[base:base + self.cb * self.slot_stride]

    def _batch(self, ops) -> None:
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def _scatter_ops(self, j: int, pieces_on_root) -> list:
        """Point-to-point operations of scatter(j): chunk j of every peer's shard, root -> peer."""
        ops, P, g, root = [], sharding.P2P_PIECE, self.group, self.root
        lo, hi = self._chunk(j)
        if self.rank == root:
            for r in range(self.world):
                if r != root:
                    for v in sharding._pieces(pieces_on_root[r][lo:hi].reshape(-1), P):
                        ops.append(dist.P2POp(dist.isend, v, r, g))
        else:
            for v in sharding._pieces(self.mine[lo:hi].view(-1), P):
                ops.append(dist.P2POp(dist.irecv, v, root, g))
        return ops

    def _gather_ops(self, j: int, run_off: List[int]) -> list:
        """Point-to-point operations of gather(j): the compacted streams of chunk j and their
        lengths, peer -> root, behind that rank's earlier chunks (the root's own: device copies)."""
        ops, P, g, root = [], sharding.P2P_PIECE, self.group, self.root
        lo, hi = self._chunk(j)
        cnt = self.chunk_counts[j]
        if self.rank == root:
            for r in range(self.world):
                at = r * self.region + run_off[r]
                dst = self.gathered[at: at + cnt[r]]
                ldst = self.all_lens[r * self.nb + lo: r * self.nb + hi]
                if r == root:
                    dst.copy_(self._dense_chunk(j)[:cnt[r]])
                    ldst.copy_(self.lens[lo:hi])
                else:
                    for v in sharding._pieces(dst, P):
                        ops.append(dist.P2POp(dist.irecv, v, r, g))
                    ops.append(dist.P2POp(dist.irecv, ldst, r, g))
                run_off[r] += cnt[r]
        else:
            for v in sharding._pieces(self._dense_chunk(j)[:cnt[self.rank]], P):
                ops.append(dist.P2POp(dist.isend, v, root, g))
            ops.append(dist.P2POp(dist.isend, self.lens[lo:hi], root, g))
        return ops

    def _comm_stage(self, s: int, pieces_on_root, run_off: List[int]) -> None:
        """C(s): scatter(s) and gather(s-2) as ONE batch of point-to-point operations."""
        ops = self._scatter_ops(s, pieces_on_root) if s < self.K else []
        if 0 <= s - 2 < self.K:
            ops += self._gather_ops(s - 2, run_off)
        self._batch(ops)

    def _count_of(self, j: int) -> torch.Tensor:
        """The byte total compact() left for chunk j: offsets[j, n], as a one-element view."""
        lo, hi = self._chunk(j)
        return self.offsets[j, hi - lo: hi - lo + 1]

    def _compute_chunk(self, j: int, x_all: torch.Tensor) -> None:
        lo, hi = self._chunk(j)
        slots = self.slots[j % len(self.slots)][:hi - lo]
        self.compress(x_all[lo:hi], slots, self.lens[lo:hi])
        self.compact(slots, self.lens[lo:hi], self._dense_chunk(j), self.offsets[j, :hi - lo + 1])

    def _finish(self) -> None:
        self.counts = [sum(c[r] for c in self.chunk_counts) for r in range(self.world)]
        self.nbytes = self.counts[self.rank]
        if self.rank == self.root:
            self._close_up()

    def _close_up(self) -> None:
        """Root: the regions of ranks 1.. moved down against each other: `out` = all streams in block
        order.  A region moves towards lower addresses by d >= region - counts[0] > 0 bytes; copying it
        in ascending pieces of at most d bytes, no piece overlaps its own destination."""
        at = self.counts[0]
        for r in range(1, self.world):
            src, n = r * self.region, self.counts[r]
            d = src - at
            if d > 0:
                step = min(d, 1 << 31)
                for o in range(0, n, step):
                    m = min(step, n - o)
                    self.gathered[at + o: at + o + m].copy_(self.gathered[src + o: src + o + m])
            at += n
        self.out = self.gathered[:at]

    # ------------------------------------------------------------------ a step
    def step(self, pieces_on_root: Optional[list]) -> Dict[str, float]:
        """One pass of the job.  ``pieces_on_root``: on the root, one tensor [nb, block_len] per rank."""
        K, root = self.K, self.root
        x_all = pieces_on_root[root] if self.rank == root else self.mine
        assert tuple(x_all.shape) == (self.nb, self.block_len)
        self.chunk_counts = [None] * K
        run_off = [0] * self.world
        ev_c, ev_c0, ev_x, ev_x0, qwork = [None] * (K + 2), [None] * (K + 2), [None] * K, [None] * K, [None] * K
        t0 = time.perf_counter()
        self.comm.after_current()
        for lane in self.comp:
            lane.after_current()
        for s in range(K + 2):
            j = s - 2
            if 0 <= j < K:                       # the byte counts of chunk j: the one place the host waits
                with self.comm:
                    qwork[j].wait()
                    self.chunk_counts[j] = [int(v) for v in self.q[j].tolist()]
            with self.comm:
                ev_c0[s] = self.comm.mark()
                self._comm_stage(s, pieces_on_root, run_off)
                ev_c[s] = self.comm.mark()
                if 1 <= s <= K:                  # q_{s-1}, queued behind C(s): it needs X(s-1), which runs beside C(s)
                    self.comm.after(ev_x[s - 1])
                    qwork[s - 1] = dist.all_gather_into_tensor(self.q[s - 1], self._count_of(s - 1), group=self.group, async_op=True)
            if s < K:                            # X(s): waits for C(s), runs beside C(s+1) (and the tail of X(s-1))
                lane = self.comp[s % len(self.comp)]
                with lane:
                    lane.after(ev_c[s])
                    ev_x0[s] = lane.mark()
                    self._compute_chunk(s, x_all)
                    ev_x[s] = lane.mark()
        with self.comm:
            self._finish()
        self.sync()
        t1 = time.perf_counter()
        self.last_stage_ms = {"comm": [_ms(a, b) for a, b in zip(ev_c0, ev_c)], "compute": [_ms(a, b) for a, b in zip(ev_x0, ev_x)]}
        return {"total": t1 - t0, "comm_busy": sum(self.last_stage_ms["comm"]) * 1e-3,
                "compute_busy": sum(self.last_stage_ms["compute"]) * 1e-3}

    # ------------------------------------------------------------------ the phases alone
    def serial_phases(self, pieces_on_root: Optional[list]) -> Dict[str, float]:
        """The same job with nothing overlapped: scatter of the whole shard, then compress + compact,
        then the gather, each ended by a device synchronize and timed on the host -- what each phase
        costs alone (round 2's job; the basis of scatter_GBps / compute_only_GBps / gather_GBps at
        every N).  Leaves the same results as step()."""
        K, root = self.K, self.root
        x_all = pieces_on_root[root] if self.rank == root else self.mine
        self.chunk_counts = [None] * K
        run_off = [0] * self.world
        self.sync()
        t0 = time.perf_counter()
        for j in range(K):                                    # SCATTER
            self._batch(self._scatter_ops(j, pieces_on_root))
        self.sync()
        t1 = time.perf_counter()
        for j in range(K):                                    # COMPRESS (+ compaction of the slots)
            self._compute_chunk(j, x_all)
        self.sync()
        t2 = time.perf_counter()
        works = [dist.all_gather_into_tensor(self.q[j], self._count_of(j), group=self.group, async_op=True) for j in range(K)]
        for j in range(K):                                    # GATHER
            works[j].wait()
            self.chunk_counts[j] = [int(v) for v in self.q[j].tolist()]
            self._batch(self._gather_ops(j, run_off))
        self._finish()
        self.sync()
        t3 = time.perf_counter()
        return {"scatter": t1 - t0, "compress": t2 - t1, "gather": t3 - t2, "total": t3 - t0}
