"""The N>1 path on CPU: world_size-2 and -3 gloo groups exercise the block sharding,
the input scatter and the variable-length compressed-output gather.  No GPU here, so each
rank's compressor is the oracle standing in for the kernel -- what is under test is the
plumbing (ranges, offsets, ordering), not the codec."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from lzs_compression_amd import sharding, workload


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nblocks, block_len, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O = oracle.oracle()
        full = workload.fill("text", nblocks, block_len) if rank == 0 else None
        root = torch.from_numpy(full) if rank == 0 else None
        mine = sharding.scatter_blocks(root, nblocks, block_len, torch.device("cpu"))
        lo, hi = sharding.shard_range(nblocks, rank, world)
        assert mine.shape == (hi - lo, block_len)
        ref_shard = workload.fill("text", hi - lo, block_len, first_block=lo) if hi > lo else np.zeros((0, block_len), np.uint8)
        assert np.array_equal(mine.numpy(), ref_shard)
        # "compress" the shard (oracle as stand-in), build the dense stream + lengths
        out, out_len, _ = oracle.run_blocks(O, mine.numpy(), threads=2) if hi > lo else (np.zeros((0, 1), np.uint8), np.zeros(0, np.uint32), 0)
        dense = np.concatenate([out[b, :out_len[b]] for b in range(hi - lo)]) if hi > lo else np.zeros(0, np.uint8)
        got, counts = sharding.gather_streams(torch.from_numpy(dense.copy()), len(dense))
        lens = sharding.gather_lengths(torch.from_numpy(out_len.astype(np.int32)))
        if rank == 0:
            want_out, want_len, _ = oracle.run_blocks(O, full, threads=2)
            want = np.concatenate([want_out[b, :want_len[b]] for b in range(nblocks)])
            assert sum(counts) == len(want)
            assert np.array_equal(got.numpy(), want)
            assert np.array_equal(lens.numpy().astype(np.uint32), want_len)
            # the gathered bytes decode block by block at the gathered offsets
            at = 0
            for b in range(nblocks):
                s = bytes(got.numpy()[at:at + int(lens[b])])
                assert O.decompress(s, block_len) == full[b].tobytes()
                at += int(lens[b])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover - surfaced by the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _job_worker(rank, world, port, nb, block_len, piece, chunk, q):
    """The config-5 job (lzs_compression_amd/sharded_job.py, what bench.py --gpus N runs) under gloo:
    CPU tensors, the oracle standing in for the compress kernel, numpy for the compaction."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lzs_compression_amd.sharded_job import ShardedCompressJob
        O = oracle.oracle()
        stride = (oracle.compressed_max(block_len) + 15) // 16 * 16
        sharding.P2P_PIECE = piece                      # several pieces per shard, like 8 GiB in 1 GiB pieces
        calls = []

        def compress(x, slots, lens):
            out, out_len, _ = oracle.run_blocks(O, x.contiguous().numpy(), threads=1)
            slots.zero_()
            slots[:, :out.shape[1]] = torch.from_numpy(out)
            lens.copy_(torch.from_numpy(out_len.astype(np.int32)))
            calls.append(x.shape[0])

        def compact(slots, lens, dense, offsets):
            at = 0
            for b in range(slots.shape[0]):
                n = int(lens[b])
                dense[at:at + n] = slots[b, :n]
                offsets[b] = at
                at += n
            offsets[slots.shape[0]] = at

        job = ShardedCompressJob(nb, block_len, stride, torch.device("cpu"), compress, compact, lambda: None, chunk_blocks=chunk)
        assert job.K == -(-nb // min(nb, chunk or nb)) and job.overlap == (job.K > 1)
        pieces = [torch.from_numpy(workload.fill("text", nb, block_len, first_block=r * nb)) for r in range(world)] if rank == 0 else None
        full = workload.fill("text", nb * world, block_len)
        want_out, want_len, _ = oracle.run_blocks(O, full, threads=2)
        want = np.concatenate([want_out[b, :want_len[b]] for b in range(nb * world)])

        def check():
            mine = pieces[0] if rank == 0 else job.mine
            assert np.array_equal(mine.numpy(), workload.fill("text", nb, block_len, first_block=rank * nb))
            assert len(job.counts) == world and job.nbytes == job.counts[rank]
            assert job.counts == [int(want_len[r * nb:(r + 1) * nb].sum()) for r in range(world)]
            if rank == 0:
                assert sum(job.counts) == len(want) and job.out.numel() == len(want)
                assert np.array_equal(job.out.numpy(), want)
                assert np.array_equal(job.all_lens.numpy().astype(np.uint32), want_len)
            else:
                assert job.out is None and job.all_lens is None

        for _ in range(2):                              # twice: buffers are reused from step to step
            calls.clear()
            times = job.step(pieces)
            assert set(times) >= {"total", "comm_busy", "compute_busy"}
            assert sum(calls) == nb and len(calls) == job.K
            check()
            if rank == 0:
                job.gathered.fill_(0xEE)                # (nothing of the last step may be needed by the next)
        phases = job.serial_phases(pieces)              # the un-overlapped pass leaves the same results
        assert set(phases) == {"scatter", "compress", "gather", "total"}
        check()
        job.step(pieces)                                # ... and the pipelined one after it
        check()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover - surfaced by the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nb,block_len,piece,chunk", [
    (2, 5, 4096, 1 << 30, None),          # un-overlapped: one chunk = the whole shard
    (2, 7, 2048, 1 << 30, 2),             # pipelined, ragged last chunk (2 + 2 + 2 + 1)
    (2, 6, 2048, 1 << 30, 1),             # a chunk per block: K + 2 = 8 stages
    (8, 16, 512, 3000, 4),                # config-5 proportions: 8 ranks, 4 chunks per rank, chunks cut into several pieces
    (8, 16, 512, 3000, None),
    (3, 4, 1024, 1 << 30, 3),
    (1, 6, 2048, 1 << 30, 4),             # world 1: nothing moves between ranks
])
def test_config5_job_scatter_compress_gather_gloo(world, nb, block_len, piece, chunk):
    """The pipelined job (scatter(s) and gather(s-2) in one batch per stage beside compress(s-1))
    and its un-overlapped form, bytes equal to the oracle's concatenation in block order; world 8 at
    config-5 proportions (16 tiny blocks per rank in place of 131072 x 64 KiB)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, nb, block_len, piece, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


@pytest.mark.parametrize("world,nblocks", [(2, 11), (3, 7), (2, 1)])
def test_scatter_compress_gather_gloo(world, nblocks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nblocks, 4096, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 16384, 1048576):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_the_c_entry_partitions_like_the_python_job():
    """lzs_shard_range() (include/lzs/lzs_shard.h, csrc/lzs_rccl.c: what a C host of the sharded job calls) gives the
    ranges sharding.shard_range gives; liblzs.so opens librccl only when a transfer call runs, never at load time."""
    import ctypes
    import subprocess
    import lzs_compression_amd as lzs
    L = lzs.lib()
    L.lzs_shard_range.restype = None
    L.lzs_shard_range.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    for n in (0, 1, 7, 100, 16384, 1048576, 1048577):
        for world in (1, 2, 3, 5, 8):
            for r in range(world):
                lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
                L.lzs_shard_range(n, r, world, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == sharding.shard_range(n, r, world), (n, r, world)
    so = os.path.join(os.path.dirname(os.path.abspath(lzs.__file__)), "liblzs.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "rccl" not in needed and "nccl" not in needed


# ---- the C half of the job at world > 1: csrc/lzs_rccl.c UNCHANGED (but for pieces of 3000 bytes instead of
# 1 GiB, so that a shard spans several), its librccl a shared-memory stand-in between forked processes that the library opens
# through LZS_RCCL_LIBRARY (tests/cpu_shim/fake_rccl.c), its device tests/cpu_shim/lzs_cpu_shim.c; under ASan + UBSan, leak
# detection on.  What the root gathers must be the oracle's streams of all blocks back to back.
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_shim")


@pytest.fixture(scope="module")
def rccl_built():
    import subprocess
    r = subprocess.run(["make", "-C", _SHIM, os.path.join(_SHIM, "_build", "rccl_asan"), os.path.join(_SHIM, "_build", "libfake_rccl.so")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return os.path.join(_SHIM, "_build")


@pytest.mark.parametrize("world,root,nblocks,block_len", [
    (2, 0, 9, 4096),        # ragged: 5 + 4 blocks; shards of 16-20 KB in pieces of 3000 bytes
    (2, 1, 16, 4096),       # root is not rank 0
    (3, 2, 37, 4096),       # nblocks % world != 0, the root last: its own rows go behind everybody else's
    (8, 5, 67, 2500),       # eight ranks, a block length that is no multiple of anything
    (8, 0, 5, 4096),        # fewer blocks than ranks: three ranks have nothing to receive or send
    (1, 0, 4, 4096),        # one rank: no transfer at all
])
def test_the_c_scatter_and_gather_between_forked_ranks_over_a_stand_in_for_rccl(rccl_built, world, root, nblocks, block_len):
    import re
    import subprocess
    env = {k: v for k, v in os.environ.items() if not k.startswith("LZS_")}
    env.update(LZS_RCCL_LIBRARY=os.path.join(rccl_built, "libfake_rccl.so"), ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(rccl_built, "rccl_asan"), str(world), str(root), str(nblocks), str(block_len)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "0 failure(s)" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    m = re.search(r"(\d+) sends, (\d+) receives in (\d+) groups, (\d+) bytes between ranks, (\d+) operations outside a group", r.stdout)
    sends, recvs, groups, moved, outside = map(int, m.groups())
    # what must have crossed: every block that is not the root's own, in pieces of <= 3000 bytes, and the peers' streams back
    lo, hi = sharding.shard_range(nblocks, root, world)
    scattered = (nblocks - (hi - lo)) * block_len
    assert sends == recvs and outside == 0
    if world == 1:
        assert sends == 0 and moved == 0
    else:
        pieces = sum(-(-((h - l) * block_len) // 3000) for l, h in (sharding.shard_range(nblocks, k, world) for k in range(world) if k != root))
        assert moved > scattered and sends >= pieces + sum(1 for k in range(world) if k != root and sharding.shard_range(nblocks, k, world)[1] > sharding.shard_range(nblocks, k, world)[0])
        assert groups <= 2 * world                     # one group per rank and call: the root's sends to all its peers are ONE group
