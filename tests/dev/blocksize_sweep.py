#!/usr/bin/env python3
"""The batch entries by BLOCK SIZE: lzs_compress_batch_device / lzs_decompress_batch_device on
1 GiB of each class cut into blocks of 1.5 KiB, 4 KiB, 16 KiB, 64 KiB and 1 MiB -- the reference's own small-block row
(4 KiB blocks; its file tool works in 512-byte pieces, c/src/utils/lzs-compress.c:28-32, and
RFC 1974 / 2395 compress packets) beside the 64 KiB blocks the headline is quoted on.

Every block is decoded again on the device and compared with its input; a sample of the blocks (SAMPLE_MB of input per
configuration, default 64; ALL of them at 4 KiB) is compared byte for byte with the oracle on the host's cores.

usage (GPU box): python tests/dev/blocksize_sweep.py [total_MiB=1024] > gpurun_out/blocksize_sweep.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lzs_compression_amd as lzs          # noqa: E402
import oracle                              # noqa: E402

TOTAL = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
SIZES = [int(s) for s in os.environ.get("SIZES", "1536,4096,16384,65536,1048576").split(",")]
CLASSES = os.environ.get("CLASSES", "text,lowent,random").split(",")
SAMPLE = int(os.environ.get("SAMPLE_MB", "64")) << 20
REPS = int(os.environ.get("REPS", "5"))
THREADS = len(os.sched_getaffinity(0))


def timed(fn):
    best = 1e9
    for _ in range(REPS):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    torch.cuda.set_device(0)
    print(lzs.backend_info())
    print(f"# total {TOTAL >> 20} MiB per configuration, best of {REPS} launches (HIP events on torch's stream, which the"
          f" library launches on); oracle check on {THREADS} host threads")
    print(f"{'class':8} {'block':>8} {'blocks':>8} {'ratio':>7} {'compress GB/s':>14} {'decompress GB/s out':>20} "
          f"{'oracle-checked blocks':>22}")
    orc = oracle.oracle()
    for cls in CLASSES:
        for bs in SIZES:
            nb = TOTAL // bs
            x = lzs.workload.fill_device(cls, nb, bs)
            cap = lzs.compressed_max(bs)
            stride = (cap + 15) // 16 * 16
            slots = torch.empty((nb, stride), dtype=torch.uint8, device="cuda")
            lens = torch.empty(nb, dtype=torch.int32, device="cuda")
            back = torch.empty((nb, bs), dtype=torch.uint8, device="cuda")
            back_len = torch.empty(nb, dtype=torch.int32, device="cuda")
            lzs.compress_blocks(x, out=slots, out_len=lens)                      # warm
            torch.cuda.synchronize()
            tc = timed(lambda: lzs.compress_blocks(x, out=slots, out_len=lens))
            lzs.decompress_blocks(slots, lens, bs, out=back, out_len=back_len)
            torch.cuda.synchronize()
            td = timed(lambda: lzs.decompress_blocks(slots, lens, bs, out=back, out_len=back_len))
            assert bool((back_len == bs).all()) and torch.equal(back, x), f"{cls} {bs}: round trip differs"
            ratio = float(lens.sum().item()) / (nb * bs)
            # the oracle on a sample (all of it at 4 KiB): lengths and bytes of every block of the sample
            ns = nb if bs == 4096 and cls == "text" else max(1, min(nb, SAMPLE // bs))
            step = nb // ns
            t0 = time.time()
            for lo in range(0, ns, 16384):                                       # (pieces: the host's memory is not the device's)
                idx = torch.arange(lo, min(ns, lo + 16384), device="cuda") * step
                hx = x[idx].cpu().numpy()
                hs, hl = slots[idx].cpu().numpy(), lens[idx].cpu().numpy()
                want, want_len, _ = oracle.run_blocks(orc, hx, threads=THREADS)
                assert (hl == want_len).all(), f"{cls} {bs}: lengths differ from the oracle"
                m = np.arange(want.shape[1])[None, :] < want_len[:, None]        # a row counts up to its length
                assert not ((hs[:, :want.shape[1]] != want) & m).any(), f"{cls} {bs}: bytes differ from the oracle"
            print(f"{cls:8} {bs:8d} {nb:8d} {ratio:7.4f} {nb * bs / tc / 1e6:14.2f} {nb * bs / td / 1e6:20.2f} "
                  f"{ns:14d} ({time.time() - t0:.0f} s)", flush=True)
            del x, slots, lens, back, back_len
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
