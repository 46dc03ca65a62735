#!/usr/bin/env python3
"""Development aid: stage times of the many-wavefront single-stream lzs_decompress() at 1 GiB."""
import os, sys, time
os.environ["LZS_STREAM_DEBUG"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np, torch
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
for cls, nblk in (("text", 16384), ("lowent", 16384), ("random", 8192)):
    x = workload.fill(cls, nblk)
    data = x.reshape(-1)
    out, nbytes = lzs.compress_stream(torch.from_numpy(data).cuda())
    comp = bytes(out[:nbytes].cpu().numpy())
    print(f"== {cls} {data.size >> 20} MiB, stream of {len(comp)} bytes", flush=True)
    t = time.time(); back = lzs.decompress(comp, data.size + 9); dt = time.time() - t
    same = len(back) == data.size and np.array_equal(np.frombuffer(back, dtype=np.uint8), data)
    print(f"   round trip {same}; total {dt*1e3:.0f} ms incl. host copies", flush=True)
