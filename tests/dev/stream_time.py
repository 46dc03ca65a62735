#!/usr/bin/env python3
"""Development aid: stage times of the multi-workgroup single-stream path (LZS_STREAM_DEBUG)."""
import os, sys, time
os.environ["LZS_STREAM_DEBUG"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
for cls, nblk in (("text", 128), ("text", 4096), ("lowent", 4096), ("random", 4096), ("text", 16384)):
    data = bytes(workload.fill(cls, nblk).reshape(-1))
    lzs.compress(data[:1 << 20])
    print(f"== {cls} {len(data) >> 20} MiB", flush=True)
    t = time.time(); out = lzs.compress(data); dt = time.time() - t
    print(f"   total {dt*1e3:.1f} ms = {len(data)/dt/1e9:.2f} GB/s incl. host copies; {len(out)} bytes", flush=True)
    if nblk <= 4096:
        back = lzs.decompress(out, len(data) + 8)
        print("   round trip", back == data, flush=True)
