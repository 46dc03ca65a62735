#!/usr/bin/env python3
"""Development aid: stage times (LZS_STREAM_DEBUG) of lzs_decompress_stream_device on 1 GiB of text, second call (warm)."""
import os, sys
os.environ["LZS_DEV_ENV"] = "1"       # the library reads its switches on every call
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
cls = sys.argv[1] if len(sys.argv) > 1 else "text"
x = torch.from_numpy(workload.fill(cls, 16384).reshape(-1)).cuda()
comp, nbytes = lzs.compress_stream(x)
out = torch.empty(x.numel() + 64, dtype=torch.uint8, device="cuda")
n = 0
for rep in range(3):
    print(f"== call {rep}", file=sys.stderr, flush=True)
    os.environ["LZS_STREAM_DEBUG"] = "1" if rep == 2 else ""
    _, n = lzs.decompress_stream(comp[:nbytes], out.numel(), out)
    torch.cuda.synchronize()
print("round trip", bool(torch.equal(out[:n], x)))
