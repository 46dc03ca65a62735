import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
x = torch.from_numpy(workload.fill("text", 16384)).cuda()
slots, lens = lzs.compress_blocks(x)
torch.cuda.synchronize()
for nb in (88, 704, 1408, 2816, 4224, 5632, 8192, 16384):
    s, l = slots[:nb].contiguous(), lens[:nb].contiguous()
    out = None
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); back, bl = lzs.decompress_blocks(s, l, 65536); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
    print(f"{nb:6} blocks: {ms:7.2f} ms  {nb*65536/ms/1e6:7.1f} GB/s", flush=True)
# two launches side by side on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for nb in (2816, 5632):
    a = slots[:nb].contiguous(); la = lens[:nb].contiguous(); b = slots[nb:2*nb].contiguous(); lb = lens[nb:2*nb].contiguous()
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(s1): r1 = lzs.decompress_blocks(a, la, 65536)
    with torch.cuda.stream(s2): r2 = lzs.decompress_blocks(b, lb, 65536)
    torch.cuda.synchronize()
    print(f"2 x {nb} blocks on two streams: {(time.perf_counter()-t)*1e3:.2f} ms", flush=True)
