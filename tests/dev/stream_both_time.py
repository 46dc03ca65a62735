import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, lzs_compression_amd as lzs
from lzs_compression_amd import workload
for cls in ("text", "random"):
    x = torch.from_numpy(workload.fill(cls, 16384)).cuda().reshape(-1)
    buf, nbytes = lzs.compress_stream(x)
    t = time.perf_counter(); buf, nbytes = lzs.compress_stream(x, buf); dt = time.perf_counter() - t
    back, got = lzs.decompress_stream(buf[:nbytes], x.numel() + 16)
    t = time.perf_counter(); back, got = lzs.decompress_stream(buf[:nbytes], x.numel() + 16, back); dd = time.perf_counter() - t
    print(cls, "stream compress %.2f ms, stream decompress %.2f ms" % (dt * 1e3, dd * 1e3), got == x.numel())
