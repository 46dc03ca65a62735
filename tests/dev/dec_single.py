import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, lzs_compression_amd as lzs
from lzs_compression_amd import workload
data = bytes(workload.fill("text", 64).reshape(-1))      # 4 MiB
comp = lzs.compress(data)
lzs.decompress(comp[:1000], 4096)
t = time.time(); back = lzs.decompress(comp, len(data)); dt = time.time() - t
print("single-stream decompress 4 MiB:", back == data, "%.1f ms = %.2f MB/s" % (dt * 1e3, len(data) / dt / 1e6))
