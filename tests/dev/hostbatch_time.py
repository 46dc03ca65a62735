"""Dev aid: PCIe-inclusive rate of the host-buffer batch calls."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
lzs.compress_batch(workload.fill("text", 8))
for nb in (256, 4096, 16384):
    x = workload.fill("text", nb)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); out, n = lzs.compress_batch(x); best = min(best, time.perf_counter() - t)
    bd = 1e9
    for _ in range(3):
        t = time.perf_counter(); back, m = lzs.decompress_batch(out, n, 65536); bd = min(bd, time.perf_counter() - t)
    assert (m == 65536).all() and np.array_equal(back[:, :65536], x)
    print(f"{nb:>6} blocks: compress_batch {x.size/best/1e9:6.2f} GB/s ({best*1e3:7.1f} ms)   decompress_batch {x.size/bd/1e9:6.2f} GB/s ({bd*1e3:7.1f} ms)", flush=True)
