"""Dev aid: lzs_decompress_batch (host buffers) by batch size: segments vs one wavefront per block."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
lzs.compress_batch(workload.fill("text", 8))
for cls in ("text", "lowent", "random"):
    for nb in (4, 16, 64, 256, 512, 1024, 2048):
        x = workload.fill(cls, nb)
        out, n = lzs.compress_batch(x)
        res = []
        for mode in ("seg", "one"):
            os.environ.pop("LZS_ONE_WAVE", None)
            if mode == "one": os.environ["LZS_ONE_WAVE"] = "1"
            best = 1e9
            for _ in range(3):
                t = time.perf_counter(); back, m = lzs.decompress_batch(out, n, 65536); best = min(best, time.perf_counter() - t)
            assert (m == 65536).all() and np.array_equal(back[:, :65536], x)
            res.append(best)
        os.environ.pop("LZS_ONE_WAVE", None)
        print(f"{cls:7} {nb:>5} blocks: segments {res[0]*1e3:7.2f} ms   one wavefront per block {res[1]*1e3:7.2f} ms", flush=True)
