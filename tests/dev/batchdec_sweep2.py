"""Dev aid: lzs_decompress_batch (host buffers) of mid-size batches: by segments (LZS_BATCH_SEG_MB raised) against the
overlapped pipeline of block-decoder launches (the default above 32 MiB of extent)."""
import sys, os, time
os.environ["LZS_DEV_ENV"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
lzs.compress_batch(workload.fill("text", 8))
for cls in ("text", "lowent", "random"):
    for nb in (256, 512, 1024, 2048, 4096):
        x = workload.fill(cls, nb)
        out, n = lzs.compress_batch(x)
        res = []
        for mb in ("16", "1024"):
            os.environ["LZS_BATCH_SEG_MB"] = mb
            best = 1e9
            for _ in range(4):
                t = time.perf_counter(); back, m = lzs.decompress_batch(out, n, 65536); best = min(best, time.perf_counter() - t)
            assert (m == 65536).all() and np.array_equal(back[:, :65536], x)
            res.append(best)
        os.environ.pop("LZS_BATCH_SEG_MB", None)
        print(f"{cls:7} {nb:>5} blocks: block decoder (pipeline from 48 MiB) {res[0]*1e3:7.2f} ms   segments {res[1]*1e3:7.2f} ms", flush=True)
