"""Dev aid: stage times (LZS_STREAM_DEBUG) of lzs_decompress_batch by segments for a mid-size host batch."""
import sys, os, time
os.environ["LZS_DEV_ENV"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = workload.fill("text", nb)
out, n = lzs.compress_batch(x)
os.environ["LZS_BATCH_SEG_MB"] = "1024"
back = np.empty((nb, 65536), dtype=np.uint8); m = np.empty(nb, dtype=np.uint32)
for rep in range(3):
    os.environ["LZS_STREAM_DEBUG"] = "1" if rep == 2 else ""
    t = time.perf_counter(); b2, m2 = lzs.decompress_batch(out, n, 65536); dt = time.perf_counter() - t
    print(f"call {rep}: {dt*1e3:.2f} ms", file=sys.stderr, flush=True)
assert (m2 == 65536).all() and np.array_equal(b2[:, :65536], x)
