"""Dev aid: lzs_compress_batch on host buffers below the pipeline's threshold (1024 blocks), call by call."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
L = lzs.lib()
cap = lzs.compressed_max(65536)
for nb in (512, 768, 1024, 2048, 4096):
    x = workload.fill("text", nb)
    out = np.zeros((nb, cap), dtype=np.uint8); out_len = np.zeros(nb, dtype=np.uint32)
    ts = []
    for _ in range(8):
        t = time.perf_counter(); rc = L.lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, nb); ts.append((time.perf_counter() - t) * 1e3); assert rc == 0
    print(nb, "blocks, ms per call:", " ".join(f"{t:.1f}" for t in ts), flush=True)
