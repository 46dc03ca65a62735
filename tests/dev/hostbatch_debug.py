"""Dev aid: where the time of a large host-buffer batch goes (LZS_STREAM_DEBUG=1: lzs_pipeline.c's stage times)."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ["LZS_STREAM_DEBUG"] = "1"
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
L = lzs.lib()
cap = lzs.compressed_max(65536)
nb = 16384
x = workload.fill("text", nb)
out = np.zeros((nb, cap), dtype=np.uint8); out_len = np.zeros(nb, dtype=np.uint32)
back = np.zeros((nb, 65536), dtype=np.uint8); back_len = np.zeros(nb, dtype=np.uint32)
for _ in range(3):
    t = time.perf_counter(); rc = L.lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, nb); print("compress", time.perf_counter() - t, rc, flush=True)
for _ in range(3):
    t = time.perf_counter(); rc = L.lzs_decompress_batch(back.ctypes.data, 65536, 65536, back_len.ctypes.data, out.ctypes.data, cap, out_len.ctypes.data, cap, nb); print("decompress", time.perf_counter() - t, rc, flush=True)
