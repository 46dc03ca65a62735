"""Dev aid: host-buffer batches of 64 .. 1024 blocks through the C API (output arrays allocated once): the
one-after-the-other route against the overlapped one (LZS_PIPE_MIN_MB lowers its threshold), compress and decompress."""
import sys, os, time, ctypes
os.environ["LZS_DEV_ENV"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
L = lzs.lib()
cap = lzs.compressed_max(65536)
for nb in (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096):
    x = workload.fill("text", nb)
    out = np.zeros((nb, cap), dtype=np.uint8); out_len = np.zeros(nb, dtype=np.uint32)
    back = np.zeros((nb, 65536), dtype=np.uint8); back_len = np.zeros(nb, dtype=np.uint32)
    res = []
    for mb in ("4096", "4"):
        os.environ["LZS_PIPE_MIN_MB"] = mb
        tc, td = [], []
        for _ in range(5):
            t = time.perf_counter(); rc = L.lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, nb); tc.append(time.perf_counter() - t); assert rc == 0
            t = time.perf_counter(); rc = L.lzs_decompress_batch(back.ctypes.data, 65536, 65536, back_len.ctypes.data, out.ctypes.data, cap, out_len.ctypes.data, 0, nb); td.append(time.perf_counter() - t); assert rc == 0
        assert np.array_equal(back, x)
        res.append((min(tc) * 1e3, min(td) * 1e3))
    os.environ["LZS_PIPE_MIN_MB"] = "4096"; os.environ["LZS_BATCH_SEG_MB"] = "1024"
    td = []
    for _ in range(5):
        t = time.perf_counter(); rc = L.lzs_decompress_batch(back.ctypes.data, 65536, 65536, back_len.ctypes.data, out.ctypes.data, cap, out_len.ctypes.data, 0, nb); td.append(time.perf_counter() - t); assert rc == 0
    assert np.array_equal(back, x)
    os.environ.pop("LZS_BATCH_SEG_MB")
    seg_ms = min(td) * 1e3
    print(f"{nb:5} blocks: one after the other  compress {res[0][0]:6.2f} ms  decompress {res[0][1]:6.2f} ms   |   overlapped  compress {res[1][0]:6.2f} ms  decompress {res[1][1]:6.2f} ms   |   by segments, one after the other: decompress {seg_ms:6.2f} ms", flush=True)
