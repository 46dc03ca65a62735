"""Dev aid: PCIe-inclusive rate of the host-buffer batch calls through the C-ABI itself, with the
caller's buffers allocated and touched beforehand (tests/dev/hostbatch_time.py goes through the numpy
wrapper, which allocates and clears the output inside the timed region)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
os.environ.setdefault("LZS_DEV_ENV", "1")      # (so that LZS_HOST_SERIAL can be flipped between calls)
L = lzs.lib()
cap = lzs.compressed_max(65536)
for serial, nb in ((0, 1024), (0, 4096), (0, 16384), (1, 16384)):
    os.environ.pop("LZS_HOST_SERIAL", None)
    if serial:
        os.environ["LZS_HOST_SERIAL"] = "1"       # round 3's path: copy in, run, copy back, one after the other
    x = workload.fill("text", nb)
    out = np.zeros((nb, cap), dtype=np.uint8); out_len = np.zeros(nb, dtype=np.uint32)
    back = np.zeros((nb, 65536), dtype=np.uint8); back_len = np.zeros(nb, dtype=np.uint32)
    best = bd = 1e9
    for _ in range(4):
        t = time.perf_counter()
        rc = L.lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, nb)
        best = min(best, time.perf_counter() - t); assert rc == 0
    for _ in range(4):
        t = time.perf_counter()
        rc = L.lzs_decompress_batch(back.ctypes.data, 65536, 65536, back_len.ctypes.data, out.ctypes.data, cap, out_len.ctypes.data, cap, nb)
        bd = min(bd, time.perf_counter() - t); assert rc == 0
    assert (back_len == 65536).all() and np.array_equal(back, x)
    print(f"{'serial ' if serial else ''}{nb:>6} blocks: lzs_compress_batch {x.size/best/1e9:6.2f} GB/s ({best*1e3:7.1f} ms)   lzs_decompress_batch {x.size/bd/1e9:6.2f} GB/s ({bd*1e3:7.1f} ms)", flush=True)
os.environ.pop("LZS_HOST_SERIAL", None)
big = workload.fill("text", 8192).reshape(-1)
dst = np.zeros(lzs.compressed_max(big.size), dtype=np.uint8)
for _ in range(3):
    t = time.perf_counter(); n = L.lzs_compress(dst.ctypes.data, dst.size, big.ctypes.data, big.size); dt = time.perf_counter() - t
print(f"lzs_compress of {big.size >> 20} MiB (one stream): {big.size/dt/1e9:.2f} GB/s ({dt*1e3:.1f} ms)")
