"""Dev aid: lzs_compress_batch / lzs_decompress_batch of 1 GiB of text on host buffers by the pipeline's shape
(LZS_PIPE_GROUP chunks per launch, LZS_PIPE_CHUNK_MB, LZS_COPY_THREADS); best of 4 calls each."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ["LZS_DEV_ENV"] = "1"
import numpy as np
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
L = lzs.lib()
cap = lzs.compressed_max(65536)
nb = 16384
x = workload.fill("text", nb)
out = np.zeros((nb, cap), dtype=np.uint8); out_len = np.zeros(nb, dtype=np.uint32)
back = np.zeros((nb, 65536), dtype=np.uint8); back_len = np.zeros(nb, dtype=np.uint32)
def comp():
    return L.lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, nb)
def decomp():
    return L.lzs_decompress_batch(back.ctypes.data, 65536, 65536, back_len.ctypes.data, out.ctypes.data, cap, out_len.ctypes.data, cap, nb)
comp()
for which, fn, groups in (("compress", comp, (1, 2)), ("decompress", decomp, (2, 4, 6, 8))):
    for threads in (4,):
        for mb in (24, 46):
            for g in groups:
                os.environ["LZS_COPY_THREADS"], os.environ["LZS_PIPE_CHUNK_MB"], os.environ["LZS_PIPE_GROUP"] = str(threads), str(mb), str(g)
                best = 1e9
                for _ in range(4):
                    t = time.perf_counter(); rc = fn(); best = min(best, time.perf_counter() - t); assert rc == 0
                print(f"{which:10} threads {threads} chunk {mb:2} MiB x {g} a launch: {best*1e3:6.1f} ms  {x.size/best/1e9:5.1f} GB/s", flush=True)
assert (back_len == 65536).all() and np.array_equal(back, x)
