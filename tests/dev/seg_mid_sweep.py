import os, sys, time
sys.path.insert(0, "/root/repo")
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
data = bytes(workload.fill("text", 1024).reshape(-1))
lzs.compress(data[:1 << 20])
segs = (1024, 2048, 4096, 8192, 16384, 65536)
print("bytes      " + "  ".join(f"{str(s):>7}" for s in segs) + "   (ms, best of 4)")
for n in (512 << 10, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 64 << 20):
    row, want = [], None
    for seg in segs:
        os.environ["LZS_STREAM_SEG"] = str(seg); os.environ["LZS_FORCE_STREAM"] = "1"
        best = 1e9
        for _ in range(4):
            t = time.perf_counter(); out = lzs.compress(data[:n]); best = min(best, time.perf_counter() - t)
        want = want or out
        assert out == want, (n, seg)
        row.append(f"{best*1e3:7.3f}")
    print(f"{n:>9}  " + "  ".join(row), flush=True)
