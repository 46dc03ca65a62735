"""Dev aid: one-shot lzs_compress() latency of SHORT inputs by segment size (LZS_STREAM_SEG, LZS_FORCE_STREAM):
where the threshold STREAM_MIN and the segment size of short streams come from."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
data = bytes(workload.fill("text", 64).reshape(-1))
lzs.compress(data[:1 << 20])
segs = ("one", 512, 1024, 2048, 4096)
print("bytes      " + "  ".join(f"{str(s):>7}" for s in segs) + "   (ms, best of 5)")
for n in (2 << 10, 4 << 10, 8 << 10, 16 << 10, 24 << 10, 32 << 10, 64 << 10, 128 << 10, 256 << 10, 1 << 20):
    row, want = [], None
    for seg in segs:
        os.environ.pop("LZS_ONE_WORKGROUP", None); os.environ.pop("LZS_STREAM_SEG", None); os.environ.pop("LZS_FORCE_STREAM", None)
        if seg == "one":
            os.environ["LZS_ONE_WORKGROUP"] = "1"
        else:
            os.environ["LZS_STREAM_SEG"] = str(seg); os.environ["LZS_FORCE_STREAM"] = "1"
        best = 1e9
        for _ in range(5):
            t = time.perf_counter(); out = lzs.compress(data[:n]); best = min(best, time.perf_counter() - t)
        want = want or out
        assert out == want, (n, seg)
        row.append(f"{best*1e3:7.3f}")
    print(f"{n:>9}  " + "  ".join(row), flush=True)
