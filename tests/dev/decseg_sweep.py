"""Dev aid: one-shot lzs_decompress() latency by stream size and segment size (LZS_DEC_SEG)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
data = bytes(workload.fill("text", 512).reshape(-1))
lzs.compress(data[:1 << 20])
for n in (16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 32 << 20):
    plain = data[:n]
    comp = lzs.compress(plain)
    row = []
    for seg in ("one", 256, 512, 1024, 2048, 8192):
        if seg == "one":
            os.environ["LZS_ONE_WAVE"] = "1"
            if n > (1 << 20): row.append("      -"); os.environ.pop("LZS_ONE_WAVE"); continue
        else:
            os.environ.pop("LZS_ONE_WAVE", None)
            os.environ["LZS_DEC_SEG"] = str(seg)
            os.environ["LZS_FORCE_STREAM"] = "1"
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); out = lzs.decompress(comp, n + 8); best = min(best, time.perf_counter() - t)
        assert out == plain
        row.append(f"{best*1e3:7.2f}")
    os.environ.pop("LZS_FORCE_STREAM", None)
    print(f"{n:>9} B ({len(comp):>9} compressed): one-wave {row[0]}  seg256 {row[1]}  seg512 {row[2]}  seg1k {row[3]}  seg2k {row[4]}  seg8k {row[5]}  ms", flush=True)
