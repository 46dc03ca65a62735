"""Dev aid: lzs_decompress_stream_device() of 1 GiB by segment size (LZS_DEC_SEG), beyond the
built-in 8 KiB maximum: needs a library built with -DLZS_DEC_SEG_MAX=65536 (LZS_LIBRARY=...)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np, torch
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
classes = sys.argv[1:] or ["text"]
for cls in classes:
    nblk = 16384
    x = torch.from_numpy(workload.fill(cls, nblk).reshape(-1)).cuda()
    buf, nbytes = lzs.compress_stream(x)
    comp = buf[:nbytes].clone()
    back = None
    for seg in [int(v) for v in os.environ.get("SEGS", "2048,3072,4096,6144,8192").split(",")]:
        os.environ["LZS_DEC_SEG"] = str(seg)
        best = 1e9
        for it in range(4):
            if it == 3: os.environ["LZS_STREAM_DEBUG"] = "1"
            t = time.perf_counter(); back, got = lzs.decompress_stream(comp, x.numel() + 16, back); dt = time.perf_counter() - t
            os.environ.pop("LZS_STREAM_DEBUG", None)
            best = min(best, dt)
        ok = got == x.numel() and bool(torch.equal(back[:got], x))
        print(f"== {cls} seg {seg}: {best*1e3:.2f} ms = {x.numel()/best/1e9:.2f} GB/s  {ok}", flush=True)
