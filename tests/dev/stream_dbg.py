import os, sys, time
os.environ["LZS_STREAM_DEBUG"]="1"
sys.path.insert(0, "/root/repo")
import numpy as np, lzs_compression_amd as lzs
from lzs_compression_amd import workload
for cls in ("text","random","lowent"):
    d = bytes(workload.fill(cls, 128).reshape(-1)[:200001])
    for rep in range(2):
        t=time.time(); out=lzs.compress(d); print(cls, rep, "%.2f ms"%((time.time()-t)*1e3), flush=True)
