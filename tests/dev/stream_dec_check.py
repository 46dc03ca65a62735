#!/usr/bin/env python3
"""Development aid: the many-wavefront single-stream path of lzs_decompress() against the input."""
import os, sys, time
os.environ["LZS_STREAM_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import oracle
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
O = oracle.oracle()
ok = True
def check(name, data, cap=None):
    global ok
    comp = O.compress(data)
    cap = len(data) + 7 if cap is None else cap
    t = time.time(); got = lzs.decompress(comp, cap); dt = time.time() - t
    want = data[:cap]
    good = got == want
    ok &= good
    print(f"[{'ok' if good else 'FAIL'}] {name}: {len(comp)} -> {len(got)} (want {len(want)}), {dt*1e3:.1f} ms = {len(got)/dt/1e6:.1f} MB/s", flush=True)
    if not good:
        m = min(len(got), len(want))
        i = next((i for i in range(m) if got[i] != want[i]), m)
        print("   first diff at byte", i, "of", len(want))
for cls in workload.CLASS_NAMES:
    blocks = workload.fill(cls, 128)
    check(cls + " 8 MiB", bytes(blocks.reshape(-1)))
check("text 8 MiB cut at 3000001", bytes(workload.fill("text", 128).reshape(-1)), 3000001)
rng = np.random.default_rng(5)
mix = bytearray()
while len(mix) < 12_000_000:
    k = int(rng.integers(0, 4))
    if k == 0: mix += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 300000))
    elif k == 1: mix += bytes(workload.fill("text", 4).reshape(-1)[: int(rng.integers(1, 200000))])
    elif k == 2: mix += bytes(rng.integers(0, 256, int(rng.integers(1, 100000)), dtype=np.uint8))
    else:
        unit = bytes(rng.integers(0, 256, int(rng.integers(2, 2500)), dtype=np.uint8))
        mix += unit * int(rng.integers(1, 200))
check("mixed 12 MB", bytes(mix))
check("zeros 64 MiB", bytes(64 << 20))
print("ALL OK" if ok else "SOME FAILED")
