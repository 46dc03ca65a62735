#!/usr/bin/env python3
"""Development aid: the multi-workgroup single-stream path of lzs_compress() against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import oracle
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
O = oracle.oracle()
ok = True
def check(name, data):
    global ok
    t = time.time(); got = lzs.compress(data); dt = time.time() - t
    t = time.time(); want = O.compress(data); dt2 = time.time() - t
    good = got == want
    ok &= good
    print(f"[{'ok' if good else 'FAIL'}] {name}: {len(data)} -> {len(got)} (want {len(want)}), gpu {dt*1e3:.1f} ms = {len(data)/dt/1e6:.1f} MB/s, oracle {dt2*1e3:.0f} ms", flush=True)
    if not good:
        n = min(len(got), len(want))
        i = next((i for i in range(n) if got[i] != want[i]), n)
        print("   first diff at byte", i)
for cls in workload.CLASS_NAMES:
    blocks = workload.fill(cls, 128)           # 8 MiB
    check(cls + " 8 MiB", bytes(blocks.reshape(-1)))
    check(cls + " 200001 B", bytes(blocks.reshape(-1)[:200001]))
rng = np.random.default_rng(5)
mix = bytearray()
while len(mix) < 6_000_000:
    k = int(rng.integers(0, 4))
    if k == 0: mix += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 300000))
    elif k == 1: mix += bytes(workload.fill("text", 4).reshape(-1)[: int(rng.integers(1, 200000))])
    elif k == 2: mix += bytes(rng.integers(0, 256, int(rng.integers(1, 100000)), dtype=np.uint8))
    else:
        unit = bytes(rng.integers(0, 256, int(rng.integers(2, 2500)), dtype=np.uint8))
        mix += unit * int(rng.integers(1, 200))
check("mixed 6 MB", bytes(mix))
check("zeros 5 MiB", bytes(5 << 20))
print("ALL OK" if ok else "SOME FAILED")
