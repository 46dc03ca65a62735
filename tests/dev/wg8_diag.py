"""Dev aid: the default kernel's other shapes in liblzs_variants.so (LZS_KERNEL=wg8|p256) against the oracle, block by block, on
ragged blocks of runs / periods / text / noise with cut capacities: prints what differs."""
import os, sys, random, ctypes
os.environ.setdefault("LZS_KERNEL", sys.argv[1] if len(sys.argv) > 1 else "wg8")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ["LZS_LIBRARY"] = os.path.join(os.path.dirname(__file__), "..", "..", "lzs_compression_amd", "liblzs_variants.so")
import numpy as np
import oracle
import lzs_compression_amd as lzs
from lzs_compression_amd import workload
O = oracle.oracle()
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
text = workload.fill('text', 8).tobytes()
NB = 40
rows = np.zeros((NB, 70000), dtype=np.uint8); lens = np.zeros(NB, dtype=np.uint32); kinds = []
for b in range(NB):
    d = bytearray(); ks = []
    want = rng.choice((0, 1, 2, 13, 100, 511, 512, 513, 1023, 1024, 1025, 4096, 20000, 65535, 65536, 70000))
    while len(d) < want:
        k = rng.randint(0, 4); ks.append(k)
        if k == 0: d += bytes([rng.randint(0, 255)]) * rng.randint(1, 9000)
        elif k == 1: a = rng.randint(0, len(text) - 2); d += text[a:a + rng.randint(1, 30000)]
        elif k == 2: d += rng.randbytes(rng.randint(1, 3000))
        elif k == 3: u = rng.randbytes(rng.randint(1, 2500)); d += u * rng.randint(1, 30)
        else: d += bytes(rng.choice(b'ab') for _ in range(rng.randint(1, 300)))
    d = bytes(d[:want]); rows[b, :len(d)] = np.frombuffer(d, dtype=np.uint8); lens[b] = len(d); kinds.append(ks)
bad = 0
for cap in (None, 3000, 7):
    out, n = lzs.compress_batch(rows, lens, cap)
    for b in range(NB):
        want = O.compress(rows[b, :lens[b]].tobytes(), lzs.compressed_max(70000) if cap is None else cap)
        got = out[b, :n[b]].tobytes()
        if got != want:
            bad += 1
            first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
            print(f"cap {cap} block {b}: {lens[b]} bytes in (segments {kinds[b][:12]}), {len(got)} out against {len(want)}, first difference at byte {first}")
print("differing:", bad)
