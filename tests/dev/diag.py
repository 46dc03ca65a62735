#!/usr/bin/env python3
"""GPU-side diagnostic: compares liblzs against the oracle on a ladder of inputs and prints
the first divergence in detail; then times a small batch.  Development aid (uses oracle/)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import torch
import oracle
import lzs_compression_amd as lzs
from lzs_compression_amd import workload

O = oracle.oracle()
print(lzs.backend_info(), flush=True)


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n if len(a) != len(b) else -1


def check(name, data):
    want = O.compress(data)
    t = time.time()
    got = lzs.compress(data)
    dt = time.time() - t
    ok = got == want
    print(f"[{'ok' if ok else 'FAIL'}] compress {name}: in {len(data)} want {len(want)} got {len(got)} ({dt*1e3:.1f} ms)", flush=True)
    if not ok:
        i = first_diff(got, want)
        print("   first diff at byte", i, "got", got[max(0,i-4):i+8].hex(), "want", want[max(0,i-4):i+8].hex())
        tr = O.trace(data)
        # locate token whose bits contain byte i (approx by cumulative bit cost)
        bits = 0
        for (pos, off, ln) in tr:
            if off == 0: cost = 9
            else:
                cost = (9 if off <= 127 else 13) + (2 if ln <= 4 else 4)
                if ln >= 8: cost += ((ln - 8) // 15 + 1) * 4
            if (bits + cost) // 8 >= i:
                print(f"   around token pos={pos} off={off} len={ln} bitpos={bits}")
                break
            bits += cost
        return False
    back = lzs.decompress(want, len(data) + 5)
    if back != data:
        i = first_diff(back, data)
        print(f"[FAIL] decompress {name}: got {len(back)} bytes, first diff at {i}")
        return False
    return True


allok = True
cases = [("empty", b""), ("a", b"a"), ("aa", b"aa"), ("aaa", b"aaa"), ("a*25", b"a" * 25),
         ("abcXabcYabc", b"abcXabcYabc"), ("zeros300", bytes(300)), ("zeros5000", bytes(5000))]
rng = np.random.default_rng(0)
cases += [("rand100", bytes(rng.integers(0, 256, 100, dtype=np.uint8))),
          ("rand5000", bytes(rng.integers(0, 256, 5000, dtype=np.uint8))),
          ("abab", b"ab" * 3000),
          ("text600", workload.fill("text", 1)[0, :600].tobytes()),
          ("text4k", workload.fill("text", 1)[0, :4096].tobytes()),
          ("text64k", workload.fill("text", 1)[0].tobytes()),
          ("lowent64k", workload.fill("lowent", 1)[0].tobytes()),
          ("random64k", workload.fill("random", 1)[0].tobytes())]
for name, data in cases:
    allok &= check(name, data)

for cls in workload.CLASS_NAMES:
    nb = 2048
    blocks = workload.fill(cls, nb)
    x = torch.from_numpy(blocks).cuda()
    slots, lens = lzs.compress_blocks(x)
    torch.cuda.synchronize()
    t = time.time()
    slots, lens = lzs.compress_blocks(x, out=slots, out_len=lens)
    torch.cuda.synchronize()
    dt = time.time() - t
    t = time.time()
    back, back_len = lzs.decompress_blocks(slots, lens, 65536)
    torch.cuda.synchronize()
    dt2 = time.time() - t
    rt = torch.equal(back[:, :65536], x)
    want, want_len, _ = oracle.run_blocks(O, blocks[:128], threads=8)
    ok = (want_len == lens.cpu().numpy()[:128]).all()
    print(f"{cls}: {nb} blocks compress {nb*65536/dt/1e9:.2f} GB/s, decompress {nb*65536/dt2/1e9:.2f} GB/s, "
          f"ratio {lens.sum().item()/blocks.size:.4f}, roundtrip {rt}, lens==oracle {ok}", flush=True)
    allok &= bool(rt and ok)
print("ALL OK" if allok else "SOME FAILED")
sys.exit(0 if allok else 1)
