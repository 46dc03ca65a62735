"""Throughput of the incremental interface by piece size (dev aid)."""
import ctypes, time, sys, os
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import lzs_compression_amd as lzs
from lzs_compression_amd import api, workload

L = lzs.lib()
total = 256 << 20
data = workload.fill(workload.CLASS_NAMES.index("text"), total // 65536, 65536, first_block=0, seed=workload.DEFAULT_SEED).tobytes()
src = ctypes.create_string_buffer(data, len(data))
dst = ctypes.create_string_buffer(lzs.compressed_max(len(data)) + 64)
back = ctypes.create_string_buffer(len(data) + 64)

def comp(piece, limit):
    p = api.CompressParameters(); L.lzs_compress_init_full(ctypes.addressof(p))
    n = min(limit, len(data)); pos = 0; out = 0
    t = time.perf_counter()
    p.outPtr = ctypes.addressof(dst); p.outLength = len(dst)
    while pos < n:
        k = min(piece, n - pos)
        p.inPtr = ctypes.addressof(src) + pos; p.inLength = k
        out += L.lzs_compress_incremental(ctypes.addressof(p), False); assert p.inLength == 0 and not (p.status & 0x10)
        pos += k
    p.inLength = 0
    out += L.lzs_compress_incremental(ctypes.addressof(p), True); assert p.status & 4
    return n, out, time.perf_counter() - t

def decomp(clen, piece, nplain):
    p = api.DecompressParameters(); L.lzs_decompress_init(ctypes.addressof(p))
    pos = 0; out = 0
    t = time.perf_counter()
    p.outPtr = ctypes.addressof(back); p.outLength = len(back)
    while pos < clen:
        k = min(piece, clen - pos)
        p.inPtr = ctypes.addressof(dst) + pos; p.inLength = k
        while p.inLength:
            out += L.lzs_decompress_incremental(ctypes.addressof(p)); assert not (p.status & 0x10)
        pos += k
    dt = time.perf_counter() - t
    assert out == nplain and back.raw[:nplain] == data[:nplain]
    return dt

comp(1 << 20, 8 << 20)
for piece, limit in ((512, 4 << 20), (65536, 64 << 20), (1 << 20, 256 << 20), (16 << 20, 256 << 20), (256 << 20, 256 << 20)):
    n, out, dt = comp(piece, limit)
    dd = decomp(out, max(piece // 2, 256), n)
    print(f"piece {piece:>10}: compress {n/dt/1e6:9.1f} MB/s ({dt/ (n/piece)*1e6:8.1f} us/call)   decompress {n/dd/1e6:9.1f} MB/s of output")
