"""Where the small calls' host route (csrc/lzs_hostcodec.c) stops paying: the one-shot calls and the incremental calls by
size on LZS_ROUTE=device and LZS_ROUTE=host, text class, buffers made beforehand, best of several calls (dev aid; run on
the GPU box: its output is profiles/r05/route_crossover.txt and the thresholds of csrc/lzs_internal.h come from it)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ.setdefault("LZS_DEV_ENV", "1")
import lzs_compression_amd as lzs
from lzs_compression_amd import api, workload

L = lzs.lib()
for f in (L.lzs_compress, L.lzs_decompress):
    f.restype = ctypes.c_size_t
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
cls = sys.argv[1] if len(sys.argv) > 1 else "text"
total = 64 << 20
data = workload.fill(cls, total // 65536, 65536).tobytes()
src = ctypes.create_string_buffer(data, len(data))
dst = ctypes.create_string_buffer(lzs.compressed_max(len(data)) + 64)
back = ctypes.create_string_buffer(len(data) + 64)


def best(fn, reps):
    t = 1e9
    for _ in range(reps):
        a = time.perf_counter(); fn(); t = min(t, time.perf_counter() - a)
    return t


print(f"class {cls}; one-shot calls (microseconds a call, best of several; MB/s of uncompressed bytes)")
print(f"{'bytes':>9} | {'compress device':>16} {'host':>10} | {'decompress device':>18} {'host':>10}")
for n in (256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20, 4 << 20):
    row = []
    clen = {}
    for route in ("device", "host"):
        os.environ["LZS_ROUTE"] = route
        reps = 30 if n <= 65536 else 8
        tc = best(lambda: clen.__setitem__(route, L.lzs_compress(dst, len(dst), src, n)), reps)
        td = best(lambda: L.lzs_decompress(back, n + 8, dst, clen[route]), reps)
        assert back.raw[:n] == data[:n]
        row += [tc, td]
    assert clen["device"] == clen["host"]
    print(f"{n:>9} | {row[0]*1e6:9.1f} us {n/row[0]/1e6:6.0f}  {row[2]*1e6:7.1f} us {n/row[2]/1e6:5.0f} | {row[1]*1e6:9.1f} us {n/row[1]/1e6:6.0f}  {row[3]*1e6:7.1f} us {n/row[3]/1e6:5.0f}"
          f"   (compressed {clen['host']})")


def inc_comp(piece, limit):
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


def inc_decomp(clen, piece, nplain):
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


print("\nincremental calls (MB/s of uncompressed bytes; the reference's tools read 512 bytes a call)")
print(f"{'piece':>9} | {'compress device':>16} {'host':>8} | {'decompress device':>18} {'host':>8}")
for piece, limit in ((64, 1 << 20), (512, 2 << 20), (4096, 8 << 20), (16384, 16 << 20), (65536, 32 << 20), (262144, 64 << 20), (1 << 20, 64 << 20)):
    row = []
    for route in ("device", "host"):
        os.environ["LZS_ROUTE"] = route
        lim = limit if route == "host" or piece >= 4096 else min(limit, 1 << 20)
        n, out, dt = inc_comp(piece, lim)
        dd = inc_decomp(out, piece, n)
        row += [n / dt / 1e6, n / dd / 1e6]
    print(f"{piece:>9} | {row[0]:13.1f}    {row[2]:8.1f} | {row[1]:15.1f}    {row[3]:8.1f}")
os.environ.pop("LZS_ROUTE")
n, out, dt = inc_comp(512, 2 << 20)
dd = inc_decomp(out, 512, n)
print(f"\ndefault route, 512-byte calls: compress {n/dt/1e6:.1f} MB/s, decompress {n/dd/1e6:.1f} MB/s of output")
t = best(lambda: L.lzs_compress(dst, len(dst), src, 4096), 50)
print(f"default route, lzs_compress of 4 KiB: {t*1e6:.1f} us")
