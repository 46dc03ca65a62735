"""Dev aid: randomized cross-check of every host-visible path against the oracle for a given number
of seconds (argv[1], default 300).  Prints the seed of any failure."""
import os
os.environ.setdefault("LZS_DEV_ENV", "1")      # this script flips the library's switches between calls (csrc/lzs_internal.h: lzs_env)
import os, sys, time, random
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import oracle
import lzs_compression_amd as lzs
from lzs_compression_amd import api, workload

O = oracle.oracle()
TRACE = os.environ.get("FUZZ_TRACE")


def stage(seed, name, **kw):
    if TRACE:
        print("stage", seed, name, kw, {k: os.environ.get(k) for k in ("LZS_STREAM_SEG", "LZS_DEC_SEG", "LZS_FORCE_STREAM")}, flush=True)

text = bytes(workload.fill("text", 16).reshape(-1))
low = bytes(workload.fill("lowent", 16).reshape(-1))


def make(rng, maxlen):
    out = bytearray()
    target = rng.randint(0, maxlen)
    while len(out) < target:
        k = rng.randint(0, 6)
        if k == 0:
            out += bytes([rng.randint(0, 255)]) * rng.randint(1, 40000)
        elif k == 1:
            a = rng.randint(0, len(text) - 2); out += text[a:a + rng.randint(1, 60000)]
        elif k == 2:
            out += rng.randbytes(rng.randint(1, 20000))
        elif k == 3:
            unit = rng.randbytes(rng.randint(1, 3000)); out += unit * rng.randint(1, 40)
        elif k == 4:
            a = rng.randint(0, len(low) - 2); out += low[a:a + rng.randint(1, 100000)]
        elif k == 5 and len(out) > 10:
            a = rng.randint(0, len(out) - 1); out += out[a:a + rng.randint(1, 5000)]      # far or near repeat
        else:
            out += bytes(rng.choice(b"ab") for _ in range(rng.randint(1, 300)))
    return bytes(out[:target])


def pieces(rng, lo, hi):
    while True:
        yield rng.randint(lo, hi)


def inc_encode(rng, data):
    c = lzs.IncrementalCompressor()
    lo, hi = rng.choice(((1, 50), (100, 5000), (5000, 300000)))
    out, pos, pending, fin, status = bytearray(), 0, b"", False, 0
    for _ in range(10 ** 7):
        if not pending and not fin and pos < len(data):
            pending = data[pos:pos + rng.randint(lo, hi)]; pos += len(pending)
        if not pending and pos >= len(data) and (status & 1 or not data):
            fin = True
        got, used, status = c.step(pending, rng.randint(max(lo, 3), hi), fin)
        out += got; pending = pending[used:]
        if status & 4:
            return bytes(out)
    raise RuntimeError("no progress")


def inc_decode(rng, stream):
    d = lzs.IncrementalDecompressor()
    lo, hi = rng.choice(((1, 50), (100, 5000), (5000, 300000)))
    out, pos, pending = bytearray(), 0, b""
    for _ in range(10 ** 7):
        if not pending and pos < len(stream):
            pending = stream[pos:pos + rng.randint(lo, hi)]; pos += len(pending)
        got, used, status = d.step(pending, rng.randint(lo, 3 * hi))
        out += got; pending = pending[used:]
        if status & 4 or (not pending and pos >= len(stream) and status & 1):
            return bytes(out)
    raise RuntimeError("no progress")


# the REAL reference's incremental interface (oracle/_ref travels to the GPU box): packets finished
# with an end marker on ONE parameter block, so that later packets refer back into earlier ones
import ctypes, struct
_REF = os.path.join(os.path.dirname(__file__), "..", "..", "oracle", "_ref", "liblzs_ref.so")
REF = ctypes.CDLL(_REF) if os.path.exists(_REF) else None
if REF:
    REF.lzs_compress_incremental.restype = ctypes.c_size_t
    REF.lzs_compress_incremental.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    REF.lzs_compress_init_full.argtypes = [ctypes.c_void_p]


def ref_packets(packets):
    raw = ctypes.create_string_buffer(14432)
    REF.lzs_compress_init_full(ctypes.addressof(raw))
    out = bytearray()
    for data in packets:
        src = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
        dst = ctypes.create_string_buffer(len(data) + len(data) // 8 + 64)
        struct.pack_into("<QQQQ", raw, 0, ctypes.addressof(src), ctypes.addressof(dst), len(data), len(dst))
        n, fin = 0, False
        for _ in range(1000):
            n += REF.lzs_compress_incremental(ctypes.addressof(raw), fin)
            if raw.raw[32] & 4:
                break
            fin = struct.unpack_from("<Q", raw, 16)[0] == 0
        else:
            raise RuntimeError("the reference did not finish the packet")
        out += dst.raw[:n]
    return bytes(out)


def our_packets(rng, packets):
    c = lzs.IncrementalCompressor()
    out = bytearray()
    for data in packets:
        lo, hi = rng.choice(((1, 50), (100, 5000), (5000, 300000)))
        pos, pending, fin, status = 0, b"", False, 0
        for _ in range(10 ** 7):
            if not pending and not fin and pos < len(data):
                pending = data[pos:pos + rng.randint(lo, hi)]; pos += len(pending)
            if not pending and pos >= len(data) and (status & 1 or not data):
                fin = True
            got, used, status = c.step(pending, rng.randint(max(lo, 3), hi), fin)
            out += got; pending = pending[used:]
            if status & 4:
                break
        else:
            raise RuntimeError("no progress")
    return bytes(out)


def check_seed(seed):
    """Every host-visible path on the inputs drawn from `seed`, against the oracle (and the compiled
    reference where it is there); raises on the first difference and leaves the inputs in
    gpurun_out/fuzz_fail_<seed>.pkl."""
    rng = random.Random(seed)
    try:
        d = make(rng, rng.choice((3000, 60000, 400000, 1500000)))
        want = O.compress(d)
        for k in ("LZS_STREAM_SEG", "LZS_DEC_SEG", "LZS_FORCE_STREAM", "LZS_ROUTE"):
            os.environ.pop(k, None)
        # which route the small calls take (round 5: csrc/lzs_hostcodec.c): the seeds below 2 000 000 -- the ones that ever
        # failed are among them -- stay on the device; of the fresh ones half do, a quarter go by size like a caller's,
        # a quarter are served by the host route whatever their size (FUZZ_ROUTE=device|host|auto overrides)
        route = os.environ.get("FUZZ_ROUTE") or ("device" if seed < 2_000_000 else ("device", "auto", "host", "device")[seed & 3])
        if route != "auto":
            os.environ["LZS_ROUTE"] = route
        if rng.random() < 0.6:
            os.environ["LZS_STREAM_SEG"] = str(rng.choice((4096, 8192, 16384, 65536)))
        if rng.random() < 0.6:
            os.environ["LZS_DEC_SEG"] = str(rng.choice((256, 512, 1024, 2048, 8192)))
        if rng.random() < 0.3:
            os.environ["LZS_FORCE_STREAM"] = "1"
        stage(seed, "compress", n=len(d))
        assert lzs.compress(d) == want, "compress"
        cap = rng.choice((len(d) + 7, len(d), max(len(d) - 1, 0), len(d) // 2, 1))
        if cap:
            stage(seed, "decompress", n=len(want), cap=cap)
            assert lzs.decompress(want, cap) == d[:cap], "decompress"
            cut = rng.randint(0, len(want))
            stage(seed, "decompress cut", cut=cut)
            assert lzs.decompress(want[:cut], len(d) + 1) == O.decompress(want[:cut], len(d) + 1), "decompress cut"
        junk = rng.randbytes(rng.randint(1, 30000))
        stage(seed, "junk", n=len(junk))
        assert lzs.decompress(junk, 500000) == O.decompress(junk, 500000), "decompress junk"
        small = d[:rng.randint(0, min(len(d), 300000))]
        stage(seed, "inc encode", n=len(small))
        assert inc_encode(rng, small) == O.compress(small), "incremental encode"
        tiny = small[:rng.randint(0, 2500)]
        stage(seed, "simple encode, little room", n=len(tiny))
        c = lzs.IncrementalCompressor(simple=True)
        out_t, pos_t, pend_t, fin_t, st_t = bytearray(), 0, b"", False, 0
        for _ in range(10 ** 6):
            if not pend_t and not fin_t and pos_t < len(tiny):
                pend_t = tiny[pos_t:pos_t + rng.randint(1, 400)]; pos_t += len(pend_t)
            if not pend_t and pos_t >= len(tiny):
                fin_t = True
            got, used, st_t = c.step(pend_t, rng.randint(1, 14), fin_t)
            out_t += got; pend_t = pend_t[used:]
            if st_t & 4:
                break
        assert bytes(out_t) == O.compress(tiny), "lzs_simple_compress_incremental with 1..14 bytes of room"
        stage(seed, "inc decode", n=len(want))
        assert inc_decode(rng, want) == d, "incremental decode"
        ccap = rng.randint(0, len(want) + 3)
        stage(seed, "compress cap", ccap=ccap)
        assert lzs.compress(d, ccap) == want[:ccap], "compress with a cut capacity"
        parts = [make(rng, 50000) for _ in range(rng.randint(1, 6))]
        cat = b"".join(O.compress(x) for x in parts)
        stage(seed, "concat", n=len(cat))
        assert lzs.decompress_concat(cat, sum(map(len, parts)) + 5) == b"".join(parts), "decompress_concat"
        if len(d) >= 1:
            import torch
            sk_in, sk_out = rng.randint(0, 7), 4 * rng.randint(0, 3)          # any input alignment; output 4-aligned
            x = torch.empty(len(d) + 8, dtype=torch.uint8, device="cuda")
            x[sk_in:sk_in + len(d)] = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
            buf = torch.empty(lzs.compressed_max(len(d)) + 1024 + 16, dtype=torch.uint8, device="cuda")
            stage(seed, "device stream", n=len(d), sk_in=sk_in, sk_out=sk_out)
            _, nb = lzs.compress_stream(x[sk_in:sk_in + len(d)], buf[sk_out:])
            assert bytes(buf[sk_out:sk_out + nb].cpu().numpy()) == want, "compress_stream_device"
            y = torch.empty(len(want) + 8, dtype=torch.uint8, device="cuda")
            y[sk_in:sk_in + len(want)] = torch.frombuffer(bytearray(want), dtype=torch.uint8).cuda()
            dcap = rng.choice((len(d) + 5, len(d), max(1, len(d) // 3)))
            ob = torch.empty(dcap + 8, dtype=torch.uint8, device="cuda")
            view = ob[rng.randint(0, 7):][:dcap]
            _, got_n = lzs.decompress_stream(y[sk_in:sk_in + len(want)], dcap, view)
            assert got_n == min(dcap, len(d)) and bytes(view[:got_n].cpu().numpy()) == d[:got_n], "decompress_stream_device"
        if REF:
            packets = [make(rng, rng.choice((300, 5000, 40000))) for _ in range(rng.randint(1, 5))]
            if rng.random() < 0.5 and packets[0]:
                packets.append(packets[0][:len(packets[0]) // 2] + packets[-1][:777])     # refers far back
            stage(seed, "packets", n=[len(x) for x in packets])
            wantp = ref_packets(packets)
            assert our_packets(rng, packets) == wantp, "packets with history kept over end markers (vs the reference)"
            d2 = lzs.IncrementalDecompressor(); back = bytearray(); pend = wantp
            for _ in range(10 ** 6):
                got, used, status = d2.step(pend[:rng.randint(1, 70000)], rng.randint(1, 100000))
                back += got; pend = pend[used:]
                if not pend and status & 1:
                    break
            assert bytes(back) == b"".join(packets), "packets decoded back"
        stage(seed, "batch")
        nb = rng.randint(1, 40)
        blocks = [make(rng, 70000) for _ in range(nb)]
        stride = max(1, max(len(b) for b in blocks))
        arr = np.zeros((nb, stride), dtype=np.uint8); lens = np.zeros(nb, dtype=np.uint32)
        for i, b in enumerate(blocks):
            arr[i, :len(b)] = np.frombuffer(b, dtype=np.uint8); lens[i] = len(b)
        out, n = lzs.compress_batch(arr, lens)
        for i, b in enumerate(blocks):
            assert out[i, :n[i]].tobytes() == O.compress(b), "compress_batch"
        back, m = lzs.decompress_batch(out, n, stride)
        for i, b in enumerate(blocks):
            assert back[i, :m[i]].tobytes() == b, "decompress_batch"
        # streams cut anywhere, in rows whose tails are not zeros (the decoders do not mask a stream's last word):
        # the host-buffer call (segments for so small a batch) and the eight-per-wavefront block decoder itself
        stage(seed, "cut streams in poisoned rows")
        import torch
        srcs = [want] + [out[i, :n[i]].tobytes() for i in range(min(nb, 6))]
        cuts_ = []
        for _ in range(rng.randint(1, 48)):
            c = srcs[rng.randrange(len(srcs))]
            cuts_.append(c[:rng.randint(0, min(len(c), 9000))])
        stride = max(1, max(len(c) for c in cuts_)) + rng.randint(0, 40)
        rows = np.frombuffer(rng.randbytes(len(cuts_) * stride), dtype=np.uint8).reshape(len(cuts_), stride).copy()
        if rng.random() < 0.3:
            rows[:] = 0xFF
        lens = np.zeros(len(cuts_), dtype=np.uint32)
        for i, c in enumerate(cuts_):
            rows[i, :len(c)] = np.frombuffer(c, dtype=np.uint8); lens[i] = len(c)
        pcap = rng.choice((1, 300, 20000))
        wants_ = [O.decompress(c, pcap) for c in cuts_]
        back, m = lzs.decompress_batch(rows, lens, pcap)
        for i, w in enumerate(wants_):
            assert back[i, :m[i]].tobytes() == w, "decompress_batch of cut streams in poisoned rows"
        bk, bm = lzs.decompress_blocks(torch.from_numpy(rows).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(), pcap)
        torch.cuda.synchronize()
        bk, bm = bk.cpu().numpy(), bm.cpu().numpy()
        for i, w in enumerate(wants_):
            assert bk[i, :bm[i]].tobytes() == w, "decompress_blocks of cut streams in poisoned rows"
        if seed % 40 == 7:
            # a host batch large enough for the overlapped route (lzs_pipeline.c): ragged blocks in a strided array, every
            # byte against the one-after-the-other route, a sample against the oracle, and back
            nb = rng.randint(700, 1600)
            stage(seed, "large host batch", nb=nb)
            kinds = [make(rng, 70000) for _ in range(24)]
            stride = 70000
            arr = np.zeros((nb, stride), dtype=np.uint8); lens = np.zeros(nb, dtype=np.uint32)
            for i in range(nb):
                b = kinds[rng.randrange(len(kinds))]
                b = b[:rng.randint(0, len(b))] if rng.random() < 0.5 else b
                arr[i, :len(b)] = np.frombuffer(b, dtype=np.uint8); lens[i] = len(b)
            os.environ.pop("LZS_HOST_SERIAL", None)
            out, n = lzs.compress_batch(arr, lens)
            os.environ["LZS_HOST_SERIAL"] = "1"
            out2, n2 = lzs.compress_batch(arr, lens)
            os.environ.pop("LZS_HOST_SERIAL", None)
            assert np.array_equal(n, n2) and all(np.array_equal(out[i, :n[i]], out2[i, :n[i]]) for i in range(nb)), "large compress_batch: the two routes differ"
            for i in [0, nb - 1] + [rng.randrange(nb) for _ in range(10)]:
                assert out[i, :n[i]].tobytes() == O.compress(arr[i, :lens[i]].tobytes()), "large compress_batch vs oracle"
            back, m = lzs.decompress_batch(out, n, stride)
            assert np.array_equal(m, lens) and all(np.array_equal(back[i, :m[i]], arr[i, :lens[i]]) for i in range(nb)), "large decompress_batch"
    except Exception as e:
        import pickle
        os.makedirs(os.path.join(os.path.dirname(__file__), "..", "..", "gpurun_out"), exist_ok=True)
        keep = {k: v for k, v in locals().items() if k in ("d", "want", "cap", "cut", "junk", "small", "blocks", "seed")}
        keep["env"] = {k: os.environ.get(k) for k in ("LZS_STREAM_SEG", "LZS_DEC_SEG", "LZS_FORCE_STREAM")}
        pickle.dump(keep, open(os.path.join(os.path.dirname(__file__), "..", "..", "gpurun_out", "fuzz_fail_%d.pkl" % seed), "wb"))
        print("FAIL seed", seed, repr(e)[:300], {k: os.environ.get(k) for k in ("LZS_STREAM_SEG", "LZS_DEC_SEG", "LZS_FORCE_STREAM")}, flush=True)
        raise


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, it = time.time(), 0
    ONLY = os.environ.get("FUZZ_ONLY")
    while time.time() - t0 < budget:
        seed = int(ONLY) if ONLY else seed0 * 100000 + it
        if ONLY and it:
            break
        it += 1
        check_seed(seed)
    print(f"fuzz ok: {it} iterations in {time.time() - t0:.0f} s (seeds {seed0 * 100000}..{seed0 * 100000 + it - 1})", flush=True)
