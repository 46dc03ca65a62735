"""The small calls' two routes (csrc/lzs_hostcodec.c).

The one-shot calls below a measured size and the incremental calls on small pieces are served by the calling thread
(one host core beats a launch and its wait there); everything else, and every batch / device-pointer call, by the GPU.
LZS_ROUTE=host|device forces either.  Here:
  * the host route against everything the reference's tests pin for the 4-argument calls (golden vector, size laws,
    the fixtures minted from the compiled reference, the live reference where it travelled) -- no device needed, so
    this part runs in the CPU suite as well as on the GPU box;
  * the route is a matter of SIZE, not of failure: without a device the default route still fails loudly;
  * (-m gpu) both routes give the same bytes for sizes on either side of every crossover, the default included.
The incremental suite (tests/test_gpu_incremental.py) and the reference's own unit test (tests/test_gpu_dropin.py) run
on both routes themselves."""
import ctypes
import os
import threading

import numpy as np
import pytest

import oracle
from conftest import golden_bytes, length_bits, uncompressible_sequence
import lzs_compression_amd as lzs
from lzs_compression_amd import workload

O = oracle.oracle()
R = oracle.ref()


@pytest.fixture
def host(monkeypatch):
    monkeypatch.setenv("LZS_ROUTE", "host")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# ------------------------------------------------------------------ the host route against the reference's fixtures
def test_golden_vector_one_shot(host):
    """c/src/test/test-lzs-decompression.c:34-96 through lzs_compress()/lzs_decompress()."""
    comp, plain = golden_bytes("kat_compressed_1.bin"), golden_bytes("kat_decompressed_1.bin")
    assert lzs.compress(plain) == comp
    assert lzs.decompress(comp, len(plain) + 520) == plain


@pytest.mark.parametrize("data,hexout", [
    (b"", "c000"), (b"a", "30e000"), (b"aa", "30987000"), (b"aaa", "30e04c00"),
    (b"a" * 9, "30e07c3000"), (b"a" * 10, "30e07c7000"), (b"a" * 24, "30e07fc300"),
    (b"a" * 25, "30e07fc700"), (b"abcXabcYabc", "30988c658c2259c23800"),
])
def test_tiny_vectors_one_shot(host, data, hexout):
    assert lzs.compress(data).hex() == hexout
    assert lzs.decompress(bytes.fromhex(hexout), 100) == data


def test_config0_4k_roundtrip(host):
    """BASELINE.json configs[0]: single 4 KiB buffer, lzs_compress + lzs_decompress."""
    plain, comp = golden_bytes("text_4k.bin"), golden_bytes("text_4k.lzs")
    got = lzs.compress(plain, lzs.compressed_max(4096))
    assert got == comp
    assert lzs.decompress(got, 4096) == plain


def test_uncompressible_size_law(host):
    """c/src/test/test-lzs.c:93-119: every prefix of the sequence with no repeated digram."""
    seq = uncompressible_sequence()
    for n in range(len(seq) + 1):
        c = lzs.compress(seq[:n])
        assert len(c) == (n * 9 + 9 + 7) // 8, n
        assert lzs.decompress(c, 1000) == seq[:n]


def test_repeated_byte_size_law(host):
    """c/src/test/test-lzs.c:121-167: 0..1000 bytes of 'X'."""
    for n in range(1001):
        c = lzs.compress(b"X" * n)
        bits = {0: 0, 1: 9, 2: 18}.get(n)
        if bits is None:
            bits = 9 + 2 + 7 + length_bits(n - 1)
        assert len(c) == (bits + 9 + 7) // 8, n
        assert lzs.decompress(c, 1000) == b"X" * n


def test_edge_vectors(host, edge_vectors):
    """tests/golden/edge_vectors.json (minted from the compiled reference): window edges 2047/2048, the 12-cap where the
    nearer match wins, long runs, cut capacities; truncated and padded streams through the decoder."""
    for v in edge_vectors["compress"]:
        d, want = bytes.fromhex(v["in"]), bytes.fromhex(v["out"])
        assert lzs.compress(d) == want, v["name"]
        for cap, n in v["capped"].items():
            got = lzs.compress(d, int(cap))
            assert len(got) == n and got == want[:int(cap)], (v["name"], cap)
    for v in edge_vectors["decompress"]:
        stream = bytes.fromhex(v["in"])
        for cap, want in v["out"].items():
            assert lzs.decompress(stream, int(cap)).hex() == want, (v["name"], cap)


@pytest.mark.parametrize("cls", workload.CLASS_NAMES)
def test_class_blocks_against_the_reference_digests(host, class_digests, cls):
    """256 x 64 KiB blocks per class through the 4-argument call on the host route: lengths and SHA-256 equal the REAL
    reference's (tests/golden/class_digests.json, minted by tests/golden/make_golden.py from oracle/_ref); then back."""
    import hashlib
    want = class_digests["classes"][cls]
    nb, bl = class_digests["nblocks"], class_digests["block_len"]
    blocks = workload.fill(cls, nb, bl, seed=class_digests["seed"])
    assert hashlib.sha256(blocks.tobytes()).hexdigest() == want["input_sha256"]
    h = hashlib.sha256()
    for b in range(nb):
        d = blocks[b].tobytes()
        c = lzs.compress(d)
        assert len(c) == want["len"][b], (cls, b)
        h.update(c)
        if b % 16 == 0:
            assert lzs.decompress(c, bl + 7) == d, (cls, b)
    assert h.hexdigest() == want["sha256"]


def test_nothing_past_the_capacity_is_touched(host):
    data = workload.fill("text", 1)[0, :5000].tobytes()
    full = O.compress(data)
    L = lzs.lib()
    for cap in (0, 1, 2, 100, len(full) - 1):
        dst = (ctypes.c_ubyte * (cap + 64))(*([0xA5] * (cap + 64)))
        n = L.lzs_compress(ctypes.addressof(dst), cap, data, len(data))
        assert n == cap and bytes(dst[:cap]) == full[:cap] and all(b == 0xA5 for b in dst[cap:])
    for cap in (0, 1, 17, 4999):
        dst = (ctypes.c_ubyte * (cap + 64))(*([0x5A] * (cap + 64)))
        n = L.lzs_decompress(ctypes.addressof(dst), cap, full, len(full))
        assert n == cap and bytes(dst[:cap]) == data[:cap] and all(b == 0x5A for b in dst[cap:])


def test_differential_against_the_oracle_and_the_live_reference(host):
    """Random inputs of every kind the survey's fuzz used (alphabets of 1-256 symbols, runs, text, cut capacities),
    garbage and truncated streams through the decoder: the host route, the oracle and -- where oracle/_ref travelled
    -- the reference itself agree byte for byte."""
    rng = np.random.default_rng(77)
    text = workload.fill("text", 1)[0].tobytes()
    for trial in range(400):
        n = int(rng.integers(0, 6001))
        k = int(rng.choice([1, 2, 3, 4, 5, 17, 256]))
        kind = trial % 4
        if kind == 0:
            d = rng.integers(0, k, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            d = text[int(rng.integers(0, 50000)):][:n]
        elif kind == 2:
            unit = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            d = (unit * (n // len(unit) + 1))[:n]
        else:
            d = (bytes(int(rng.integers(0, 3000))) + text[:n])[:n]
        want = O.compress(d)
        assert lzs.compress(d) == want, (trial, n)
        if R is not None:
            assert R.compress(d) == want
        cap = int(rng.choice([0, 1, 2, 3, 7, 100, max(len(want) - 1, 0)]))
        assert lzs.compress(d, cap) == want[:cap]
        assert lzs.decompress(want, n + 3) == d
        cut = want[:int(rng.integers(0, len(want) + 1))]
        for ocap in (n + 3, int(rng.integers(0, n + 1))):
            assert lzs.decompress(cut, ocap) == O.decompress(cut, ocap), (trial, len(cut), ocap)
    for trial in range(400):
        g = rng.integers(0, 256, int(rng.integers(0, 700)), dtype=np.uint8).tobytes()
        if trial % 3 == 0:
            g = b"\xff" * int(rng.integers(0, 200)) + g
        for ocap in (0, 9, 1000, 40000):
            want = O.decompress(g, ocap)
            assert lzs.decompress(g, ocap) == want, (trial, ocap)
            if R is not None:
                assert R.decompress(g, ocap) == want


def test_many_threads_at_once(host):
    """The calls stay re-entrant like the reference's (lzs-compression.c:100-124: no globals): the host route's tables
    belong to the calling thread."""
    blocks = workload.fill("text", 8, 20000)
    want = [O.compress(blocks[i].tobytes()) for i in range(8)]
    bad = []

    def work(i):
        for _ in range(30):
            c = lzs.compress(blocks[i].tobytes())
            if c != want[i] or lzs.decompress(c, 20001) != blocks[i].tobytes():
                bad.append(i)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


# ------------------------------------------------------------------ a route, not a fallback
@pytest.mark.parametrize("route", ["", "device"])
def test_without_a_device_the_default_route_still_fails_loudly(monkeypatch, route):
    """The host route is taken BY SIZE on a box that has its device (require_device() comes first) or BY NAME
    (LZS_ROUTE=host); a box without a device gets an error from the small calls too, never a silent CPU result."""
    if _gpu_present():
        pytest.skip("this box has a device")
    if route:
        monkeypatch.setenv("LZS_ROUTE", route)
    else:
        monkeypatch.delenv("LZS_ROUTE", raising=False)
    L = lzs.lib()
    dst = (ctypes.c_ubyte * 64)()
    assert L.lzs_compress(ctypes.addressof(dst), 64, b"hello hello", 11) == 0
    assert "no HIP device" in lzs.last_error()
    assert L.lzs_decompress(ctypes.addressof(dst), 64, bytes.fromhex("30e000"), 3) == 0
    assert "no HIP device" in lzs.last_error()
    with pytest.raises(lzs.LzsError, match="no HIP device"):         # (the ERROR status flag, raised by the wrapper)
        lzs.IncrementalDecompressor().step(bytes.fromhex("30e000"), 10)


# ------------------------------------------------------------------ on the GPU box: the crossovers
@pytest.mark.gpu
def test_both_routes_and_the_default_agree_across_the_crossovers(monkeypatch):
    """Sizes on either side of every crossover (csrc/lzs_internal.h: HOST_COMPRESS_MAX, HOST_DECOMPRESS_MAX), each class:
    LZS_ROUTE=device, LZS_ROUTE=host and the default give the oracle's bytes."""
    sizes = (1, 100, 4095, 4096, 6144, 6145, 16383, 16384, 16385, 40000, 65536, 65537, 200000)
    for name in ("text", "lowent", "random"):
        blob = workload.fill(name, 4, 65536).tobytes()
        for n in sizes:
            d = blob[:n]
            want = O.compress(d)
            for route in ("device", "host", ""):
                if route:
                    monkeypatch.setenv("LZS_ROUTE", route)
                else:
                    monkeypatch.delenv("LZS_ROUTE", raising=False)
                assert lzs.compress(d) == want, (name, n, route)
                assert lzs.decompress(want, n + 1) == d, (name, n, route)
                assert lzs.decompress(want[: len(want) // 2], n) == O.decompress(want[: len(want) // 2], n), (name, n, route)


# ------------------------------------------------------------------ malformed streams through the incremental decoder
def _first_long_offset_zero(stream: bytes):
    """Bytes the stream has produced when its first `1 0 00000000000` token (long offset 0) begins, or None if it has none
    before the bits run out: a plain walk over the token grammar, end markers realigning to a byte
    as the incremental decoder does (lzs-decompression.c:564-576)."""
    bits = "".join(f"{b:08b}" for b in stream)
    at, made = 0, 0
    while True:
        if at + 1 > len(bits):
            return None
        if bits[at] == "0":
            if at + 9 > len(bits):
                return None
            at, made = at + 9, made + 1
            continue
        if at + 2 > len(bits):
            return None
        short = bits[at + 1] == "1"
        used = 9 if short else 13
        if at + used > len(bits):
            return None
        off = int(bits[at + 2:at + used], 2)
        if off == 0 and short:
            at = (at + used + 7) // 8 * 8
            continue
        if off == 0:
            return made
        at += used
        if at + 2 > len(bits):
            return None
        code = int(bits[at:at + 4].ljust(4, "0"), 2)
        width = 2 if code < 12 else 4
        if at + width > len(bits):
            return None
        length = 2 + (code >> 2) if code < 12 else code - 7
        at += width
        made += length
        while length in (8, 15) and (length == 8 or True):
            if at + 4 > len(bits):
                return None
            e = int(bits[at:at + 4], 2)
            at, made = at + 4, made + e
            if e != 15:
                break
            length = 15


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/liblzs_ref.so was not built (needs /root/reference)")
def test_incremental_decoder_on_malformed_streams_against_the_reference_itself(host):
    """Garbage through lzs_decompress_incremental() (host route: what a caller's small pieces get) beside the compiled
    reference's own incremental decoder on a zeroed parameter block.  Byte for byte the same -- except from the first token
    with a LONG OFFSET OF 0 on, which no compressor emits: the reference's incremental decoder reads a length there and
    copies from the oldest bytes of its ring, which lzs_decompress_init() never cleared (lzs-decompression.c:420-428,
    585-593: the caller's memory until 2047 bytes have been produced), while every decoder of this build applies the
    one-shot decoder's rule (:280: no length field, no copy) whatever the size of the piece.  INTEGRATION.md section 2
    says so; up to that token the outputs agree."""
    import random
    import struct
    ref = ctypes.CDLL(os.path.join(os.path.dirname(oracle.__file__), "_ref", "liblzs_ref.so"))
    ref.lzs_decompress_incremental.restype = ctypes.c_size_t
    ref.lzs_decompress_incremental.argtypes = [ctypes.c_void_p]
    ref.lzs_decompress_init.argtypes = [ctypes.c_void_p]

    def ref_decode(stream, room):
        raw = ctypes.create_string_buffer(2096)                 # LzsDecompressParameters_t, zeroed
        ref.lzs_decompress_init(ctypes.addressof(raw))
        src = ctypes.create_string_buffer(bytes(stream), max(len(stream), 1))
        dst = ctypes.create_string_buffer(room)
        struct.pack_into("<QQQQ", raw, 0, ctypes.addressof(src), ctypes.addressof(dst), len(stream), room)
        n = 0
        for _ in range(64):                                     # (a call returns at every end marker)
            n += ref.lzs_decompress_incremental(ctypes.addressof(raw))
            if struct.unpack_from("<Q", raw, 16)[0] == 0 and not raw.raw[32] & lzs.STATUS_END_MARKER:
                break
        return dst.raw[:n]

    def our_decode(stream, room, piece):
        d, out, pos = lzs.IncrementalDecompressor(), bytearray(), 0
        while pos < len(stream) and len(out) < room:
            pending = stream[pos:pos + piece]
            pos += len(pending)
            for _ in range(64):
                got, used, status = d.step(pending, room - len(out))
                out += got
                pending = pending[used:]
                if not pending or len(out) >= room:
                    break
        return bytes(out)

    rng = random.Random(20261004)
    valid = O.compress(bytes(workload.fill("text", 1, 3000).reshape(-1)))
    same = differing = 0
    for it in range(400):
        n = rng.randint(1, 360)
        kind = it % 4
        if kind == 0:
            stream = rng.randbytes(n)
        elif kind == 1:                                         # mostly matches: high bits set
            stream = bytes(rng.getrandbits(8) | 0x80 for _ in range(n))
        elif kind == 2:                                         # a valid stream with a few bytes hit
            s = bytearray(valid[:n + 40])
            for _ in range(3):
                s[rng.randrange(len(s))] ^= 1 << rng.randrange(8)
            stream = bytes(s)
        else:                                                   # some literals, `1 0` + eleven zero bits, then anything
            bits = "".join("0" + f"{rng.getrandbits(8):08b}" for _ in range(rng.randint(0, 40))) + "10" + "0" * 11
            bits += "".join(rng.choice("01") for _ in range(8 * n))
            bits = bits[:len(bits) // 8 * 8]
            stream = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        room = 20000
        want = ref_decode(stream, room)
        got = our_decode(stream, room, rng.choice((1, 7, 64, 4096)))
        cut = _first_long_offset_zero(stream)
        if cut is None or cut >= room:
            assert got == want, (it, stream.hex())
            same += 1
        else:
            assert got[:cut] == want[:cut], (it, cut, stream.hex())
            differing += got != want
    assert same >= 100 and differing >= 20          # (both kinds were met)
