"""The incremental interface (lzs_compress_init / lzs_compress_incremental,
lzs_decompress_init / lzs_decompress_incremental; reference c/src/liblzs/lzs.h:90-232) on the
GPU, driven through the C-ABI with the reference's own calling patterns
(c/src/test/test-lzs-decompression.c:130-290, c/src/utils/lzs-compress.c:91-134,
lzs-decompress.c:82-121).  Whatever the chunking, the stream is the one-shot stream, bit for bit.
"""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

import oracle
from conftest import golden_bytes
import lzs_compression_amd as lzs
from lzs_compression_amd import api, workload

torch = pytest.importorskip("torch")
O = oracle.oracle()
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


# Every test here runs on BOTH routes of the small calls: "device" -- LZS_ROUTE=device, every call on
# the GPU whatever its size, the parity tests proper (-m gpu) -- and "host" -- LZS_ROUTE=host, the calling thread's own
# codec (csrc/lzs_hostcodec.c) for every size, which needs no device and therefore runs in the CPU suite too.  The default
# (by size) is a mix of the two that switches at the crossovers; tests/test_routes.py covers the switch itself.
@pytest.fixture(autouse=True, params=[pytest.param("device", marks=pytest.mark.gpu), "host", pytest.param("by-size", marks=pytest.mark.gpu)])
def route(request, monkeypatch):
    if request.param != "host" and not torch.cuda.is_available():
        pytest.fail("the device route needs a GPU (it has no fallback)")
    if request.param == "by-size":                    # what a caller gets: a stream changes route from call to call with the size of its pieces
        monkeypatch.delenv("LZS_ROUTE", raising=False)
    else:
        monkeypatch.setenv("LZS_ROUTE", request.param)
    return request.param


def device_only(route):
    """For a test of what only the device route does (segments, many wavefronts)."""
    if route != "device":
        pytest.skip("a property of the device route")


def _decode(stream, in_chunks, out_chunks, stop_at_markers=1):
    """Drive lzs_decompress_incremental like test-lzs-decompression.c: input and output handed
    over in pieces of the given sizes (iterators), until `stop_at_markers` end markers were seen
    and the input is used up."""
    d = lzs.IncrementalDecompressor()
    out, pos, pending, markers, calls = bytearray(), 0, b"", 0, 0
    while True:
        if not pending and pos < len(stream):
            k = next(in_chunks)
            pending = stream[pos:pos + k]
            pos += len(pending)
        got, used, status = d.step(pending, next(out_chunks))
        calls += 1
        out += got
        pending = pending[used:]
        if status & api.STATUS_END_MARKER:
            markers += 1
        if not pending and pos >= len(stream) and (status & api.STATUS_INPUT_STARVED):
            break
        assert calls < 200000, "no progress"
    return bytes(out), markers


def _const(k):
    while True:
        yield k


def _rand(rng, lo, hi):
    while True:
        yield rng.randint(lo, hi)


def _sample(kind, n, seed=5):
    if kind == "zeros":
        return bytes(n)
    blocks = workload.fill(workload.CLASS_NAMES.index(kind), 1, n, first_block=seed, seed=workload.DEFAULT_SEED)
    return np.asarray(blocks).tobytes()[:n]


# ------------------------------------------------------------------ decoding
@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "dropin-test-lzs-decompression")),
                    reason="oracle/_ref/dropin-test-lzs-decompression was not built (needs /root/reference)")
def test_reference_decompression_unit_test_passes_against_our_library():
    """c/src/test/test-lzs-decompression.c compiled against OUR header, linked with OUR library:
    the golden vector through lzs_decompress and the three incremental driving patterns
    (all at once :130-171, 10 input bytes a call :177-231, 10 output bytes a call :236-290)."""
    r = subprocess.run([os.path.join(REFDIR, "dropin-test-lzs-decompression")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 Tests 0 Failures 0 Ignored" in r.stdout, r.stdout


def test_golden_vector_decoded_in_pieces():
    comp, plain = golden_bytes("kat_compressed_1.bin"), golden_bytes("kat_decompressed_1.bin")
    for ins, outs in ((_const(len(comp)), _const(1000)), (_const(10), _const(1000)), (_const(1000), _const(10)),
                      (_const(1), _const(1)), (_const(3), _const(7))):
        got, markers = _decode(comp, ins, outs)
        assert got == plain and markers == 1


@pytest.mark.parametrize("kind", ["text", "lowent", "random", "zeros"])
def test_decode_random_pieces_vs_original(kind):
    rng = random.Random(11)
    plain = _sample(kind, 300000)
    comp = O.compress(plain)
    for lo, hi in ((1, 40), (100, 5000), (20000, 90000)):
        got, markers = _decode(comp, _rand(rng, lo, hi), _rand(rng, lo, 3 * hi))
        assert got == plain and markers == 1


def test_decode_continues_after_end_markers_with_history():
    """Streams one after another: the decoder stops at each end marker (status END_MARKER),
    realigns to the byte boundary and goes on (lzs-decompression.c:564-576); the second and third
    streams here were made by the REFERENCE's incremental compressor, which keeps its history over
    an end marker (RFC 1974), so they reach back into the streams before them."""
    packets = [golden_bytes("inc_packet_%d.bin" % i) for i in range(3)]
    stream = golden_bytes("inc_packets.lzs")
    rng = random.Random(3)
    got, markers = _decode(stream, _rand(rng, 1, 300), _rand(rng, 1, 300), stop_at_markers=3)
    assert got == b"".join(packets) and markers == 3


# ------------------------------------------------------------------ compression
def _encode(data, in_chunks, out_chunks, finish=True, comp=None):
    """Drive lzs_compress_incremental like c/src/utils/lzs-compress.c:91-134: input in pieces,
    `finish` raised once all of it was taken and the compressor says it is starved."""
    c = comp or lzs.IncrementalCompressor()
    out, pos, pending, fin, status, calls = bytearray(), 0, b"", False, 0, 0
    while True:
        if not pending and not fin and pos < len(data):
            pending = data[pos:pos + next(in_chunks)]
            pos += len(pending)
        if not pending and pos >= len(data) and (status & api.STATUS_INPUT_STARVED or not data):
            if not finish:
                break
            fin = True
        got, used, status = c.step(pending, next(out_chunks), fin)
        calls += 1
        out += got
        pending = pending[used:]
        if status & api.STATUS_END_MARKER:
            break
        if not finish and not pending and pos >= len(data) and not (status & api.STATUS_NO_OUTPUT_BUFFER_SPACE):
            break
        assert calls < 400000, "no progress"
    return bytes(out), c


def test_golden_vector_encoded_in_pieces():
    comp, plain = golden_bytes("kat_compressed_1.bin"), golden_bytes("kat_decompressed_1.bin")
    for ins, outs in ((_const(len(plain)), _const(1000)), (_const(10), _const(1000)), (_const(1000), _const(10)),
                      (_const(1), _const(3)), (_const(512), _const(512)), (_const(7), _const(5))):
        got, _ = _encode(plain, ins, outs)
        assert got == comp


def test_smoke_pattern_50_in_100_out():
    """__graft_entry__.smoke()'s pattern: 50 bytes offered and 100 bytes of room per call, driven
    like the reference's tool (utils/lzs-compress.c:91-134).  The compressor collects small pieces,
    so most of the stream appears at the flush and must be drained over several calls."""
    comp, plain = golden_bytes("kat_compressed_1.bin"), golden_bytes("kat_decompressed_1.bin")
    assert lzs.incremental_compress(plain, 50, 100) == comp
    assert lzs.incremental_compress(plain, 50, 3) == comp
    assert lzs.incremental_compress(b"", 50, 100) == bytes.fromhex("c000")
    # a single flush call into too little room reports NO_OUTPUT_BUFFER_SPACE, never END_MARKER
    c = lzs.IncrementalCompressor()
    out, used, status = c.step(plain, 100, True)
    assert used == len(plain) and len(out) == 100 and out == comp[:100]
    assert status & api.STATUS_NO_OUTPUT_BUFFER_SPACE and not status & api.STATUS_END_MARKER
    rest = b""
    while not status & api.STATUS_END_MARKER:
        got, _, status = c.step(b"", 100, True)
        rest += got
    assert out + rest == comp


def test_simple_compressor_block_gives_the_same_stream():
    """lzs_simple_compress / lzs_simple_compress_init / lzs_simple_compress_incremental
    (reference lzs.h:224-227, lzs-compression-simple.c): same output as the hashed compressor
    (SURVEY [probe]); here the same device code behind the 2112-byte block."""
    comp, plain = golden_bytes("kat_compressed_1.bin"), golden_bytes("kat_decompressed_1.bin")
    dst = ctypes.create_string_buffer(1024)
    n = lzs.lib().lzs_simple_compress(ctypes.addressof(dst), 1024, plain, len(plain))
    assert dst.raw[:n] == comp
    for ins, outs in ((len(plain), 1000), (50, 100), (10, 1000), (1000, 26), (1000, 13), (7, 13), (1, 64), (512, 512)):
        assert lzs.incremental_compress(plain, ins, outs, simple=True) == comp, (ins, outs)
    assert lzs.incremental_compress(b"", 10, 100, simple=True) == bytes.fromhex("c000")
    rng = random.Random(3)
    for kind in ("text", "lowent", "random", "zeros"):
        data = _sample(kind, 120000)
        want = O.compress(data)
        assert lzs.incremental_compress(data, 4096, 8192, simple=True) == want, kind
        assert lzs.incremental_compress(data, 100000, 200000, simple=True) == want, kind
        # random pieces; room never below what the block can promise for
        c, out, pos, pending, fin, status = lzs.IncrementalCompressor(simple=True), bytearray(), 0, b"", False, 0
        for _ in range(200000):
            if not pending and not fin and pos < len(data):
                pending = data[pos:pos + rng.randint(1, 3000)]
                pos += len(pending)
            if not pending and pos >= len(data):
                fin = True
            got, used, status = c.step(pending, rng.randint(13, 4000), fin)
            out += got
            pending = pending[used:]
            if status & api.STATUS_END_MARKER:
                break
        assert bytes(out) == want, kind
    # Any room at all makes progress (lzs-compression-simple.c:435-647 goes on with outLength >= 1; VERDICT r03:
    # below 13 bytes of room this block used to return NO_OUTPUT_BUFFER_SPACE without taking anything, and a
    # caller with a small buffer span): the reference's pattern with 1, 2 and 5 bytes of room per call ...
    for ins, outs in ((50, 1), (507, 1), (3, 2), (50, 5), (1, 1), (16, 3)):
        assert lzs.incremental_compress(plain, ins, outs, simple=True) == comp, (ins, outs)
    # ... on long matches that are still open when the room runs out, on literals, on text; and in random small pieces
    for kind in ("text", "lowent", "random", "zeros"):
        data = _sample(kind, 6000)
        want = O.compress(data)
        for ins, outs in ((6000, 1), (100, 2), (17, 5)):
            assert lzs.incremental_compress(data, ins, outs, simple=True) == want, (kind, ins, outs)
        c, out, pos, pending, fin, status = lzs.IncrementalCompressor(simple=True), bytearray(), 0, b"", False, 0
        for _ in range(200000):
            if not pending and not fin and pos < len(data):
                pending = data[pos:pos + rng.randint(1, 300)]
                pos += len(pending)
            if not pending and pos >= len(data):
                fin = True
            got, used, status = c.step(pending, rng.randint(1, 12), fin)
            assert len(got) <= 12
            out += got
            pending = pending[used:]
            if status & api.STATUS_END_MARKER:
                break
        assert bytes(out) == want, kind
    # every call with room either takes input or hands out bytes (no spinning), and says when room was the limit
    c = lzs.IncrementalCompressor(simple=True)
    pending, out, status, idle = plain, bytearray(), 0, 0
    while not status & api.STATUS_END_MARKER:
        got, used, status = c.step(pending, 2, not pending)
        idle = 0 if (got or used) else idle + 1
        assert idle < 2, (len(out), len(pending), status)
        assert len(got) <= 2
        out += got
        pending = pending[used:]
    assert bytes(out) == comp


@pytest.mark.parametrize("data,hexout", [
    (b"", "c000"), (b"a", "30e000"), (b"aa", "30987000"), (b"a" * 9, "30e07c3000"),
    (b"a" * 24, "30e07fc300"), (b"abcXabcYabc", "30988c658c2259c23800"),
])
def test_tiny_vectors_incremental(data, hexout):
    for k in (1, 2, 100):
        got, _ = _encode(data, _const(k), _const(100))
        assert got.hex() == hexout


@pytest.mark.parametrize("kind", ["text", "lowent", "random", "zeros", "mixed"])
def test_encode_random_pieces_equals_one_shot(kind):
    """Any chunking gives the one-shot stream (the reference's property; SURVEY.md §8f N3 probe)."""
    rng = random.Random(17)
    if kind == "mixed":
        t = _sample("text", 200000)
        plain = t[:50000] + bytes(70000) + t[50000:90000] + b"ab" * 40000 + _sample("random", 30000) + bytes(5)
    else:
        plain = _sample(kind, 300000)
    want = O.compress(plain)
    for lo, hi in ((1, 60), (100, 5000), (20000, 200000)):
        got, _ = _encode(plain, _rand(rng, lo, hi), _rand(rng, max(lo, 3), hi))
        assert got == want, (kind, lo, hi, len(got), len(want))


def test_encode_one_big_call_uses_many_segments(route):
    device_only(route)
    plain = _sample("text", 3 << 20)
    got, _ = _encode(plain, _const(len(plain)), _const(4 << 20))
    assert got == O.compress(plain)
    got, _ = _encode(plain, _const(700001), _const(1 << 20))
    assert got == O.compress(plain)


def test_packets_with_history_kept_over_end_markers_equal_the_reference():
    """RFC 1974 use: each packet is finished with an end marker and the SAME block goes on, so later
    packets refer back into earlier ones.  Expected bytes: the reference's own incremental compressor
    (tests/golden/make_incremental_golden.py)."""
    packets = [golden_bytes("inc_packet_%d.bin" % i) for i in range(3)]
    want = golden_bytes("inc_packets.lzs")
    for chunk, space in ((512, 512), (100000, 100000), (33, 7)):
        c, out = None, b""
        for p in packets:
            got, c = _encode(p, _const(chunk), _const(space), comp=c)
            out += got
        assert out == want, (chunk, space)


def test_incremental_round_trip_through_both_directions():
    rng = random.Random(23)
    plain = _sample("lowent", 500000)
    comp, _ = _encode(plain, _rand(rng, 1000, 50000), _rand(rng, 1000, 50000))
    back, markers = _decode(comp, _rand(rng, 1000, 50000), _rand(rng, 1000, 50000))
    assert back == plain and markers == 1


# ------------------------------------------------------------------ the reference's tools on this library
@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "dropin-lzs-compress")),
                    reason="oracle/_ref/dropin-lzs-* were not built (needs /root/reference)")
def test_reference_file_tools_built_on_this_library(tmp_path):
    """c/src/utils/lzs-compress.c and lzs-decompress.c (512-byte reads through the incremental
    interface) compiled against OUR header and linked with OUR library."""
    plain = _sample("text", 150000)
    src, comp, back = tmp_path / "in.bin", tmp_path / "in.lzs", tmp_path / "back.bin"
    src.write_bytes(plain)
    subprocess.run([os.path.join(REFDIR, "dropin-lzs-compress"), str(src), str(comp)], check=True, timeout=600)
    assert comp.read_bytes() == O.compress(plain)
    subprocess.run([os.path.join(REFDIR, "dropin-lzs-decompress"), str(comp), str(back)], check=True, timeout=600)
    assert back.read_bytes() == plain


def test_garbage_decodes_like_the_reference_incremental_decoder():
    """Random bytes: same output and the same number of end markers as the reference's
    lzs_decompress_incremental() (tests/golden/inc_garbage.json), whole and in small pieces."""
    import json
    vecs = json.load(open(os.path.join(ROOT, "tests", "golden", "inc_garbage.json")))
    assert len(vecs) >= 60
    rng = random.Random(5)
    for v in vecs:
        stream, want = bytes.fromhex(v["in"]), bytes.fromhex(v["out"])
        for ins, outs in ((_const(len(stream)), _const(4096)), (_rand(rng, 1, 9), _rand(rng, 1, 50))):
            got, markers = _decode(stream, ins, outs)
            assert got == want and markers == v["markers"], v["in"]


def test_large_pieces_decode_through_many_wavefronts(route):
    """Pieces of 16 KiB and more are decoded by many wavefronts as far as whole segments go
    (history, the bits left over from the call before and a running extension carried in), the
    rest by the one wavefront: several streams back to back (end markers in the middle of pieces),
    long runs, output space smaller and larger than what a piece produces."""
    device_only(route)
    rng = random.Random(41)
    t = _sample("text", 900000)
    plains = [t[:400000], bytes(300000) + t[400000:500000] + b"q" * 200000, _sample("random", 150000), t[500000:900000]]
    stream = b"".join(O.compress(x) for x in plains)
    want = b"".join(plains)
    for (ilo, ihi), (olo, ohi) in (((16384, 16384), (1 << 20, 1 << 20)), ((20000, 300000), (5000, 400000)),
                                   ((len(stream), len(stream)), (4096, 100000)), ((100000, 100000), (1 << 22, 1 << 22))):
        got, markers = _decode(stream, _rand(rng, ilo, ihi), _rand(rng, olo, ohi))
        assert got == want and markers == len(plains), (ilo, ihi, olo, ohi, len(got), markers)


def test_pieces_of_several_mib_and_the_pass_size_the_block_remembers(route):
    """Round 6: how much input one pass of the many-wavefront path takes is kept in the caller's block (1 MiB doubling while
    passes are used up to their end, halving when an end marker comes early), so a stream fed in pieces of several MiB is
    decoded in one pass a piece.  One stream of 40 MiB of text in pieces of 4 MiB and of 1.5 - 6 MiB (the size grows), then
    -- on the SAME block, after the marker -- sixty short streams back to back in one piece (markers early in every pass: the size
    shrinks again), then the long stream once more; every byte against the input."""
    device_only(route)
    rng = random.Random(77)
    big = _sample("text", 40 << 20)
    sbig = lzs.compress(big)
    shorts = [_sample("text", rng.randint(20000, 200000), seed=9 + i) for i in range(60)]
    sshort = b"".join(O.compress(x) for x in shorts)
    d = lzs.IncrementalDecompressor()

    def feed(stream, sizes, room):
        out, pos = bytearray(), 0
        while pos < len(stream):
            pending = stream[pos:pos + next(sizes)]
            pos += len(pending)
            for _ in range(100000):
                got, used, status = d.step(pending, room)
                out += got
                pending = pending[used:]
                assert not (status & lzs.STATUS_ERROR)
                if not pending:
                    break
        return bytes(out)

    assert feed(sbig, _const(4 << 20), 16 << 20) == big
    assert feed(sshort, _const(len(sshort)), 64 << 20) == b"".join(shorts)
    assert feed(sbig, _rand(rng, 3 << 19, 6 << 20), 8 << 20) == big


def test_no_large_piece_is_left_to_one_wavefront(route):
    """A copy still running when a call returns must not send the whole next piece to the one wavefront (round 4: half
    a MiB took 176 ms that way, one piece in fifty, against 0.7 ms for the others).  Sixty pieces of 256 KiB of one
    text stream, the slowest of them well under what one wavefront needs for such a piece (~90 ms)."""
    device_only(route)
    import time
    plain = _sample("text", 24 << 20)
    stream = O.compress(plain)
    d = lzs.IncrementalDecompressor()
    out, worst, pos, n_pieces = bytearray(), 0.0, 0, 0
    while pos < len(stream):
        pending = stream[pos:pos + (256 << 10)]
        pos += len(pending)
        t = time.perf_counter()
        while pending:
            got, used, status = d.step(pending, 1 << 20)
            out += got
            pending = pending[used:]
        dt = time.perf_counter() - t
        if n_pieces:                                   # (the first call allocates)
            worst = max(worst, dt)
        n_pieces += 1
    assert bytes(out) == plain
    assert worst < 0.040, f"a 256 KiB piece took {worst * 1e3:.1f} ms"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "liblzs_ref.so")),
                    reason="oracle/_ref/liblzs_ref.so was not built (needs /root/reference)")
def test_random_packets_against_the_reference_library_itself():
    """Where the compiled reference travelled along (oracle/_ref/liblzs_ref.so): random packets,
    each finished with an end marker on ONE parameter block (history kept, so later packets refer
    back), random piece sizes on our side -- byte for byte what the reference's own
    lzs_compress_incremental() writes, and our decoder reads it back."""
    import ctypes
    import struct
    ref = ctypes.CDLL(os.path.join(REFDIR, "liblzs_ref.so"))
    ref.lzs_compress_incremental.restype = ctypes.c_size_t
    ref.lzs_compress_incremental.argtypes = [ctypes.c_void_p, ctypes.c_bool]
    ref.lzs_compress_init_full.argtypes = [ctypes.c_void_p]

    def ref_packets(packets):
        raw = ctypes.create_string_buffer(14432)            # the reference's LzsCompressParameters_t, as bytes
        ref.lzs_compress_init_full(ctypes.addressof(raw))
        out = bytearray()
        for data in packets:
            src = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
            dst = ctypes.create_string_buffer(len(data) + len(data) // 8 + 64)
            struct.pack_into("<QQQQ", raw, 0, ctypes.addressof(src), ctypes.addressof(dst), len(data), len(dst))
            n, fin = 0, False
            for _ in range(1000):
                n += ref.lzs_compress_incremental(ctypes.addressof(raw), fin)
                if raw.raw[32] & api.STATUS_END_MARKER:
                    break
                fin = struct.unpack_from("<Q", raw, 16)[0] == 0
            out += dst.raw[:n]
        return bytes(out)

    rng = random.Random(97)
    text = _sample("text", 300000)
    for it in range(25):
        packets = []
        for _ in range(rng.randint(1, 5)):
            kind = rng.randint(0, 3)
            n = rng.choice((0, 1, 40, 700, 9000, 60000))
            a = rng.randint(0, len(text) - n - 1)
            packets.append([text[a:a + n], bytes(n), rng.randbytes(n), (text[a:a + 50] * (n // 50 + 1))[:n]][kind])
        if packets[0]:
            packets.append(packets[0][:len(packets[0]) // 2] + b"!" + packets[-1][:333])
        want = ref_packets(packets)
        c, got = None, b""
        for p in packets:
            lo, hi = rng.choice(((1, 40), (200, 3000), (10000, 100000)))
            part, c = _encode(p, _rand(rng, lo, hi), _rand(rng, max(lo, 3), hi), comp=c)
            got += part
        assert got == want, (it, [len(p) for p in packets])
        back, markers = _decode(want, _rand(rng, 1, 50000), _rand(rng, 1, 80000), stop_at_markers=len(packets))
        assert back == b"".join(packets) and markers == len(packets)
