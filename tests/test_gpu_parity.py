"""Parity of the HIP path with the oracle and the reference-minted fixtures, all calls
through the C-ABI of liblzs.so (include/lzs/lzs.h, include/lzs/lzs_batch.h).

Bit-exact is the bar: this is byte/bit work.  The oracle (oracle/) is the checker only.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle
from conftest import golden_bytes, golden_json, length_bits, uncompressible_sequence
import lzs_compression_amd as lzs
from lzs_compression_amd import workload

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
O = oracle.oracle()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (no fallback exists)")
    assert "gfx950" in lzs.backend_info()


@pytest.fixture(autouse=True)
def _device_route(monkeypatch):
    """Parity of the HIP path: every 4-argument call of this module is a launch, whatever its size (LZS_ROUTE=device).
    The small calls' host route has the same fixtures to itself in tests/test_routes.py, and the route a caller gets
    by default -- by size -- is compared with both there."""
    monkeypatch.setenv("LZS_ROUTE", "device")


def _rows(datas, stride=None):
    """Pack byte strings into a [n, stride] uint8 array + length vector."""
    stride = stride or max(1, max(len(d) for d in datas))
    arr = np.zeros((len(datas), stride), dtype=np.uint8)
    lens = np.zeros(len(datas), dtype=np.uint32)
    for i, d in enumerate(datas):
        arr[i, :len(d)] = np.frombuffer(d, dtype=np.uint8)
        lens[i] = len(d)
    return arr, lens


def _gpu_compress_many(datas, cap=None):
    arr, lens = _rows(datas)
    out, out_len = lzs.compress_batch(arr, lens, cap)
    return [out[i, :out_len[i]].tobytes() for i in range(len(datas))]


def _gpu_decompress_many(streams, cap):
    arr, lens = _rows(streams)
    out, out_len = lzs.decompress_batch(arr, lens, cap)
    return [out[i, :out_len[i]].tobytes() for i in range(len(streams))]


def _gpu_decompress_blocks_kernel(streams, cap):
    """The same through the device-pointer call: always ONE launch of the eight-streams-per-wavefront
    block decoder (the host-buffer call cuts small batches into segments for the stream decoder)."""
    import torch
    arr, lens = _rows(streams)
    x = torch.from_numpy(arr).cuda()
    n = torch.from_numpy(lens.astype(np.int32)).cuda()
    out, out_len = lzs.decompress_blocks(x, n, cap)
    torch.cuda.synchronize()
    out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
    return [out[i, :out_len[i]].tobytes() for i in range(len(streams))]


# ------------------------------------------------------------ reference KATs, one-shot ABI
def test_golden_vector_one_shot():
    """c/src/test/test-lzs-decompression.c:34-96 through lzs_compress()/lzs_decompress()."""
    comp, plain = golden_bytes("kat_compressed_1.bin"), golden_bytes("kat_decompressed_1.bin")
    assert lzs.compress(plain) == comp
    assert lzs.decompress(comp, len(plain) + 520) == plain


@pytest.mark.parametrize("data,hexout", [
    (b"", "c000"), (b"a", "30e000"), (b"aa", "30987000"), (b"aaa", "30e04c00"),
    (b"a" * 9, "30e07c3000"), (b"a" * 10, "30e07c7000"), (b"a" * 24, "30e07fc300"),
    (b"a" * 25, "30e07fc700"), (b"abcXabcYabc", "30988c658c2259c23800"),
])
def test_tiny_vectors_one_shot(data, hexout):
    assert lzs.compress(data).hex() == hexout
    assert lzs.decompress(bytes.fromhex(hexout), 100) == data


def test_config0_4k_roundtrip():
    """BASELINE.json configs[0]: single 4 KiB buffer, lzs_compress + lzs_decompress."""
    plain, comp = golden_bytes("text_4k.bin"), golden_bytes("text_4k.lzs")
    got = lzs.compress(plain, lzs.compressed_max(4096))
    assert got == comp
    assert lzs.decompress(got, 4096) == plain


def test_uncompressible_size_law():
    """c/src/test/test-lzs.c:93-119, all 507 prefix lengths as one ragged batch."""
    seq = uncompressible_sequence()
    datas = [seq[:n] for n in range(len(seq) + 1)]
    comps = _gpu_compress_many(datas, 1000)
    for n, c in enumerate(comps):
        assert len(c) == (n * 9 + 9 + 7) // 8, n
        assert c == O.compress(datas[n])
    backs = _gpu_decompress_many(comps, 1000)
    assert backs == datas


def test_repeated_byte_size_law():
    """c/src/test/test-lzs.c:121-167, lengths 0..1000 as one ragged batch."""
    datas = [b"X" * n for n in range(1001)]
    comps = _gpu_compress_many(datas, 1000)
    for n, c in enumerate(comps):
        bits = {0: 0, 1: 9, 2: 18}.get(n)
        if bits is None:
            bits = 9 + 2 + 7 + length_bits(n - 1)
        assert len(c) == (bits + 9 + 7) // 8, n
    assert _gpu_decompress_many(comps, 1000) == datas


# ------------------------------------------------------------ fixtures minted from the reference
def test_edge_vectors_compress(edge_vectors):
    vecs = edge_vectors["compress"]
    datas = [bytes.fromhex(v["in"]) for v in vecs]
    comps = _gpu_compress_many(datas)
    for v, c in zip(vecs, comps):
        assert c.hex() == v["out"], v["name"]
    # reduced output capacities: the stream is cut, never altered (lzs-compression.c:306-309)
    for v, d in zip(vecs, datas):
        want = bytes.fromhex(v["out"])
        for cap, n in v["capped"].items():
            got = lzs.compress(d, int(cap))
            assert len(got) == n and got == want[:int(cap)], (v["name"], cap)


def test_edge_vectors_decompress(edge_vectors):
    for v in edge_vectors["decompress"]:
        stream = bytes.fromhex(v["in"])
        for cap, want in v["out"].items():
            assert lzs.decompress(stream, int(cap)).hex() == want, (v["name"], cap)


def test_truncation_does_not_touch_past_capacity():
    data = workload.fill("text", 1)[0, :5000].tobytes()
    full = O.compress(data)
    arr, lens = _rows([data])
    for cap in (0, 1, 255, 256, 257, 1000, len(full) - 1):
        x = torch.from_numpy(arr).cuda()
        out = torch.full((1, len(full) + 64), 0xA5, dtype=torch.uint8, device="cuda")
        out_len = torch.zeros(1, dtype=torch.int32, device="cuda")
        lzs.compress_blocks(x, torch.from_numpy(lens.astype(np.int32)).cuda(), cap, out, out_len)
        got = out.cpu().numpy()[0]
        n = int(out_len.item())
        assert n == min(cap, len(full))
        assert got[:n].tobytes() == full[:n]
        assert (got[n:] == 0xA5).all(), cap


@pytest.mark.parametrize("cls", workload.CLASS_NAMES)
def test_class_digests_device_batch(class_digests, cls):
    """256 x 64 KiB blocks per class: lengths and SHA-256 equal the REAL reference's
    (tests/golden/class_digests.json), then a device round trip."""
    want = class_digests["classes"][cls]
    nb, bl = class_digests["nblocks"], class_digests["block_len"]
    blocks = workload.fill(cls, nb, bl, seed=class_digests["seed"])
    assert hashlib.sha256(blocks.tobytes()).hexdigest() == want["input_sha256"]
    x = torch.from_numpy(blocks).cuda()
    slots, lens = lzs.compress_blocks(x)
    torch.cuda.synchronize()
    lens_h = lens.cpu().numpy()
    assert [int(v) for v in lens_h] == want["len"]
    slots_h = slots.cpu().numpy()
    h = hashlib.sha256()
    for b in range(nb):
        h.update(slots_h[b, :lens_h[b]].tobytes())
    assert h.hexdigest() == want["sha256"]
    if cls == "text":
        assert slots_h[0, :lens_h[0]].tobytes() == golden_bytes("text_block0.lzs")
    back, back_len = lzs.decompress_blocks(slots, lens, bl)
    torch.cuda.synchronize()
    assert bool((back_len == bl).all()) and torch.equal(back[:, :bl], x)
    # dense gather: concatenated streams + offsets
    dense, offsets = lzs.compact(slots, lens)
    torch.cuda.synchronize()
    offs = offsets.cpu().numpy()
    assert offs[0] == 0 and (np.diff(offs) == lens_h).all()
    assert hashlib.sha256(dense[:offs[-1]].cpu().numpy().tobytes()).hexdigest() == want["sha256"]


def test_every_variant_of_the_compress_kernel_gives_the_reference_bytes_on_every_class():
    """Round 5: a launch runs one variant of the compress kernel per class of block (small bucket tables and six
    workgroups per CU for blocks of few distinct grams, one full step per pass of the SEARCH loop for blocks that are
    nearly all literals, the default for the rest), and the blocks say which is theirs on the device.  The choice must
    not show in the bytes: each variant forced on ALL three classes (LZS_VARIANT, read once per process: a process
    each) gives the REAL reference's lengths and SHA-256 (tests/golden/class_digests.json), and so does the choice."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = (
        "import hashlib, json, os, torch, lzs_compression_amd as lzs\n"
        "from lzs_compression_amd import workload\n"
        "d = json.load(open(os.path.join(%r, 'tests', 'golden', 'class_digests.json')))\n"
        "for cls in workload.CLASS_NAMES:\n"
        "    x = torch.from_numpy(workload.fill(cls, d['nblocks'], d['block_len'], seed=d['seed'])).cuda()\n"
        "    slots, n = lzs.compress_blocks(x)\n"
        "    torch.cuda.synchronize()\n"
        "    out, n = slots.cpu().numpy(), n.cpu().numpy()\n"
        "    assert n.tolist() == d['classes'][cls]['len'], cls\n"
        "    h = hashlib.sha256()\n"
        "    for b in range(len(n)): h.update(out[b, :n[b]].tobytes())\n"
        "    assert h.hexdigest() == d['classes'][cls]['sha256'], cls\n"
        "print('ok')\n" % root)
    for variant in ("", "text", "few", "lit"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("LZS_VARIANT", None)
        if variant:
            env["LZS_VARIANT"] = variant
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout.split(), (variant, r.stderr[-2000:])


def test_the_blocks_of_the_three_classes_get_their_variants_and_mixed_batches_stay_exact():
    """lzs_classify_blocks_kernel: the distinct 2-grams among a block's first 2048 positions (a 4096-bit hashed set) --
    text ~430, the low-entropy class <= 53, random bytes ~1610 -- choose the variant; blocks shorter than that take the
    default.  A batch of all three classes interleaved, ragged, is compressed by three launches over one grid and every
    block equals the oracle's."""
    import ctypes
    L = lzs.lib()
    L.lzs_hip_classify_blocks.restype = ctypes.c_int
    L.lzs_hip_classify_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    nb = 96
    parts = [workload.fill(c, nb) for c in workload.CLASS_NAMES]
    mixed = np.stack([parts[b % 3][b // 3] for b in range(3 * nb)])
    x = torch.from_numpy(mixed).cuda()
    codes = torch.zeros(3 * nb, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert L.lzs_hip_classify_blocks(codes.data_ptr(), x.data_ptr(), 65536, None, 65536, 3 * nb, None) == 0
    torch.cuda.synchronize()
    got = (codes.cpu().numpy().astype(np.int64) & 0xF).tolist()
    assert got == [1, 2, 3] * nb, got[:12]
    lens = np.full(3 * nb, 65536, dtype=np.uint32)
    lens[5::7] = np.random.default_rng(3).integers(0, 65536, len(lens[5::7]))
    lens[0], lens[1], lens[2] = 2048, 2049, 0
    slots, n = lzs.compress_blocks(x, torch.from_numpy(lens.astype(np.int32)).cuda())
    torch.cuda.synchronize()
    out, n = slots.cpu().numpy(), n.cpu().numpy()
    for b in range(3 * nb):
        want = O.compress(mixed[b, :lens[b]].tobytes())
        assert int(n[b]) == len(want) and out[b, :n[b]].tobytes() == want, (b, lens[b])


def test_batches_of_changing_classes_follow_the_launches_before_and_stay_exact():
    """Which variants of the compress kernel a launch runs follows what the classifiers of the launches before it met
    (three words of pinned host memory, read without waiting): a block whose class has no variant in the launch goes to
    the default's.  That guess can be slow, never wrong: batches of one class after the other, back and forth, mixed,
    with the device left no time between them and with a wait after each -- every block against the oracle."""
    nb = 128
    parts = {c: workload.fill(c, nb) for c in workload.CLASS_NAMES}
    mixed = np.stack([parts[workload.CLASS_NAMES[b % 3]][b // 3] for b in range(nb)])
    want = {c: [O.compress(parts[c][b].tobytes()) for b in range(0, nb, 9)] for c in workload.CLASS_NAMES}
    want["mixed"] = [O.compress(mixed[b].tobytes()) for b in range(0, nb, 9)]
    dev = {c: torch.from_numpy(parts[c]).cuda() for c in workload.CLASS_NAMES}
    dev["mixed"] = torch.from_numpy(mixed).cuda()
    order = ["text", "text", "lowent", "lowent", "random", "text", "lowent", "text", "mixed", "random", "random", "mixed", "lowent"]
    for wait in (True, False):
        results = []
        for name in order * 2:
            slots, lens = lzs.compress_blocks(dev[name])
            if wait:
                torch.cuda.synchronize()
            results.append((name, slots, lens))
        torch.cuda.synchronize()
        for name, slots, lens in results:
            out, n = slots.cpu().numpy(), lens.cpu().numpy()
            for k, b in enumerate(range(0, nb, 9)):
                assert int(n[b]) == len(want[name][k]) and out[b, :n[b]].tobytes() == want[name][k], (wait, name, b)


# ------------------------------------------------------------ differential vs the oracle
def _fuzz_inputs(rng, count, maxlen):
    for _ in range(count):
        n = int(rng.integers(0, maxlen))
        kind = int(rng.integers(0, 6))
        if kind == 0:
            yield bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            yield bytes(rng.integers(0, int(rng.integers(1, 6)), n, dtype=np.uint8) + 65)
        elif kind == 2:
            piece = bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8))
            yield (piece * (n // len(piece) + 1))[:n]
        elif kind == 3:
            out = bytearray()
            while len(out) < n:
                if rng.integers(0, 2):
                    out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 3000))
                else:
                    out += bytes(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8))
            yield bytes(out[:n])
        elif kind == 4:
            words = [bytes(rng.integers(97, 123, int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(200)]
            out = bytearray()
            while len(out) < n:
                out += words[int(rng.integers(0, 200))] + b" "
            yield bytes(out[:n])
        else:
            yield bytes(rng.integers(0, 4, n, dtype=np.uint8) + 48)


def test_fuzz_ragged_batch_vs_oracle():
    rng = np.random.default_rng(2024)
    datas = list(_fuzz_inputs(rng, 600, 9000))
    comps = _gpu_compress_many(datas)
    for d, c in zip(datas, comps):
        assert c == O.compress(d)
    backs = _gpu_decompress_many(comps, 9000)
    assert backs == datas


def test_long_matches_across_chunk_and_pool_boundaries():
    """Matches that fill the 12-byte search cap and run on: exact totals around the limits of the
    in-kernel extension (8 + 59, then "open"), starting at every alignment to the 64-position
    chunks and the 512-position pools, alone, back to back and nested in runs (the pass that
    completes them derives most lengths from a neighbour instead of comparing bytes)."""
    rng = np.random.default_rng(4242)
    datas = []
    for total in (12, 13, 14, 19, 20, 23, 24, 27, 66, 67, 68, 69, 70, 127, 128, 300, 700):
        for _ in range(12):
            lead = int(rng.integers(0, 1100))
            gap = int(rng.integers(0, 1900))
            phrase = bytes(rng.integers(0, 256, total, dtype=np.uint8))
            a = bytes(rng.integers(0, 256, lead, dtype=np.uint8))
            g = bytes(rng.integers(0, 256, gap, dtype=np.uint8))
            tail = bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))
            datas.append(a + phrase + g + phrase + tail)
            # the same phrase three times: nested candidates at two offsets
            datas.append(a + phrase + g[: gap // 2] + phrase + g[gap // 2:] + phrase[: total - 1] + tail)
            # a match that ends exactly at the end of the input
            datas.append(a + phrase + g + phrase)
    # runs of one byte and of short periods whose ends fall around every chunk boundary
    for period in (1, 2, 3, 16, 63, 64, 65):
        for _ in range(10):
            lead = int(rng.integers(0, 700))
            run = int(rng.integers(13, 2600))
            unit = bytes(rng.integers(0, 256, period, dtype=np.uint8))
            a = bytes(rng.integers(0, 256, lead, dtype=np.uint8))
            datas.append(a + (unit * (run // period + 1))[:run] + bytes(rng.integers(0, 256, 30, dtype=np.uint8)))
    comps = _gpu_compress_many(datas)
    for d, c in zip(datas, comps):
        assert c == O.compress(d)


def test_matches_into_what_an_open_match_ran_over():
    """After an open match (longer than the in-kernel extension) only the last offset + 12 of the
    positions it ran over go into the chains: a position whose 12 bytes repeat one period on inside
    the match is dominated by that nearer position.  Periods of every size class
    against runs of many lengths, followed by text that matches INTO the run at its start, its
    middle, just before the dominated zone ends and at its very end, at distances up to the window,
    with the next long match close behind (run mode: small pools parsed at once), and with several
    runs in a row; against the oracle."""
    rng = np.random.default_rng(777)
    datas = []

    def rnd(n):
        return bytes(rng.integers(0, 256, int(n), dtype=np.uint8))

    for period in (1, 2, 3, 5, 12, 13, 16, 17, 63, 64, 65, 127, 128, 129, 500, 1000, 2046, 2047):
        for run in (70, 90, 200, 700, 2100, 4080, 4081, 4200, 9000):
            unit = rnd(period)
            body = (unit * (run // period + 2))[:run]
            lead = rnd(rng.integers(0, 900))
            pieces = [lead, unit, body]
            # text that copies from inside the run: phrase of 3..30 bytes taken at several places
            for where in (0, run // 2, max(0, run - period - 40), max(0, run - period - 13), max(0, run - period - 12),
                          max(0, run - 30), max(0, run - 12)):
                n = int(rng.integers(3, 31))
                pieces += [rnd(rng.integers(1, 20)), body[where:where + n]]
            pieces.append(rnd(rng.integers(0, 1500)))
            pieces.append(body[max(0, run - 2000):][:25])            # ... and from far away, near the window's edge
            # the next long match close behind, at another period
            other = rnd(int(rng.integers(1, 40)))
            pieces += [rnd(rng.integers(0, 12)), other * (int(rng.integers(80, 900)) // len(other) + 1), rnd(5)]
            datas.append(b"".join(pieces))
    # several runs in a row with nothing or little between them, zero runs among them
    for _ in range(40):
        pieces = [rnd(rng.integers(0, 300))]
        for _ in range(int(rng.integers(2, 9))):
            period = int(rng.choice([1, 1, 2, 4, 16, 16, 33, 200]))
            unit = bytes(period) if rng.integers(0, 3) == 0 else rnd(period)
            pieces += [(unit * 400)[:int(rng.integers(13, 5000))], rnd(rng.integers(0, 4))]
        datas.append(b"".join(pieces))
    comps = _gpu_compress_many(datas)
    for i, (d, c) in enumerate(zip(datas, comps)):
        assert c == O.compress(d), i
    # and as ONE stream each through the segment route (half-KiB segments: a run covers many of them)
    for d in datas[::7]:
        assert lzs.compress(d) == O.compress(d)


def test_sparse_matches_in_high_entropy_data():
    """High-entropy mode: after a pool of nearly all literals, positions without a
    seed walk the 2-byte chain alone, with the full nearest-longest rule.  Random bytes with sparse
    copies of 2..14 bytes from near and far (several candidates of different lengths for the same
    two first bytes, the longer one farther away; copies of copies; short runs that seed offset 1),
    so that the pools stay above 8 bits a position while the matches that exist must all be found;
    then the same data flowing into text and back (the mode switches with a lag of two pools)."""
    rng = np.random.default_rng(31337)
    datas = []
    text = workload.fill("text", 2, 65536).tobytes()
    for trial in range(60):
        buf = bytearray(rng.integers(0, 256, int(rng.integers(2100, 5000)), dtype=np.uint8).tobytes())
        for _ in range(int(rng.integers(30, 200))):
            buf += rng.integers(0, 256, int(rng.integers(20, 160)), dtype=np.uint8).tobytes()
            kind = int(rng.integers(0, 5))
            back = int(rng.integers(1, min(len(buf), 2300)))
            n = int(rng.integers(2, 15))
            if kind == 0:                                      # a plain copy
                buf += buf[len(buf) - back:len(buf) - back + n]
            elif kind == 1:                                    # two candidates with the same first bytes: near short, far long
                far = buf[len(buf) - back:len(buf) - back + n]
                buf += far[:2] + bytes([far[2] ^ 1 if len(far) > 2 else 7]) + rng.integers(0, 256, 9, dtype=np.uint8).tobytes() + far
            elif kind == 2:                                    # a short run: offset 1 seeds the search
                buf += bytes([int(rng.integers(0, 256))]) * int(rng.integers(2, 11))
            elif kind == 3:                                    # a copy of a copy, one byte shorter
                c = buf[len(buf) - back:len(buf) - back + n]
                buf += c + rng.integers(0, 256, 3, dtype=np.uint8).tobytes() + c[:-1]
            else:                                              # a copy from exactly the window's edge and one past it
                if len(buf) > 2060:
                    buf += buf[len(buf) - 2047:len(buf) - 2047 + n] + bytes([1]) + buf[len(buf) - 2049:len(buf) - 2049 + n]
        if trial % 3 == 0:
            at = int(rng.integers(0, 60000))
            buf += text[at:at + int(rng.integers(600, 3000))] + rng.integers(0, 256, 1500, dtype=np.uint8).tobytes() + buf[100:160]
        datas.append(bytes(buf))
    comps = _gpu_compress_many(datas)
    for i, (d, c) in enumerate(zip(datas, comps)):
        assert c == O.compress(d), i


def test_fuzz_mixed_segments_up_to_200k_vs_oracle():
    """Longer inputs stitched from segments of different kinds: runs and periodic repeats far
    longer than the window (the kernel finishes those as one open match and skips most of the
    chain inserts), text, noise and copies of earlier material at offsets around the window
    size -- so that every kind of segment is entered from every other kind, at every phase
    of the 512-position pools."""
    rng = np.random.default_rng(909)
    words = [bytes(rng.integers(97, 123, int(rng.integers(1, 10)), dtype=np.uint8)) for _ in range(400)]
    datas = []
    for _ in range(60):
        n = int(rng.integers(20_000, 200_000))
        out = bytearray()
        while len(out) < n:
            kind = int(rng.integers(0, 6))
            if kind == 0:
                out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 7000))
            elif kind == 1:
                unit = bytes(rng.integers(0, 256, int(rng.integers(2, 300)), dtype=np.uint8))
                out += unit * int(rng.integers(1, 6000 // len(unit) + 2))
            elif kind == 2:
                out += bytes(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8))
            elif kind == 3:
                for _ in range(int(rng.integers(1, 400))):
                    out += words[int(rng.integers(0, 400))] + b" "
            elif kind == 4 and len(out) > 100:
                back = int(rng.integers(1, min(len(out), 2300)))
                take = int(rng.integers(2, 400))
                for i in range(take):                       # overlapping copy, byte by byte
                    out.append(out[len(out) - back])
            else:
                out += bytes(rng.integers(0, 3, int(rng.integers(1, 2000)), dtype=np.uint8) + 120)
        datas.append(bytes(out[:n]))
    comps = _gpu_compress_many(datas)
    for d, c in zip(datas, comps):
        assert c == O.compress(d)
    backs = _gpu_decompress_many(comps, 200_000)
    assert backs == datas


def test_block_decoder_two_tokens_per_trip():
    """The block decoder takes a second token in a trip when its copy reads nothing the first writes
    (DESIGN.md 3.3).  Streams of short matches at SMALL offsets (the second token often depends on
    the first: must wait for the next trip), literal runs of every length in front of matches,
    matches of 8 that open an extension right behind a short one, sources before out[0] (zero
    fill) in the second token; whole, and with the output cut at every one of the last 40
    positions (the room test of the second token); against the oracle's decoder."""
    rng = np.random.default_rng(9001)
    datas = []
    for trial in range(24):
        buf = bytearray(rng.integers(97, 123, 40, dtype=np.uint8).tobytes())
        while len(buf) < 20000:
            kind = int(rng.integers(0, 6))
            if kind == 0:                                      # copy of the last few bytes: offset <= length of the two tokens together
                back = int(rng.integers(1, 17)); n = int(rng.integers(2, 9))
                for _ in range(n): buf.append(buf[-back])
            elif kind == 1:                                    # literals, 1..9 of them
                buf += rng.integers(0, 256, int(rng.integers(1, 10)), dtype=np.uint8).tobytes()
            elif kind == 2:                                    # a copy from far away, 2..7 bytes
                back = int(rng.integers(20, min(len(buf), 2047))); n = int(rng.integers(2, 8))
                buf += buf[len(buf) - back:len(buf) - back + n]
            elif kind == 3:                                    # exactly 8, and 8 + a few (extension right behind)
                back = int(rng.integers(20, min(len(buf), 2047))); n = int(rng.integers(8, 12))
                buf += buf[len(buf) - back:len(buf) - back + n]
            elif kind == 4:                                    # two short copies back to back from far away
                for _ in range(2):
                    back = int(rng.integers(30, min(len(buf), 2047))); n = int(rng.integers(2, 5))
                    buf += buf[len(buf) - back:len(buf) - back + n]
            else:                                              # a run
                buf += bytes([int(rng.integers(0, 256))]) * int(rng.integers(2, 40))
        datas.append(bytes(buf))
    comps = _gpu_compress_many(datas)
    for d, c in zip(datas, comps):
        assert c == O.compress(d) and 0.25 < len(c) / len(d) < 0.9      # the ratio that selects the two-token form
    big = max(len(d) for d in datas)
    for route in (_gpu_decompress_blocks_kernel, _gpu_decompress_many):      # the block decoder; the segment decoder
        for d, back in zip(datas, route(comps, big)):
            assert back == d
    # cut capacities: the last 40 positions (all streams in one batch per capacity: one capacity for all)
    short = min(len(d) for d in datas)
    for k in range(0, 40):
        cap = short - k
        for c, back in zip(comps, _gpu_decompress_blocks_kernel(comps, cap)):
            assert back == O.decompress(c, cap)
    # sources before out[0]: streams that begin in the middle of another one's tokens decode to zeros there
    junk = [c[cut:] for c in comps[:8] for cut in (7, 64, 333)]
    for j, back in zip(junk, _gpu_decompress_blocks_kernel(junk, 30000)):
        assert back == O.decompress(j, 30000)


def test_fuzz_decoder_on_garbage_vs_oracle():
    rng = np.random.default_rng(77)
    streams = [bytes(rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8)) for _ in range(300)]
    streams += [bytes(rng.integers(128, 256, int(rng.integers(0, 100)), dtype=np.uint8)) for _ in range(100)]
    for cap in (0, 7, 4096):
        got = _gpu_decompress_many(streams, cap)
        for s, g in zip(streams, got):
            assert g == O.decompress(s, cap)


def test_block_decoder_counts_bits_before_it_believes_them():
    """The block decoder does not mask a stream's last word: whatever follows a stream in
    its row -- here all-ones, random bytes, the continuation of the stream that was cut -- must never be
    decoded, whatever token the cut falls into.  Streams of every kind cut at every length over a range,
    rows wider than the streams with the tail poisoned, against the oracle on the cut stream."""
    import torch
    rng = np.random.default_rng(41)
    text = bytes(workload.fill("text", 1).reshape(-1))[:3000]
    kinds = [O.compress(text),                                                    # short matches and literals
             O.compress(bytes(rng.integers(0, 256, 1500, dtype=np.uint8))),       # runs of literals
             O.compress(b"ab" * 40 + b"\0" * 900 + text[:200] + b"x" * 300),      # length nibbles
             O.compress(text[:700]) + O.compress(text[100:900])]                  # an end marker in the middle
    cuts = []
    for c in kinds:
        cuts += [c[:n] for n in list(range(0, 140)) + list(range(len(c) - 40, len(c) + 1))]
    stride = max(len(c) for c in cuts) + 24
    for poison in ("ones", "random", "rest"):
        arr = np.full((len(cuts), stride), 0xFF, dtype=np.uint8)
        if poison == "random":
            arr = rng.integers(0, 256, arr.shape, dtype=np.uint8)
        for i, c in enumerate(cuts):
            arr[i, :len(c)] = np.frombuffer(c, dtype=np.uint8)
        if poison == "rest":                                                      # the stream goes on behind its stated length
            k = 0
            for c in kinds:
                for n in list(range(0, 140)) + list(range(len(c) - 40, len(c) + 1)):
                    m = min(len(c), stride)
                    arr[k, :m] = np.frombuffer(c[:m], dtype=np.uint8)
                    k += 1
        lens = torch.tensor([len(c) for c in cuts], dtype=torch.int32, device="cuda")
        for cap in (5000, 777):
            out, out_len = lzs.decompress_blocks(torch.from_numpy(arr).cuda(), lens, cap)
            torch.cuda.synchronize()
            out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
            for i, c in enumerate(cuts):
                assert out[i, :out_len[i]].tobytes() == O.decompress(c, cap), (poison, cap, i, len(c))


def test_unaligned_strides_and_bases():
    """Any alignment of block bases / strides is accepted (slow path), same bytes."""
    rng = np.random.default_rng(3)
    datas = list(_fuzz_inputs(rng, 40, 5000))
    stride_in, stride_out = 5003, lzs.compressed_max(5003) + 1      # odd strides
    flat = torch.zeros(len(datas) * stride_in + 7, dtype=torch.uint8, device="cuda")
    x = flat[3:3 + len(datas) * stride_in].view(len(datas), stride_in)      # base misaligned by 3
    lens = torch.tensor([len(d) for d in datas], dtype=torch.int32, device="cuda")
    for i, d in enumerate(datas):
        x[i, :len(d)] = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda() if d else x[i, :0]
    oflat = torch.zeros(len(datas) * stride_out + 5, dtype=torch.uint8, device="cuda")
    out = oflat[1:1 + len(datas) * stride_out].view(len(datas), stride_out)
    out_len = torch.zeros(len(datas), dtype=torch.int32, device="cuda")
    lzs.compress_blocks(x, lens, stride_out - 1, out, out_len)
    torch.cuda.synchronize()
    o, l = out.cpu().numpy(), out_len.cpu().numpy()
    for i, d in enumerate(datas):
        assert o[i, :l[i]].tobytes() == O.compress(d), i
    back = torch.zeros_like(x)
    back_len = torch.zeros_like(lens)
    lzs.decompress_blocks(out, out_len, stride_in, back, back_len)
    torch.cuda.synchronize()
    assert torch.equal(back_len, lens)
    bh = back.cpu().numpy()
    for i, d in enumerate(datas):
        assert bh[i, :len(d)].tobytes() == d


def test_single_long_stream_keeps_history_across_64k():
    """The 4-argument call never splits a buffer: one stream, history carried throughout."""
    data = workload.fill("text", 4)[:, :].tobytes()[:200_000]
    got = lzs.compress(data)
    assert got == O.compress(data)
    assert lzs.decompress(got, len(data)) == data


def test_one_shot_calls_from_many_host_threads():
    """The reference's calls are re-entrant (no globals: lzs-compression.c:100-124 holds only
    const tables); ours must stay callable concurrently from many host threads."""
    import threading
    rng = np.random.default_rng(9)
    datas = list(_fuzz_inputs(rng, 48, 20000))
    want = [O.compress(d) for d in datas]
    errors = []

    def worker(tid):
        try:
            for rep in range(3):
                for i in range(tid, len(datas), 8):
                    got = lzs.compress(datas[i])
                    if got != want[i] or lzs.decompress(got, len(datas[i]) + 1) != datas[i]:
                        errors.append((tid, i))
        except Exception as e:      # noqa: BLE001 - collected for the assertion below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_segment_paths_and_incremental_calls_from_many_host_threads():
    """The same for the paths that run rounds of launches with host decisions in between
    (segments of one stream, scan / decode / resolve, small batches, the incremental interface):
    each host thread has its own stream and staging buffers, so they must not disturb one another."""
    import threading
    rng = np.random.default_rng(12)
    text = bytes(workload.fill("text", 8).reshape(-1))
    datas = []
    for i in range(18):
        a = int(rng.integers(0, len(text) - 300_001))
        d = text[a: a + int(rng.integers(30_000, 300_000))]
        if i % 3 == 0:
            d = d[:20_000] + bytes(int(rng.integers(1000, 90_000))) + d[20_000:]
        datas.append(d)
    want = [O.compress(d) for d in datas]
    errors = []

    def worker(tid):
        try:
            r = np.random.default_rng(100 + tid)
            for i in range(tid, len(datas), 6):
                d, w = datas[i], want[i]
                if lzs.compress(d) != w or lzs.decompress(w, len(d) + 3) != d:
                    errors.append((tid, i, "one-shot"))
                c, out, pos = lzs.IncrementalCompressor(), b"", 0
                while pos < len(d):
                    k = int(r.integers(1000, 70_000))
                    out += c.step(d[pos:pos + k], 200_000)[0]
                    pos += k
                out += c.step(b"", 200_000, True)[0]
                if out != w:
                    errors.append((tid, i, "incremental compress"))
                dd, back, pend = lzs.IncrementalDecompressor(), b"", w
                while pend:
                    got, used, _ = dd.step(pend[:int(r.integers(1000, 70_000))], 500_000)
                    back += got
                    pend = pend[used:]
                if back != d:
                    errors.append((tid, i, "incremental decompress"))
                got = _gpu_decompress_many([w, want[(i + 1) % len(want)]], 300_000)
                if got != [d, datas[(i + 1) % len(datas)]]:
                    errors.append((tid, i, "small batch"))
        except Exception as e:      # noqa: BLE001 - collected for the assertion below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_multi_megabyte_single_stream():
    """One 3 MiB stream through the 4-argument call: positions well past 2^16 (16-bit head
    aliasing would show here) and a mix of the three classes, bit-exact vs the oracle."""
    parts = [workload.fill("text", 20).tobytes(), workload.fill("lowent", 12).tobytes(),
             workload.fill("random", 8).tobytes(), workload.fill("text", 8, first_block=100).tobytes()]
    data = b"".join(parts)[:3 * 1024 * 1024 + 12345]
    got = lzs.compress(data)
    assert got == O.compress(data)
    assert lzs.decompress(got, len(data)) == data


def test_short_segments_leave_lanes_that_never_took_a_position(monkeypatch):
    """Round 5's regression: a segment of a stream may hold fewer positions than the workgroup has lanes (256-byte segments;
    a last segment of a few bytes; a piece of the incremental interface), so some lanes never take a position in SEARCH.
    With "idle" told by a lane standing on its own position (the result stored when the walk has ended, not every step),
    such a lane must start out as one whose walk HAS ended -- a key that is none, not 0 -- or it wanders off along
    garbage links and SEARCH never ends: the single-stream file tool hung on 313 KB.  Every segment size from the
    smallest, lengths that leave 1..300 bytes in the last segment."""
    blob = workload.fill("text", 6).tobytes()
    for seg in (256, 512, 1024):
        monkeypatch.setenv("LZS_STREAM_SEG", str(seg))
        monkeypatch.setenv("LZS_FORCE_STREAM", "1")
        for n in (1, 2, 17, 255, 256, 257, 300, 6145, 6145 + 77, 40000 + 1, 65536 + 255, 312778):
            d = blob[:n]
            assert lzs.compress(d) == O.compress(d), (seg, n)
    monkeypatch.delenv("LZS_STREAM_SEG")
    monkeypatch.delenv("LZS_FORCE_STREAM")


def test_one_long_stream_on_many_workgroups_vs_oracle():
    """lzs_compress() of a buffer of 6 KiB or more is cut into segments (0.5-64 KiB), one workgroup
    each, stitched at the bit level.  Same bytes as the reference: every class,
    lengths that are not multiples of the segment, runs and repeats spanning many segments, and
    a capacity that cuts the stream."""
    rng = np.random.default_rng(31)
    datas = []
    for cls in workload.CLASS_NAMES:
        blocks = workload.fill(cls, 40).reshape(-1)
        datas.append(bytes(blocks[: 40 * 65536 - 12345]))
        datas.append(bytes(blocks[: 131073]))
    mix = bytearray()
    text = bytes(workload.fill("text", 8).reshape(-1))
    while len(mix) < 3_000_000:
        k = int(rng.integers(0, 4))
        if k == 0:
            mix += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 250_000))
        elif k == 1:
            a = int(rng.integers(0, len(text) - 1)); mix += text[a: a + int(rng.integers(1, 150_000))]
        elif k == 2:
            mix += bytes(rng.integers(0, 256, int(rng.integers(1, 70_000)), dtype=np.uint8))
        else:
            unit = bytes(rng.integers(0, 256, int(rng.integers(2, 2500)), dtype=np.uint8))
            mix += unit * int(rng.integers(1, 150))
    datas.append(bytes(mix))
    datas.append(bytes(1_000_000))                      # one run over 15 segments
    datas.append(text[:70_000] * 9)                     # a period longer than a segment: no matches across
    for d in datas:
        want = O.compress(d)
        assert lzs.compress(d) == want
        cut = len(want) // 3
        assert lzs.compress(d, cut) == want[:cut]


def test_block_decoder_input_feed_edges():
    """The block decoder reads its input with buffer loads bounded by the end of the last stream of
    a wavefront: streams at every byte alignment, lengths around the 16- and
    128-byte chunk borders, the very last bytes of the allocation, and streams so far apart that
    the launcher puts one on a wavefront (a 32-bit extent would not span eight)."""
    import torch
    rng = np.random.default_rng(5)
    text = bytes(workload.fill("text", 2).reshape(-1))
    plains = [text[:n] for n in (1, 2, 13, 14, 15, 16, 17, 100, 111, 112, 113, 127, 128, 129, 140, 141, 142, 143, 144, 145, 255, 256, 257, 1000, 4096, 65536)]
    comps = [O.compress(p) for p in plains]
    # (a) packed back to back at odd offsets, the last one ending exactly at the end of the tensor
    for shift in range(0, 5):
        stride = max(len(c) for c in comps) + shift                       # rows at every alignment as b * stride walks
        stride += (stride % 2 == 0)                                       # odd stride: all four alignments occur
        buf = np.zeros(((len(comps) - 1) * stride + len(comps[-1]),), dtype=np.uint8)
        for b, c in enumerate(comps):
            buf[b * stride: b * stride + len(c)] = np.frombuffer(c, dtype=np.uint8)
        flat = torch.from_numpy(buf).cuda()
        x = torch.as_strided(flat, (len(comps), len(comps[-1])), (stride, 1))
        lens = torch.tensor([len(c) for c in comps], dtype=torch.int32, device="cuda")
        out, out_len = lzs.decompress_blocks(x, lens, 65536)
        torch.cuda.synchronize()
        got_len = out_len.cpu().numpy()
        out = out.cpu().numpy()
        for b, p in enumerate(plains):
            assert got_len[b] == len(p) and out[b, :len(p)].tobytes() == p, (shift, b, len(p))
    # (b) far apart: 160 MB between streams, lengths on the device -> one stream per wavefront
    far = 160 << 20
    few = comps[-5:]
    flat = torch.zeros(far * (len(few) - 1) + len(few[-1]), dtype=torch.uint8, device="cuda")
    for b, c in enumerate(few):
        flat[b * far: b * far + len(c)] = torch.from_numpy(np.frombuffer(c, dtype=np.uint8).copy()).cuda()
    x = torch.as_strided(flat, (len(few), len(few[-1])), (far, 1))
    lens = torch.tensor([len(c) for c in few], dtype=torch.int32, device="cuda")
    out, out_len = lzs.decompress_blocks(x, lens, 65536)
    torch.cuda.synchronize()
    for b, p in enumerate(plains[-5:]):
        assert int(out_len[b]) == len(p) and out[b, :len(p)].cpu().numpy().tobytes() == p, ("far", b)


def test_stream_decompress_resolve_shortcuts_agree_with_plain_rounds(monkeypatch):
    """How the copies across segment borders are settled after DECODE is a matter of
    speed, not of result: tails by chunks of segments, pointer jumping on the tails alone, or plain
    rounds over everything (LZS_NO_CHUNKS / LZS_NO_TAILS) give the plain text back -- on chains of
    copies as long as the stream, on segments that produce less than a window (tails spanning
    many of them) and far more than one, and at segment sizes from 256 bytes to 8 KiB."""
    import torch
    rng = np.random.default_rng(77)
    text = bytes(workload.fill("text", 24).reshape(-1))
    datas = [
        text,                                                              # 1.5 MiB: words copied from copies of copies
        bytes(workload.fill("lowent", 16).reshape(-1)),                    # a segment produces 25 x its size
        bytes(workload.fill("random", 6).reshape(-1)),                     # a 256-byte segment produces ~228 bytes
        (text[:1500] + bytes(rng.integers(0, 256, 700, dtype=np.uint8))) * 400,   # a period just over the window: copies 2200 back never, 1500 back always
        bytes(rng.integers(0, 256, 1800, dtype=np.uint8)) * 700,           # one 1800-byte unit, copied on and on
    ]
    for d in datas:
        comp = lzs.compress(d)
        for seg in ("256", "1024", "8192"):
            monkeypatch.setenv("LZS_DEC_SEG", seg)
            monkeypatch.setenv("LZS_FORCE_STREAM", "1")
            for off in ((), ("LZS_NO_CHUNKS",), ("LZS_NO_TAILS",)):
                for name in ("LZS_NO_CHUNKS", "LZS_NO_TAILS"):
                    monkeypatch.delenv(name, raising=False)
                for name in off:
                    monkeypatch.setenv(name, "1")
                assert lzs.decompress(comp, len(d) + 8) == d, (len(d), seg, off)
                x = torch.frombuffer(bytearray(comp), dtype=torch.uint8).cuda()
                back, n = lzs.decompress_stream(x, len(d) + 16)
                assert n == len(d) and back[:n].cpu().numpy().tobytes() == d, (len(d), seg, off, "device")


def test_stream_device_entry_point_1gib_text():
    """lzs_compress_stream_device() at full size: 1 GiB of text as ONE stream (16384 segments).
    The oracle needs minutes for that, so the first 24 MiB of input are compressed by the oracle
    and compared byte for byte: the stream of a prefix is a prefix of the stream, up to the last
    tokens before the cut (a token looks at most 12 + 59 bytes ahead) and the end marker."""
    import torch
    x = torch.from_numpy(workload.fill("text", 16384)).cuda()
    out, nbytes = lzs.compress_stream(x.reshape(-1))
    head = bytes(x.reshape(-1)[: 24 << 20].cpu().numpy())
    want = O.compress(head)
    got = bytes(out[: len(want)].cpu().numpy())
    # the stream of a prefix is a prefix of the stream, up to the last few tokens before the cut
    # (the search looks at most 12 + 59 bytes ahead of a token start; the oracle's end marker differs)
    keep = len(want) - 64
    assert got[:keep] == want[:keep]
    assert 0.5 < nbytes / x.numel() < 0.6
    # and back, still on the device: the stream decompressed by many wavefronts is the input
    back, n = lzs.decompress_stream(out[:nbytes], x.numel() + 16)
    assert n == x.numel() and torch.equal(back[:n], x.reshape(-1))


def test_largest_stream_the_calls_accept():
    """LZS_BLOCK_MAX (3 GiB, lzs_batch.h): one stream of exactly that size through
    lzs_compress_stream_device() and back through lzs_decompress_stream_device() -- positions are
    32-bit inside the kernels, this is where they are largest.  One byte more is refused."""
    import torch
    limit = 3 << 30
    blocks = torch.from_numpy(workload.fill("text", 4096)).cuda().reshape(-1)           # 256 MiB of text ...
    x = torch.empty(limit, dtype=torch.uint8, device="cuda")
    for i in range(0, limit, blocks.numel()):                                           # ... twelve times,
        x[i:i + blocks.numel()] = blocks
    x[(1 << 30) + 5: (1 << 30) + 5 + (200 << 20)] = 7                                   # with a 200 MiB run inside
    out, nbytes = lzs.compress_stream(x)
    head = bytes(x[: 8 << 20].cpu().numpy())
    want = O.compress(head)
    assert bytes(out[: len(want) - 64].cpu().numpy()) == want[: len(want) - 64]
    back, n = lzs.decompress_stream(out[:nbytes], limit + 16)
    assert n == limit and torch.equal(back[:n], x)
    del back
    big = torch.empty(limit + 1, dtype=torch.uint8, device="cuda")
    with pytest.raises(lzs.LzsError):
        lzs.compress_stream(big)


def _bits_before_the_end_marker(stream: np.ndarray, nbits: int) -> np.ndarray:
    """The last `nbits` bits in front of a stream's end marker (1 1 0000000 + pad: its last set bit
    is the marker's second bit), as an array of 0/1 -- independent of the byte phase."""
    tail = np.unpackbits(stream[-(nbits // 8 + 16):])
    last = int(np.flatnonzero(tail)[-1])
    return tail[last - 1 - nbits:last - 1]


def test_buffers_beyond_the_32_bit_positions_of_a_launch():
    """The reference's one-shot calls take any size_t (lzs.h:218,229).  5 GiB through the plain
    lzs_compress() / lzs_decompress() (carried across pieces of <= 1 GiB: history, undecided tail,
    a long match that runs over a piece border, the partial output byte): the first 16 MiB of the
    stream are the oracle's, the last 8 Mbit in front of the end marker are those of the oracle run
    on the tail of the input (the search is a pure function of position and window, greedy parses
    entered a window earlier have merged), and the round trip is exact."""
    import ctypes
    n = 5 << 30
    part = workload.fill("text", 8192).reshape(-1)                      # 512 MiB of text
    x = np.empty(n, dtype=np.uint8)
    for i in range(0, n, part.size):
        x[i:i + part.size] = part
    x[(2 << 30) - 1000:(2 << 30) + (40 << 20)] = 9                      # a run across the 2 GiB mark
    x[(3 << 30) - 7:(3 << 30) + 5] = np.arange(12, dtype=np.uint8)      # something at the old limit
    cap = lzs.compressed_max(n)
    out = np.empty(cap, dtype=np.uint8)
    L = lzs.lib()
    got = L.lzs_compress(out.ctypes.data, cap, x.ctypes.data, n)
    assert 0 < got < n, lzs.last_error()
    head = O.compress(x[:17 << 20].tobytes())
    assert out[:len(head) - 64].tobytes() == head[:len(head) - 64]
    tail_in = x[n - (17 << 20):]
    tail = np.frombuffer(O.compress(tail_in.tobytes()), dtype=np.uint8)
    nb = 8 << 20
    assert np.array_equal(_bits_before_the_end_marker(out[:got], nb), _bits_before_the_end_marker(tail, nb))
    # cut capacity: the prefix of the same stream, and nothing past it is touched
    small = np.full(1 << 20, 0xA5, dtype=np.uint8)
    assert L.lzs_compress(small.ctypes.data, (1 << 20) - 7, x.ctypes.data, n) == (1 << 20) - 7
    assert np.array_equal(small[:(1 << 20) - 7], out[:(1 << 20) - 7]) and (small[(1 << 20) - 7:] == 0xA5).all()
    del small, tail_in
    back = np.empty(n + 64, dtype=np.uint8)
    m = L.lzs_decompress(back.ctypes.data, n + 64, out.ctypes.data, got)
    assert m == n, lzs.last_error()
    for i in range(0, n, 1 << 30):
        assert np.array_equal(back[i:i + (1 << 30)], x[i:i + (1 << 30)]), i
    # ... and an output buffer that is too small is filled to the brim (lzs-decompression.c:200)
    assert L.lzs_decompress(back.ctypes.data, (4 << 30) + 11, out.ctypes.data, got) == (4 << 30) + 11
    assert np.array_equal(back[(4 << 30) - 4096:(4 << 30) + 11], x[(4 << 30) - 4096:(4 << 30) + 11])


def test_one_long_stream_decompressed_by_many_wavefronts():
    """lzs_decompress() of a long stream is cut into 8 KiB segments, one wavefront
    each: the segments agree on the decoder state at their borders in a few rounds, decode with
    per-byte origins for copies that reach into another segment's output, and resolve those by
    pointer jumping.  Same bytes and the same stop rules as one wavefront: every class, a mixture
    with runs and repeats over many segments, a cut capacity, garbage, a stream without end marker
    and one with data after the end marker."""
    rng = np.random.default_rng(77)
    datas = []
    for cls in workload.CLASS_NAMES:
        datas.append(bytes(workload.fill(cls, 100).reshape(-1)[: 100 * 65536 - 4321]))
    mix = bytearray()
    text = bytes(workload.fill("text", 8).reshape(-1))
    while len(mix) < 9_000_000:
        k = int(rng.integers(0, 4))
        if k == 0:
            mix += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400_000))
        elif k == 1:
            a = int(rng.integers(0, len(text) - 1)); mix += text[a: a + int(rng.integers(1, 150_000))]
        elif k == 2:
            mix += bytes(rng.integers(0, 256, int(rng.integers(1, 70_000)), dtype=np.uint8))
        else:
            unit = bytes(rng.integers(0, 256, int(rng.integers(2, 2500)), dtype=np.uint8))
            mix += unit * int(rng.integers(1, 300))
    datas.append(bytes(mix))
    for d in datas:
        comp = O.compress(d)
        assert lzs.decompress(comp, len(d) + 5) == d
        assert lzs.decompress(comp, len(d) // 3) == d[: len(d) // 3]
        assert lzs.decompress(comp[: len(comp) // 2], len(d)) == O.decompress(comp[: len(comp) // 2], len(d))
        assert lzs.decompress(comp + comp[:5000], len(d) + 5) == d     # the first end marker ends it
    for _ in range(3):                                        # garbage: same result as the serial rules
        junk = bytes(rng.integers(0, 256, int(rng.integers(300_000, 900_000)), dtype=np.uint8))
        for cap in (1000, 4_000_000):
            assert lzs.decompress(junk, cap) == O.decompress(junk, cap)
    ones = bytes([0xFF]) * 400_000                            # an endless extension (offset 127)
    assert lzs.decompress(ones, 3_000_000) == O.decompress(ones, 3_000_000)


def test_short_streams_decompressed_by_many_wavefronts_in_small_segments():
    """From 4 KiB of compressed input on, lzs_decompress() spreads a stream over many wavefronts in
    segments sized to the stream (256 bytes for short ones: a 64 KiB block takes 0.7 ms instead of
    7.7).  Same bytes and stop rules as the oracle at every size: whole, cut capacity, truncated
    input, data after the end marker, garbage, all-0xFF, long runs."""
    rng = np.random.default_rng(5)
    text = bytes(workload.fill("text", 8).reshape(-1))
    rnd = bytes(workload.fill("random", 4).reshape(-1))
    low = bytes(workload.fill("lowent", 40).reshape(-1))
    datas = []
    for n in (7000, 9000, 16384, 65536, 65537, 100_000, 262_144, 500_001):
        datas.append(text[:n])
    datas += [rnd[:5000], rnd[:40_000], rnd[:200_001], low[:150_000], low[: 40 * 65536], bytes(300_000),
              text[:3000] + bytes(50_000) + text[3000:9000] + b"xy" * 30_000 + rnd[:7000]]
    walked = 0
    for d in datas:
        comp = O.compress(d)
        walked += len(comp) >= 4096
        assert lzs.decompress(comp, len(d) + 5) == d
        assert lzs.decompress(comp, len(d)) == d
        for cap in (1, len(d) // 3, len(d) - 1):
            assert lzs.decompress(comp, cap) == d[:cap]
        for cut in (len(comp) // 2, len(comp) - 1, len(comp) - 2, len(comp) - 3):
            assert lzs.decompress(comp[:cut], len(d)) == O.decompress(comp[:cut], len(d))
        assert lzs.decompress(comp + comp[:3000], len(d) + 5) == d
    assert walked >= 12
    for _ in range(12):
        junk = bytes(rng.integers(0, 256, int(rng.integers(4096, 120_000)), dtype=np.uint8))
        for cap in (777, 2_000_000):
            assert lzs.decompress(junk, cap) == O.decompress(junk, cap)
    for n in (4096, 5000, 70_000):
        ones = bytes([0xFF]) * n
        assert lzs.decompress(ones, 30 * n + 100) == O.decompress(ones, 30 * n + 100)


def test_small_batches_decompressed_in_segments():
    """lzs_decompress_batch() of a batch too small to fill the device with a wavefront per block
    cuts every block into segments for many wavefronts (4 blocks: 0.7 ms instead of 8): ragged
    blocks of every kind, empty ones, a capacity that cuts some of them, garbage and truncated
    streams in between -- block b is lzs_decompress() of block b alone, as the oracle does it."""
    rng = np.random.default_rng(21)
    text = bytes(workload.fill("text", 4).reshape(-1))
    plains = [text[:70000], b"", bytes(90000), text[100:5000], bytes(rng.integers(0, 256, 30000, dtype=np.uint8)),
              b"ab" * 20000 + text[:100], text[:1], bytes(workload.fill("lowent", 1).reshape(-1)), text[5000:66000]]
    streams = [O.compress(p) for p in plains]
    streams.append(bytes(rng.integers(0, 256, 20000, dtype=np.uint8)))           # garbage
    streams.append(streams[0][:len(streams[0]) // 2])                            # truncated
    streams.append(streams[4] + b"trailing bytes after the end marker")
    for cap in (100000, 65536, 4097, 1):
        got = _gpu_decompress_many(streams, cap)
        for i, st in enumerate(streams):
            assert got[i] == O.decompress(st, cap), (cap, i)


def test_long_match_ending_at_a_segment_border_of_all_ones():
    """A long run is thousands of 1111 nibbles; a segment that is nothing but 0xFF inside a running
    extension is not walked, its exit is worked out by the host -- which is only right if the last
    nibble that starts in it is 1111 as well, and that one reaches up to 3 bits into the next
    segment.  Runs of every length modulo a segment's worth, at every bit phase, in the smallest
    segments (found by tests/dev/fuzz_all.py: seeds 100067, 100358)."""
    import os
    os.environ["LZS_DEC_SEG"] = "256"
    try:
        bad = 0
        for phase in range(4):
            head = b"abc"[:phase] + b"\x00"
            for run in range(9000, 9000 + 7700, 7):
                d = head + b"\x90" * run + b"tail of the block"
                comp = O.compress(d)
                assert len(comp) >= 256
                os.environ["LZS_FORCE_STREAM"] = "1"
                bad += lzs.decompress(comp, len(d) + 3) != d
        assert bad == 0
    finally:
        os.environ.pop("LZS_DEC_SEG", None); os.environ.pop("LZS_FORCE_STREAM", None)


def test_concatenated_streams_decompressed_by_many_wavefronts():
    """lzs_decompress_concat() (the file rule: go on after every end marker, history kept,
    lzs-decompression.c:564-576) takes the many-wavefront route too from 4 KiB on.  Blocks
    compressed one by one and laid end to end decode to the blocks laid end to end; on anything
    else (cut capacity, truncated input, garbage with chance end markers in it) the result is the
    one wavefront's (LZS_ONE_WAVE=1, the kernel the edge vectors pin)."""
    import os
    rng = np.random.default_rng(9)
    blocks = [bytes(b) for cls in workload.CLASS_NAMES for b in workload.fill(cls, 12)]
    blocks += [b"", b"a", bytes(100_000), blocks[0][:777]]
    order = rng.permutation(len(blocks))
    plain = b"".join(blocks[i] for i in order)
    stream = b"".join(O.compress(blocks[i]) for i in order)
    assert lzs.decompress_concat(stream, len(plain) + 9) == plain

    def one_wave(data, cap):
        os.environ["LZS_ONE_WAVE"] = "1"
        try:
            return lzs.decompress_concat(data, cap)
        finally:
            del os.environ["LZS_ONE_WAVE"]

    small = stream[:300_000]
    assert one_wave(small, 2_000_000) == lzs.decompress_concat(small, 2_000_000)
    for cap in (1, 50_000, 123_457):
        assert lzs.decompress_concat(small, cap) == one_wave(small, cap)
    for _ in range(6):
        junk = bytes(rng.integers(0, 256, int(rng.integers(4096, 60_000)), dtype=np.uint8))
        assert lzs.decompress_concat(junk, 1_500_000) == one_wave(junk, 1_500_000)


VARIANTS_SO = os.path.join(os.path.dirname(os.path.abspath(lzs.__file__)), "liblzs_variants.so")


def _run_with(code, **extra_env):
    import subprocess, sys
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip()) == 64, r.stdout
    return r.stdout.strip()


@pytest.mark.parametrize("variant", ["chain", "scan", "wg8", "p256"])
def test_other_compress_kernels_agree(variant):
    """liblzs_variants.so (the library built once more with the earlier kernels, tests and A/B only)
    with LZS_KERNEL=chain|scan -- one wavefront per block / brute force over all offsets, independent
    implementations of the same rule -- gives the bytes of the product library; and so do the default kernel's
    two other shapes that kernels/compress_wg.inc takes as parameters (round 6: LZS_KERNEL=wg8, eight waves per
    workgroup with pools of 1024; p256, pools of 256) -- on the three classes, and on ragged blocks of runs, periods,
    text and noise with cut capacities (open matches, run mode, the last partial pool, capacity stops)."""
    code = (
        "import numpy as np, hashlib, random, lzs_compression_amd as lzs\n"
        "from lzs_compression_amd import workload\n"
        "h = hashlib.sha256()\n"
        "for cls in workload.CLASS_NAMES:\n"
        "    out, n = lzs.compress_batch(workload.fill(cls, 24))\n"
        "    for b in range(24): h.update(out[b, :n[b]].tobytes())\n"
        "rng = random.Random(606)\n"
        "text = workload.fill('text', 8).tobytes()\n"
        "rows = np.zeros((40, 70000), dtype=np.uint8); lens = np.zeros(40, dtype=np.uint32)\n"
        "for b in range(40):\n"
        "    d = bytearray()\n"
        "    want = rng.choice((0, 1, 2, 13, 100, 511, 512, 513, 1023, 1024, 1025, 4096, 20000, 65535, 65536, 70000))\n"
        "    while len(d) < want:\n"
        "        k = rng.randint(0, 4)\n"
        "        if k == 0: d += bytes([rng.randint(0, 255)]) * rng.randint(1, 9000)\n"
        "        elif k == 1: a = rng.randint(0, len(text) - 2); d += text[a:a + rng.randint(1, 30000)]\n"
        "        elif k == 2: d += rng.randbytes(rng.randint(1, 3000))\n"
        "        elif k == 3: u = rng.randbytes(rng.randint(1, 2500)); d += u * rng.randint(1, 30)\n"
        "        else: d += bytes(rng.choice(b'ab') for _ in range(rng.randint(1, 300)))\n"
        "    d = bytes(d[:want]); rows[b, :len(d)] = np.frombuffer(d, dtype=np.uint8); lens[b] = len(d)\n"
        "for cap in (None, 3000, 7):\n"
        "    out, n = lzs.compress_batch(rows, lens, cap)\n"
        "    h.update(n.tobytes())\n"
        "    for b in range(40): h.update(out[b, :n[b]].tobytes())\n"
        "print(h.hexdigest())\n")
    # (LZS_ONE_WAVE... is not needed: a host batch of this size is one launch of the block kernel either way)
    assert _run_with(code) == _run_with(code, LZS_LIBRARY=VARIANTS_SO, LZS_KERNEL=variant)


@pytest.mark.parametrize("decoder", ["v1", "v2"])
def test_wave_per_stream_decoders_agree_with_the_default(decoder):
    """LZS_DECODER=v1|v2 in liblzs_variants.so (one wavefront per stream, round 1) and the default
    decoder (eight streams per wavefront) give the same bytes and lengths: valid streams of every
    class, cut capacities, truncated and garbage input."""
    code = (
        "import numpy as np, hashlib, lzs_compression_amd as lzs\n"
        "from lzs_compression_amd import workload\n"
        "h = hashlib.sha256()\n"
        "rng = np.random.default_rng(5)\n"
        "for cls in workload.CLASS_NAMES:\n"
        "    out, n = lzs.compress_batch(workload.fill(cls, 23))\n"
        "    for cap in (65536, 65535, 40000, 7, 0):\n"
        "        back, m = lzs.decompress_batch(out, n, cap)\n"
        "        h.update(m.tobytes())\n"
        "        for b in range(23): h.update(back[b, :m[b]].tobytes())\n"
        "    back, m = lzs.decompress_batch(out, (n * 0.7).astype(np.uint32), 65536)\n"
        "    h.update(m.tobytes())\n"
        "    for b in range(23): h.update(back[b, :m[b]].tobytes())\n"
        "junk = rng.integers(0, 256, (37, 3000), dtype=np.uint8)\n"
        "back, m = lzs.decompress_batch(junk, rng.integers(0, 3001, 37).astype(np.uint32), 90000)\n"
        "h.update(m.tobytes())\n"
        "for b in range(37): h.update(back[b, :m[b]].tobytes())\n"
        "print(h.hexdigest())\n")
    # LZS_ONE_WAVE keeps small batches off the segment route: the block kernels are under test
    assert _run_with(code, LZS_ONE_WAVE="1") == _run_with(code, LZS_ONE_WAVE="1", LZS_LIBRARY=VARIANTS_SO, LZS_DECODER=decoder)


# ------------------------------------------------------------ BASELINE.json full-size configs
@pytest.mark.parametrize("cls", workload.CLASS_NAMES)
def test_full_size_1gib_every_block_vs_oracle_and_reference_digests(cls):
    """configs[1..3]: 1 GiB = 16384 x 64 KiB.  EVERY block's length and bytes against the CPU
    oracle ("every block compared against the CPU restatement"), every group of 1024
    blocks against the REAL reference's digests (tests/golden/class_digests_full.json, minted here
    from oracle/_ref), and, where oracle/_ref travelled, every block against the reference itself;
    plus the size-independent properties on the whole gigabyte (decode(encode(x)) == x on device,
    every length within the worst-case bound)."""
    nb, bl = 16384, 65536
    full = golden_json("class_digests_full.json")
    assert full["nblocks"] == nb and full["block_len"] == bl
    g = full["group"]
    blocks = workload.fill(cls, nb, bl, seed=full["seed"])
    x = torch.from_numpy(blocks).cuda()
    slots, lens = lzs.compress_blocks(x)
    back, back_len = lzs.decompress_blocks(slots, lens, bl)
    torch.cuda.synchronize()
    assert bool((back_len == bl).all())
    assert torch.equal(back[:, :bl], x)
    del back
    lens_h = lens.cpu().numpy()
    assert lens_h.max() <= lzs.compressed_max(bl) and lens_h.min() >= 2
    codecs = [O] + ([oracle.ref()] if oracle.have_ref() else [])
    total = 0
    for k, want in enumerate(full["classes"][cls]["groups"]):
        lo, hi = k * g, (k + 1) * g
        got, got_len = slots[lo:hi].cpu().numpy(), lens_h[lo:hi]
        assert hashlib.sha256(got_len.astype("<u4").tobytes()).hexdigest() == want["len_sha256"], (cls, k)
        h = hashlib.sha256()
        for b in range(g):
            h.update(got[b, :got_len[b]].tobytes())
        assert h.hexdigest() == want["sha256"], (cls, k)
        total += int(got_len.sum())
        for codec in codecs:
            cpu, cpu_len, _ = oracle.run_blocks(codec, blocks[lo:hi], threads=16)
            assert (cpu_len == got_len).all(), (cls, k)
            mask = np.arange(cpu.shape[1])[None, :] < cpu_len[:, None]
            assert np.array_equal(np.where(mask, got[:, :cpu.shape[1]], 0), np.where(mask, cpu, 0)), (cls, k)
    assert total == full["classes"][cls]["bytes"]


def test_packet_sized_blocks_262144_x_4kib_every_block_vs_oracle():
    """The reference's own small-block case at the batch entry (4 KiB blocks; its file tool works in
    512-byte pieces, c/src/utils/lzs-compress.c:28-32, and RFC 1974 / 2395 compress packets): 1 GiB of text as 262 144 blocks
    of 4 KiB through lzs_compress_batch_device -- one workgroup and eight pools a block -- EVERY block's length and bytes
    against the oracle, every block decoded again on the device (profiles/r06/blocksize_sweep.txt has the rates: no collapse
    at packet sizes)."""
    nb, bl = 262144, 4096
    x = workload.fill_device("text", nb, bl)
    slots, lens = lzs.compress_blocks(x)
    back, back_len = lzs.decompress_blocks(slots, lens, bl)
    torch.cuda.synchronize()
    assert bool((back_len == bl).all()) and torch.equal(back[:, :bl], x)
    del back
    lens_h = lens.cpu().numpy()
    assert lens_h.max() <= lzs.compressed_max(bl) and lens_h.min() >= 2
    threads = min(64, len(os.sched_getaffinity(0)))
    for lo in range(0, nb, 32768):
        blocks = x[lo:lo + 32768].cpu().numpy()
        got = slots[lo:lo + 32768].cpu().numpy()
        cpu, cpu_len, _ = oracle.run_blocks(O, blocks, threads=threads)
        assert (cpu_len == lens_h[lo:lo + 32768]).all(), lo
        mask = np.arange(cpu.shape[1])[None, :] < cpu_len[:, None]
        assert not ((got[:, :cpu.shape[1]] != cpu) & mask).any(), lo
    # ... and the same generator on the host gives the same blocks (what the oracle saw is what BASELINE's class is)
    assert np.array_equal(x[:64].cpu().numpy(), workload.fill("text", 64, bl))


def test_what_a_thread_keeps_goes_back_on_release_and_on_thread_exit():
    """Ownership (reference lzs.h:218,229: the library keeps nothing after return).  This build keeps the
    calling thread's staging for its next call; 32 threads that each decoded a 1 GiB stream once hold on to what that took --
    until they call lzs_release_thread_cache() (half of them) or exit (the other half): then the device's free memory is
    back to within 1 GiB of where it was."""
    import threading
    nthreads, parallel = 32, 8
    x = workload.fill_device("text", 16384).reshape(-1)
    stream, nbytes = lzs.compress_stream(x)
    stream = stream[:nbytes].clone()
    outs = [torch.empty(x.numel() + 16, dtype=torch.uint8, device="cuda") for _ in range(parallel)]
    torch.cuda.synchronize()
    lzs.release_thread_cache()                               # (this thread's own, from compress_stream)
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    slots = threading.Semaphore(parallel)
    decoded, may_go, errors = threading.Barrier(nthreads + 1), threading.Event(), []
    free_slots, lock = list(range(parallel)), threading.Lock()

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            with slots:
                with lock:
                    k = free_slots.pop()
                back, n = lzs.decompress_stream(stream, x.numel() + 16, out=outs[k])
                ok = n == x.numel() and bool(torch.equal(back[:n], x))
                with lock:
                    free_slots.append(k)
            if not ok:
                errors.append((tid, "round trip"))
            decoded.wait()                                   # everybody has decoded and still lives: what is kept is kept
            may_go.wait()
            if tid % 2 == 0:
                lzs.release_thread_cache()
        except Exception as e:      # noqa: BLE001 - collected for the assertion below
            errors.append((tid, repr(e)))
            decoded.abort()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for t in threads:
        t.start()
    decoded.wait()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                 # (torch's own cache of the comparisons' temporaries is not the library's)
    kept = free0 - torch.cuda.mem_get_info()[0]
    may_go.set()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    # (join() returns when the Python side of a thread is done; the C library's per-thread destructor -- which frees the staging of
    # the threads that just exit -- runs as the OS thread goes, a moment later: sixteen times 4 GiB to free)
    import time
    t_end = time.time() + 30.0
    while True:
        freed = free0 - torch.cuda.mem_get_info()[0]
        if freed < (1 << 30) or time.time() > t_end:
            break
        time.sleep(0.05)
    print(f"kept by {nthreads} threads: {kept >> 20} MiB; outstanding after release / exit: {freed >> 20} MiB")
    # each thread kept at least its copy of the stream and of the output (and at most LZS_KEEP_MAX_MB: 1/32 of the device)
    assert kept >= nthreads * (1 << 30), f"the threads kept {kept >> 20} MiB: is the staging not per thread any more?"
    assert kept <= nthreads * (torch.cuda.mem_get_info()[1] // 32 + (64 << 20)), f"the threads kept {kept >> 20} MiB, more than the limit allows"
    assert freed < (1 << 30), f"{freed >> 20} MiB of device memory did not come back (the threads had kept {kept >> 20} MiB)"


@pytest.mark.parametrize("cls", workload.CLASS_NAMES)
def test_other_seeds_4096_blocks_every_block_vs_oracle(cls):
    """The same three generators under two other seeds and from another block index on: 4096 blocks
    (256 MiB) each, every block's length and bytes against the oracle, decoded back on the device
    by the block decoder and, as one batch of 64, by the segment decoder."""
    for seed, first in ((workload.DEFAULT_SEED + 1, 0), (0xC0FFEE, 777777)):
        nb, bl = 4096, 65536
        blocks = workload.fill(cls, nb, bl, first_block=first, seed=seed)
        x = torch.from_numpy(blocks).cuda()
        slots, lens = lzs.compress_blocks(x)
        back, back_len = lzs.decompress_blocks(slots, lens, bl)
        torch.cuda.synchronize()
        assert bool((back_len == bl).all()) and torch.equal(back[:, :bl], x)
        got, got_len = slots.cpu().numpy(), lens.cpu().numpy()
        cpu, cpu_len, _ = oracle.run_blocks(O, blocks, threads=16)
        assert (cpu_len == got_len).all(), (cls, seed)
        mask = np.arange(cpu.shape[1])[None, :] < cpu_len[:, None]
        assert np.array_equal(np.where(mask, got[:, :cpu.shape[1]], 0), np.where(mask, cpu, 0)), (cls, seed)
        out, out_len = lzs.decompress_blocks_sync(slots[:64], got_len[:64], bl)
        assert (out_len == bl).all() and torch.equal(out[:, :bl], x[:64])


def test_device_batch_calls_capture_into_a_hip_graph():
    """lzs_batch.h says the device-pointer calls neither allocate nor synchronise and may be captured
    into a hipGraph.  Compress + compaction + decompress of 512 blocks captured once on a side stream
    (torch.cuda.graph), replayed on three different inputs written into the same buffers."""
    lzs.backend_info()                                     # (the one-time LDS ordering check happens here, not inside the capture)
    nb, bl = 512, 65536
    cap = lzs.compressed_max(bl)
    stride = (cap + 15) // 16 * 16
    x = torch.empty((nb, bl), dtype=torch.uint8, device="cuda")
    slots = torch.empty((nb, stride), dtype=torch.uint8, device="cuda")
    lens = torch.empty(nb, dtype=torch.int32, device="cuda")
    dense = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    offsets = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    back = torch.empty((nb, bl), dtype=torch.uint8, device="cuda")
    back_len = torch.empty(nb, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    x.copy_(torch.from_numpy(workload.fill("text", nb, bl)))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        lzs.compress_blocks(x, None, cap, slots, lens)     # once outside the capture (lazy module loading)
        lzs.compact(slots, lens, dense=dense, offsets=offsets)
        lzs.decompress_blocks(slots, lens, bl, back, back_len)
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            lzs.compress_blocks(x, None, cap, slots, lens)
            lzs.compact(slots, lens, dense=dense, offsets=offsets)
            lzs.decompress_blocks(slots, lens, bl, back, back_len)
    for cls, first in (("lowent", 5), ("random", 9), ("text", 4096)):
        blocks = workload.fill(cls, nb, bl, first_block=first)
        x.copy_(torch.from_numpy(blocks))
        slots.fill_(0xEE); back.fill_(0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(back, x) and bool((back_len == bl).all())
        want, want_len, _ = oracle.run_blocks(O, blocks, threads=8)
        got_len = lens.cpu().numpy()
        assert (got_len == want_len).all()
        offs = offsets.cpu().numpy()
        cat = np.concatenate([want[b, :want_len[b]] for b in range(nb)])
        assert int(offs[-1]) == cat.size and np.array_equal(dense[:cat.size].cpu().numpy(), cat)


def test_small_batch_in_device_memory_decompressed_in_segments():
    """lzs_decompress_batch_device_sync(): the segment route for batches whose buffers are already
    in HBM (lengths on the host, synchronous).  Same results as the one-launch device call."""
    for cls, nb in (("text", 7), ("lowent", 40), ("random", 3), ("text", 200)):
        x = torch.from_numpy(workload.fill(cls, nb)).cuda()
        slots, lens = lzs.compress_blocks(x)
        torch.cuda.synchronize()
        host_lens = lens.cpu().numpy()
        for cap in (65536, 30000):
            out, out_len = lzs.decompress_blocks_sync(slots, host_lens, cap)
            ref, ref_len = lzs.decompress_blocks(slots, lens, cap)
            torch.cuda.synchronize()
            assert (out_len == ref_len.cpu().numpy()).all() and (out_len == cap).all()
            assert torch.equal(out[:, :cap], ref[:, :cap]) and torch.equal(out[:, :cap], x[:, :cap])


def test_a_batch_whose_ring_does_not_fit_takes_the_serial_route_and_succeeds(monkeypatch):
    """lzs_pipeline.c promises that a batch whose device ring (or pinned pieces) cannot be reserved takes the
    one-after-the-other route instead of failing.  The failed hipMalloc stays the runtime's "last error" until it is
    fetched, and every launcher returns the last error after its launch -- so the fallback's first launch used to be
    in danger of reporting the stale out-of-memory as its own.  LZS_STAGING_FAIL_MB makes reservations
    above a size fail by a real hipMalloc that cannot succeed: 800 blocks want a ring of 2 x 512 slots (75 MB), the
    serial route 59 MB."""
    x = workload.fill("text", 800)
    cap = lzs.compressed_max(65536)
    monkeypatch.setenv("LZS_STAGING_FAIL_MB", "70")
    monkeypatch.setenv("LZS_STREAM_DEBUG", "1")               # (the pipeline reports itself on stderr when it runs)
    out = np.zeros((800, cap), dtype=np.uint8)
    out_len = np.zeros(800, dtype=np.uint32)
    rc = lzs.lib().lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, 800)
    assert rc == 0, lzs.last_error()
    monkeypatch.delenv("LZS_STAGING_FAIL_MB")
    monkeypatch.delenv("LZS_STREAM_DEBUG")
    for blk in (0, 255, 256, 511, 799):
        want = O.compress(bytes(x[blk]))
        assert int(out_len[blk]) == len(want) and out[blk, :len(want)].tobytes() == want, blk
    ref, ref_len = lzs.compress_batch(x)                      # (the overlapped route, unhindered)
    assert np.array_equal(ref_len, out_len) and all(np.array_equal(ref[b, :ref_len[b]], out[b, :out_len[b]]) for b in range(0, 800, 37))


def test_large_host_batches_take_the_overlapped_route_and_give_the_same_bytes(monkeypatch):
    """lzs_compress_batch / lzs_decompress_batch on HOST buffers from a few hundred MiB on: copy in, kernel and copy
    back run as a pipeline over chunks of blocks, host threads filling and emptying pinned pieces (lzs_pipeline.c).
    Same results as the one-after-the-other route (LZS_HOST_SERIAL=1) and as the oracle -- uniform blocks, ragged
    blocks in a strided array, a cut capacity, a last chunk of a few blocks, and from two host threads at once;
    nothing past out_len[b] is touched."""
    import threading
    nb = 2 * 640 * 2 + 37                                   # four chunks and a short fifth
    x = np.concatenate([workload.fill("text", nb - 600), workload.fill("lowent", 300), workload.fill("random", 300)])
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 65537, nb).astype(np.uint32)
    lens[::7] = 65536
    cap = lzs.compressed_max(65536)

    def run(serial, in_len_each, out_cap, fill):
        if serial:
            monkeypatch.setenv("LZS_HOST_SERIAL", "1")
        else:
            monkeypatch.delenv("LZS_HOST_SERIAL", raising=False)
        out = np.full((nb, cap + 5), fill, dtype=np.uint8)
        out_len = np.zeros(nb, dtype=np.uint32)
        rc = lzs.lib().lzs_compress_batch(out.ctypes.data, cap + 5, out_cap, out_len.ctypes.data, x.ctypes.data, 65536,
                                          in_len_each.ctypes.data if in_len_each is not None else None, 65536, nb)
        assert rc == 0, lzs.last_error()
        return out, out_len

    for in_len_each, out_cap in ((None, cap), (lens, cap), (None, 30000)):
        a, al = run(False, in_len_each, out_cap, 0xA5)
        b, bl = run(True, in_len_each, out_cap, 0xA5)
        assert np.array_equal(al, bl) and np.array_equal(a, b)                     # bytes AND the untouched fill behind them
        for blk in (0, 639, 640, 1279, 1280, nb - 601, nb - 300, nb - 1):
            n = int(in_len_each[blk]) if in_len_each is not None else 65536
            want = O.compress(bytes(x[blk, :n]))[:out_cap]
            assert int(al[blk]) == len(want) and a[blk, :len(want)].tobytes() == want, (blk, out_cap)
            assert (a[blk, len(want):] == 0xA5).all()
    # and back: ragged streams in a strided array, into exact-size slots
    comp, clen = run(False, None, cap, 0)
    back = np.full((nb, 65536 + 3), 0x5A, dtype=np.uint8)
    back_len = np.zeros(nb, dtype=np.uint32)
    rc = lzs.lib().lzs_decompress_batch(back.ctypes.data, 65536 + 3, 65536, back_len.ctypes.data, comp.ctypes.data, cap + 5,
                                        clen.ctypes.data, cap, nb)
    assert rc == 0 and (back_len == 65536).all() and np.array_equal(back[:, :65536], x) and (back[:, 65536:] == 0x5A).all()
    # garbage and truncated streams, empty blocks: whatever the decoder makes of them, both routes make the same of it
    junk = rng.integers(0, 256, (nb, 4000), dtype=np.uint8)
    junk[::3, :2000] = comp[::3, :2000]                        # (some real streams, cut)
    jl = rng.integers(0, 4001, nb).astype(np.uint32)
    jl[::11] = 0
    outs = []
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("LZS_HOST_SERIAL", "1")
        else:
            monkeypatch.delenv("LZS_HOST_SERIAL", raising=False)
        o = np.full((nb, 70000), 0x33, dtype=np.uint8)
        ol = np.zeros(nb, dtype=np.uint32)
        rc = lzs.lib().lzs_decompress_batch(o.ctypes.data, 70000, 69000, ol.ctypes.data, junk.ctypes.data, 4000, jl.ctypes.data, 4000, nb)
        assert rc == 0, lzs.last_error()
        outs.append((o, ol))
    monkeypatch.delenv("LZS_HOST_SERIAL", raising=False)
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][0], outs[1][0])
    for blk in (0, 3, 11, nb - 1):
        want = O.decompress(bytes(junk[blk, :jl[blk]]), 69000)
        assert int(outs[0][1][blk]) == len(want) and outs[0][0][blk, :len(want)].tobytes() == want, blk
    # the smallest batches that take this route (24 MiB: cut into four chunks of a hundred-odd blocks), ragged, both ways
    for small in (350, 413, 650):
        xs, ls = x[nb - 600 - 200: nb - 600 - 200 + small], lens[:small]
        res = []
        for serial in (False, True):
            if serial:
                monkeypatch.setenv("LZS_HOST_SERIAL", "1")
            else:
                monkeypatch.delenv("LZS_HOST_SERIAL", raising=False)
            so = np.full((small, cap + 1), 0x77, dtype=np.uint8)
            sl = np.zeros(small, dtype=np.uint32)
            assert lzs.lib().lzs_compress_batch(so.ctypes.data, cap + 1, cap, sl.ctypes.data, xs.ctypes.data, 65536, ls.ctypes.data, 65536, small) == 0
            sb = np.full((small, 65536), 0x11, dtype=np.uint8)
            sbl = np.zeros(small, dtype=np.uint32)
            assert lzs.lib().lzs_decompress_batch(sb.ctypes.data, 65536, 65536, sbl.ctypes.data, so.ctypes.data, cap + 1, sl.ctypes.data, cap, small) == 0
            res.append((so, sl, sb, sbl))
        monkeypatch.delenv("LZS_HOST_SERIAL", raising=False)
        assert all(np.array_equal(p_, q_) for p_, q_ in zip(res[0], res[1])), small
        assert np.array_equal(res[0][3], ls) and all(np.array_equal(res[0][2][i, :ls[i]], xs[i, :ls[i]]) for i in range(small))
        for blk in (0, small // 2, small - 1):
            want = O.compress(bytes(xs[blk, :ls[blk]]))
            assert res[0][0][blk, :res[0][1][blk]].tobytes() == want
    # two host threads at once, each with its own streams and pinned pieces
    results = [None, None]

    def work(i):
        out = np.zeros((nb, cap), dtype=np.uint8)
        out_len = np.zeros(nb, dtype=np.uint32)
        rc = lzs.lib().lzs_compress_batch(out.ctypes.data, cap, cap, out_len.ctypes.data, x.ctypes.data, 65536, None, 65536, nb)
        results[i] = (rc, out_len.copy(), out[nb - 1, :out_len[nb - 1]].tobytes())

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for rc, out_len, last in results:
        assert rc == 0 and np.array_equal(out_len, clen) and last == comp[nb - 1, :clen[nb - 1]].tobytes()
