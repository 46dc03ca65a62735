"""File tools on the batch path and file-format compatibility with the
reference's own tools (built from /root/reference into oracle/_ref by oracle/Makefile):

* `lzs-compress -b 0` writes the same bytes as the reference's lzs-compress (one stream);
* a blocked file (independent 64 KiB streams back to back) is decoded by the REFERENCE's
  lzs-decompress, by our serial concat decoder and by our indexed parallel decoder;
* a file written by the reference's tool is decoded by ours.
"""
import os
import subprocess

import numpy as np
import pytest

import lzs_compression_amd as lzs
from lzs_compression_amd import workload

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BIN = os.path.join(ROOT, "lzs_compression_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")
have_ref_tools = os.path.exists(os.path.join(REF, "ref-lzs-decompress"))


def run(*cmd):
    r = subprocess.run(list(cmd), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (cmd, r.stdout, r.stderr)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    data = workload.fill("text", 5)[:, :].tobytes()[:300_001] + bytes(5000) + workload.fill("random", 1)[0, :7777].tobytes()
    p = d / "plain.bin"
    p.write_bytes(data)
    return d, p, data


def test_blocked_file_roundtrip_all_decoders(sample):
    d, plain, data = sample
    run(os.path.join(BIN, "lzs-compress"), "-x", str(d / "blk.idx"), str(plain), str(d / "blk.lzs"))
    comp = (d / "blk.lzs").read_bytes()
    assert len(comp) < len(data)
    # our serial concat decoder (through the C-ABI and through the tool)
    assert lzs.decompress_concat(comp, len(data) + 10) == data
    run(os.path.join(BIN, "lzs-decompress"), str(d / "blk.lzs"), str(d / "out1.bin"))
    assert (d / "out1.bin").read_bytes() == data
    # our parallel decoder with the index
    run(os.path.join(BIN, "lzs-decompress"), "-x", str(d / "blk.idx"), str(d / "blk.lzs"), str(d / "out2.bin"))
    assert (d / "out2.bin").read_bytes() == data
    # the one-shot call stops at the first end marker (reference lzs-decompression.c:255-260)
    assert lzs.decompress(comp, len(data)) == data[:65536]
    if have_ref_tools:
        run(os.path.join(REF, "ref-lzs-decompress"), str(d / "blk.lzs"), str(d / "out3.bin"))
        assert (d / "out3.bin").read_bytes() == data


@pytest.mark.skipif(not have_ref_tools, reason="reference tools not built (needs /root/reference)")
def test_single_stream_file_is_byte_identical_to_reference_tool(sample):
    d, plain, data = sample
    run(os.path.join(BIN, "lzs-compress"), "-b", "0", str(plain), str(d / "one.lzs"))
    run(os.path.join(REF, "ref-lzs-compress"), str(plain), str(d / "one_ref.lzs"))
    assert (d / "one.lzs").read_bytes() == (d / "one_ref.lzs").read_bytes()
    run(os.path.join(BIN, "lzs-decompress"), str(d / "one_ref.lzs"), str(d / "out4.bin"))
    assert (d / "out4.bin").read_bytes() == data


def test_empty_file(tmp_path):
    (tmp_path / "e").write_bytes(b"")
    run(os.path.join(BIN, "lzs-compress"), str(tmp_path / "e"), str(tmp_path / "e.lzs"))
    assert (tmp_path / "e.lzs").read_bytes() == bytes.fromhex("c000")
    run(os.path.join(BIN, "lzs-decompress"), str(tmp_path / "e.lzs"), str(tmp_path / "e.out"))
    assert (tmp_path / "e.out").read_bytes() == b""
