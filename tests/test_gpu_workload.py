"""The device generator (lzs_gen_blocks_kernel, csrc/lzs_workload_gen.hip) produces the host
generator's bytes (csrc/lzs_workload.c) bit for bit: SURVEY.md §7 step 3 "identical in C and HIP",
§8d config 5 (the root GPU generates its 64 GiB in HBM)."""
import hashlib
import json
import os

import numpy as np
import pytest

from lzs_compression_amd import workload

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (no fallback exists)")


@pytest.mark.parametrize("cls", workload.CLASS_NAMES)
def test_device_blocks_hash_to_the_committed_input_digests(cls):
    d = json.load(open(os.path.join(GOLD, "class_digests.json")))
    x = workload.fill_device(cls, d["nblocks"], d["block_len"], seed=d["seed"])
    torch.cuda.synchronize()
    assert hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest() == d["classes"][cls]["input_sha256"]


@pytest.mark.parametrize("cls", workload.CLASS_NAMES)
@pytest.mark.parametrize("first,nblocks,block_len", [
    (0, 1, 4096), (5, 3, 65536), (16384, 130, 65536), (1048575, 2, 65536), (7, 70, 4099), (123456789012, 65, 1000), (3, 2, 7),
])
def test_device_generator_equals_host_generator(cls, first, nblocks, block_len):
    """several first_blocks (the ranks' shards of config 5 start at rank * 131072), ragged block
    lengths (odd lengths take the byte-store path), more than one wavefront of blocks"""
    want = workload.fill(cls, nblocks, block_len, first_block=first)
    got = workload.fill_device(cls, nblocks, block_len, first_block=first)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)


def test_device_generator_into_a_slice_of_a_larger_buffer():
    """bench.py fills its 8 GiB pieces in place: an unaligned destination inside a bigger tensor"""
    big = torch.zeros(3 * 65536 + 64, dtype=torch.uint8, device="cuda")
    view = big[3:3 + 2 * 65536].view(2, 65536)
    workload.fill_device("text", 2, 65536, first_block=9, out=view)
    torch.cuda.synchronize()
    assert np.array_equal(view.cpu().numpy(), workload.fill("text", 2, 65536, first_block=9))
    assert int(big[:3].sum()) == 0 and int(big[3 + 2 * 65536:].sum()) == 0
