"""The N > 1 job on the real backend: torch.distributed "nccl" (= RCCL) with world size 1 on the one
GPU of the test box -- process-group set-up, the all_gathers and the job's control flow run on the
device; the point-to-point legs need a second GPU and are covered under gloo (tests/test_sharding.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (no fallback exists)")


def test_bench_sharded_job_world1_nccl():
    """bench.py's N > 1 leg as the driver would start it, forced onto one GPU: device generator,
    scatter (no peers), compress, compact, gather-v, the on-device round trip and the oracle check
    of sampled gathered blocks."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded-job", "--blocks", "2048",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2
    assert line["checks"] == {"every_rank_round_trip_on_device": True, "gathered_samples_equal_oracle": True}
    assert set(line["phases_ms"]) >= {"scatter", "compress", "gather"}
    assert line["gathered_bytes"] > 0 and 0.5 < line["config"]["compression_ratio"] < 0.62
    assert line["value"] > 0 and line["compute_only_GBps"] >= line["end_to_end_GBps"]


def test_scatter_and_gather_primitives_world1_nccl():
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import lzs_compression_amd as lzs
from lzs_compression_amd import sharding, workload
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
x = workload.fill_device("lowent", 300, 65536)
mine = sharding.scatter_blocks(x, 300, 65536, dev)
assert torch.equal(mine, x)
slots, lens = lzs.compress_blocks(mine)
dense, offsets = lzs.compact(slots, lens)
n = int(offsets[-1].item())
got, counts = sharding.gather_streams(dense, n)
all_lens = sharding.gather_lengths(lens)
torch.cuda.synchronize()
assert counts == [n] and got.numel() == n and torch.equal(got, dense[:n]) and torch.equal(all_lens, lens)
assert sharding.gather_counts(n, dev) == [n]
dist.destroy_process_group()
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout.split(), r.stderr[-2000:]      # (RCCL prints its version banner after it)
