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


def _bench(args, timeout=900, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LZS_BENCH_ROLE", "LZS_BENCH_DIR")}
    env.update(MASTER_ADDR="127.0.0.1", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r


@pytest.mark.parametrize("extra", [["--chunk-blocks", "512"], ["--chunk-blocks", "600"], ["--no-overlap"]])
def test_bench_sharded_job_world1_nccl(extra):
    """bench.py's N > 1 leg as the driver would start it, forced onto one GPU: device generator,
    the pipelined job (four chunks; ragged chunks; un-overlapped) with its all_gathers on RCCL, the
    un-overlapped pass for the phase times, the on-device round trip and the oracle check of the
    gathered streams."""
    line, _ = _bench(["--gpus", "1", "--sharded-job", "--blocks", "2048", "--steps", "2", "--warmup", "1", "--check-every", "1"] + extra,
                     MASTER_PORT="29533")
    assert line["n_gpus"] == 1 and line["steps"] == 2
    assert line["checks"]["every_rank_round_trip_on_device"] is True and line["checks"]["gathered_samples_equal_oracle"] is True
    assert line["checks"]["gathered_blocks_compared_with_oracle"] == 2048
    assert line["overlap"] == (extra[0] != "--no-overlap") and line["chunks_per_rank"] == {"512": 4, "600": 4}.get(extra[-1], 1)
    assert set(line["phases_ms"]) >= {"scatter", "compress", "gather", "step"}
    assert line["gathered_bytes"] > 0 and 0.5 < line["config"]["compression_ratio"] < 0.62
    assert line["value"] > 0 and line["compute_only_GBps"] > 0 and line["serial_end_to_end_GBps"] > 0
    assert len(line["overlapped_step"]["stage_compute_ms_rank0_last_step"]) == line["chunks_per_rank"]
    assert "fallback" not in line
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1      # (north_star: next to every N's figure)
    assert line["config"]["memory_plan"] == "as asked"


def test_bench_self_launch_two_ranks_on_one_gpu_falls_back():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py launches its own ranks (child
    torch.distributed.run), each rank a supervisor with a worker child.  Two ranks on ONE GPU: RCCL
    refuses a communicator with a duplicate device, so the first attempt fails for real and every rank
    takes the no-collective fallback -- the path an 8-GPU lease would take if the RCCL job broke."""
    line, r = _bench(["--gpus", "2", "--allow-shared-gpu", "--blocks", "1024", "--steps", "2", "--warmup", "1"],
                     LZS_BENCH_JOB_DEADLINE="240")
    assert line["n_gpus"] == 2 and "fallback" in line and "failed" in line["fallback"]["reason"], r.stderr[-2000:]
    assert line["checks"] == {"every_rank_round_trip_on_device": True, "sampled_blocks_equal_oracle": True}
    # the fallback line carries NO value (a compute-only figure must not be read as config 5's rate)
    assert line["value"] is None and line["valid"] is False and line["compute_only_GBps"] > 0 and len(line["per_rank_elapsed_s"]) == 2
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] in ("reference", "port")


def test_bench_default_line_checks_every_block_and_carries_the_config5_fields():
    """The driver's N = 1 command (smaller batch): the contract line, the every-block comparison with
    the host codec, and the config-5 job at world size 1 with the fields of the N > 1 lines."""
    line, _ = _bench(["--gpus", "1", "--blocks", "2048", "--steps", "3", "--warmup", "1", "--no-single-stream"])
    assert line["n_gpus"] == 1 and line["unit"] == "GB/s" and line["roofline"]["bound"] == "hbm"
    cb = line["cpu_baseline"]
    assert cb["gpu_output_bit_exact_on_sample"] is True and cb["check"]["blocks_compared"] == cb["check"]["of"] == 2048
    c5 = line["config5_world1"]
    assert "error" not in c5, c5
    assert c5["checks"]["every_rank_round_trip_on_device"] and c5["checks"]["gathered_samples_equal_oracle"]
    assert c5["overlap"] is False or c5["chunks_per_rank"] >= 1
    # BASELINE.json configs[2] and [3] on the same line, measured and checked like the headline
    oc = line["other_classes"]
    assert set(oc) == {"lowent", "random"}
    for cls, rec in oc.items():
        assert "error" not in rec, rec
        assert rec["value"] > 0 and rec["launches"] == 10 and rec["check"]["bit_exact"] is True and rec["check"]["blocks_compared"] == 2048, (cls, rec)
        assert rec["roofline"]["frac"] == rec["roofline"]["achieved"] / 8000.0
    assert oc["lowent"]["compression_ratio"] < 0.06 and 1.10 < oc["random"]["compression_ratio"] < 1.13
    assert "limiter" in line["roofline"]


def test_bench_under_torch_distributed_run_with_one_rank():
    """The driver's launch form at N = 1, should it use it: `torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1`.  The config-5 extra makes a rendezvous of its own there (the environment
    points at the agent's store, which would be waited for until the timeout: it once took the bench
    from 9 s to 4 minutes)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LZS_BENCH_ROLE", "LZS_BENCH_DIR")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "2048", "--steps", "2",
                        "--warmup", "1", "--no-single-stream", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=150, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "error" not in line["config5_world1"], line["config5_world1"]
    assert line["config5_world1"]["checks"]["gathered_samples_equal_oracle"] is True


def test_supervisor_worker_rendezvous_chain_with_one_rank():
    """What every rank of an N > 1 run does, with N = 1: torch.distributed.run starts bench.py, which
    supervises a WORKER child; the worker joins the process group through the agent's store
    (env://, as a grandchild of the agent) and runs the pipelined job on RCCL.  No fallback line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LZS_BENCH_ROLE", "LZS_BENCH_DIR")}
    env["LZS_BENCH_FORCE_SUPERVISE"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "2048", "--chunk-blocks", "512",
                        "--steps", "2", "--warmup", "1", "--check-every", "4"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                        # the supervisor relays ONE line
    line = json.loads(lines[0])
    assert "fallback" not in line and line["overlap"] is True and line["chunks_per_rank"] == 4
    assert line["checks"]["every_rank_round_trip_on_device"] and line["checks"]["gathered_samples_equal_oracle"]


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


def test_the_c_entries_of_the_sharded_job_at_world_one():
    """include/lzs/lzs_shard.h from a C host's point of view (ctypes here): lzs_shard_range, lzs_rccl_scatter_blocks,
    the batch call, lzs_compact_device, lzs_rccl_gather_streams -- at world size 1 on the one GPU of the test box, with a
    real one-rank communicator (ncclCommInitAll), so that the library's run-time binding of librccl and its
    ncclAllGather of the byte counts are exercised on hardware.  The gathered bytes are the oracle's streams back to back."""
    import ctypes
    import oracle
    import lzs_compression_amd as lzs
    from lzs_compression_amd import workload
    L = lzs.lib()
    vp, sz, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)
    L.lzs_rccl_scatter_blocks.restype = ctypes.c_int
    L.lzs_rccl_scatter_blocks.argtypes = [vp, vp, vp, sz, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.lzs_rccl_gather_streams.restype = ctypes.c_int
    L.lzs_rccl_gather_streams.argtypes = [vp, vp, u64p, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = ctypes.CDLL(name)
            break
        except OSError:
            continue
    assert rccl is not None, "librccl is not on this box"
    comm = ctypes.c_void_p()
    dev0 = (ctypes.c_int * 1)(0)
    assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, dev0) == 0
    try:
        nb = 96
        host = workload.fill("text", nb, 65536)
        everything = torch.from_numpy(host).cuda()
        mine = torch.empty_like(everything)
        stream = torch.cuda.current_stream().cuda_stream
        assert L.lzs_rccl_scatter_blocks(comm, mine.data_ptr(), everything.data_ptr(), nb, 65536, 0, 1, 0, stream) == 0, lzs.last_error()
        slots, lens = lzs.compress_blocks(mine)
        dense, offs = lzs.compact(slots, lens)
        d_counts = torch.zeros(1, dtype=torch.int64, device="cuda")
        counts = (ctypes.c_uint64 * 1)()
        out = torch.zeros(int(dense.numel()), dtype=torch.uint8, device="cuda")
        rc = L.lzs_rccl_gather_streams(comm, out.data_ptr(), counts, d_counts.data_ptr(), dense.data_ptr(),
                                       offs.data_ptr() + 8 * nb, 0, 1, 0, stream)
        assert rc == 0, lzs.last_error()
        torch.cuda.synchronize()
        want = b"".join(oracle.oracle().compress(host[b].tobytes()) for b in range(nb))
        assert counts[0] == len(want) and out[:counts[0]].cpu().numpy().tobytes() == want
    finally:
        rccl.ncclCommDestroy(comm)
