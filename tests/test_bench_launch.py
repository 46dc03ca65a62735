"""bench.py's launching machinery, without a GPU: `--gpus N` with no WORLD_SIZE starts its own ranks
(a child torch.distributed.run), every rank is a supervisor that runs a worker in a child process,
and a worker that fails or hangs on ANY rank sends EVERY rank to the no-collective fallback.  The GPU
workers are replaced by tests/helpers/stub_bench_worker.py (LZS_BENCH_WORKER_CMD)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
STUB = f"{sys.executable} {os.path.join(ROOT, 'tests', 'helpers', 'stub_bench_worker.py')}"


def _run(args, mode, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LZS_BENCH_ROLE", "LZS_BENCH_DIR")}
    env.update(LZS_BENCH_WORKER_CMD=STUB, LZS_STUB_MODE=mode, **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    return r, lines


def test_self_launch_relays_rank0_line():
    r, lines = _run(["--gpus", "2", "--steps", "3"], "ok")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, lines                      # ONE line on stdout, whatever the ranks printed
    doc = json.loads(lines[0])
    assert doc["from"] == "job" and doc["n_gpus"] == 2 and doc["argv"] == ["--gpus", "2", "--steps", "3"]
    assert "launching 2 ranks" in r.stderr


def test_a_failing_rank_sends_every_rank_to_the_fallback():
    r, lines = _run(["--gpus", "3"], "fail_rank1")
    assert r.returncode == 0, r.stderr[-3000:]
    doc = json.loads(lines[-1])
    assert doc["from"] == "independent" and doc["n_gpus"] == 3
    assert "rank 1: exit code 3" in doc["fallback"]["reason"]


def test_a_hanging_rank_is_ended_at_the_deadline():
    r, lines = _run(["--gpus", "2"], "hang_rank0", {"LZS_BENCH_JOB_DEADLINE": "4"})
    assert r.returncode == 0, r.stderr[-3000:]
    doc = json.loads(lines[-1])
    assert doc["from"] == "independent" and "no result after 4 s" in doc["fallback"]["reason"]


def test_forced_fallback_and_total_failure():
    r, lines = _run(["--gpus", "2"], "ok", {"LZS_BENCH_FORCE_FALLBACK": "1"})
    assert r.returncode == 0 and json.loads(lines[-1])["from"] == "independent"
    r, lines = _run(["--gpus", "2"], "fail_all")
    assert r.returncode != 0 and not lines               # nothing to report: a non-zero exit, no line


def test_world_size_must_match():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_the_launcher_never_imports_torch_or_the_library():
    """The parent of a self-launched run (and every supervisor) must stay clean of the GPU so that it
    may start other programs: with `torch` and the package made unimportable in the launcher process
    only, `bench.py --gpus 2` still launches, supervises and relays."""
    code = ("import sys, runpy; sys.modules['torch'] = None; sys.modules['lzs_compression_amd'] = None; sys.modules['numpy'] = None; "
            "sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']; runpy.run_path(%r, run_name='__main__')" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LZS_BENCH_ROLE", "LZS_BENCH_DIR")}
    env.update(LZS_BENCH_WORKER_CMD=STUB, LZS_STUB_MODE="ok")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["from"] == "job"


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # (imports nothing heavy at module level)
    return mod


def test_check_ranges_every_block_and_samples():
    """What the root compares with the CPU oracle per chunk: with --check-every 1 every block of every chunk
    of every rank (world 8, ragged last chunk), otherwise the first 1/N and the last block, never twice."""
    b = _bench_module()
    nb, cb, world = 37, 8, 8
    chunks = [(lo, min(nb, lo + cb)) for lo in range(0, nb, cb)]
    for r in range(world):
        seen = []
        for lo, hi in chunks:
            for a, z in b.check_ranges(lo, hi, 1):
                seen += list(range(r * nb + a, r * nb + z))
        assert seen == list(range(r * nb, (r + 1) * nb))
    for every in (2, 3, 16, 1000):
        for lo, hi in chunks + [(0, 1), (5, 7)]:
            got = [i for a, z in b.check_ranges(lo, hi, every) for i in range(a, z)]
            assert len(got) == len(set(got)) and got[0] == lo and got[-1] == hi - 1 and all(lo <= i < hi for i in got)
            assert len(got) <= max(2, (hi - lo + every - 1) // every + 1)


def test_run_token_names_one_launch():
    b = _bench_module()
    tok = b.run_token()
    assert str(os.getppid()) in tok and tok == b.run_token()
    os.environ["MASTER_PORT"], keep = "29555", os.environ.get("MASTER_PORT")
    try:
        assert b.run_token() != tok or keep == "29555"
    finally:
        if keep is None:
            os.environ.pop("MASTER_PORT")
        else:
            os.environ["MASTER_PORT"] = keep


def test_a_supervisor_takes_its_worker_along_on_sigterm(tmp_path):
    """torch.distributed.run ends the remaining ranks with SIGTERM when one fails: the supervisor must end
    its GPU worker (its own child) with it, not leave it on the device."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("LZS_BENCH_ROLE",)}
    env.update(LZS_BENCH_WORKER_CMD=STUB, LZS_STUB_MODE="hang_rank0", LZS_BENCH_DIR=str(tmp_path), WORLD_SIZE="1", RANK="0",
               LZS_BENCH_FORCE_SUPERVISE="1", LZS_STUB_PIDFILE=str(tmp_path / "worker.pid"))
    sup = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, cwd=ROOT,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t_end = time.time() + 60
    while not (tmp_path / "worker.pid").exists() and time.time() < t_end:
        time.sleep(0.05)
    pid = int((tmp_path / "worker.pid").read_text())
    sup.send_signal(signal.SIGTERM)
    assert sup.wait(timeout=30) != 0
    for _ in range(100):                                 # the worker is gone (reaped by the supervisor before it left)
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, signal.SIGKILL)                     # (this exact pid: the stub this test started)
        pytest.fail("the worker outlived its supervisor")


def test_the_launcher_takes_its_ranks_along_on_sigterm(tmp_path):
    """`python bench.py --gpus 2` starts its ranks as a session of their own, so a SIGTERM meant for the launcher (a
    `timeout 600 python bench.py`, the driver's limit, Ctrl-C) does not reach them by itself: the launcher must end the
    group -- agent, supervisors and the workers on the GPUs -- before it leaves."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LZS_BENCH_ROLE", "LZS_BENCH_DIR")}
    env.update(LZS_BENCH_WORKER_CMD=STUB, LZS_STUB_MODE="hang_rank0", LZS_STUB_PIDFILE=str(tmp_path / "worker.pid"))
    launcher = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t_end = time.time() + 120
    while not (tmp_path / "worker.pid.rank0").exists() and time.time() < t_end:
        time.sleep(0.05)
    pid = int((tmp_path / "worker.pid.rank0").read_text())          # the rank that hangs "on its GPU"
    launcher.send_signal(signal.SIGTERM)
    assert launcher.wait(timeout=60) == 128 + signal.SIGTERM
    for _ in range(200):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, signal.SIGKILL)                     # (this exact pid: the stub this test started)
        pytest.fail("a worker outlived the launcher")


def test_memory_plan_counts_the_roots_input_and_fits_288_gb_at_eight_ranks():
    """The root of config 5 holds every rank's generated input and one worst-case region per rank next to its own
    job buffers: at 8 ranks x 131072 blocks that is ~150 GB of the 288 -- it must be counted (the
    pre-check used to leave `pieces` out) and it must fit; a peer needs a fraction of it; and a shard that is
    halved needs about half."""
    b = _bench_module()
    sys.path.insert(0, ROOT)
    from lzs_compression_amd.sharded_job import ShardedCompressJob
    slot = (65536 + 8192 + 3 + 15) // 16 * 16
    nb, cb, world = 131072, 8192, 8
    root = b.hbm_needed(ShardedCompressJob.memory_needed(nb, 65536, slot, world, cb, True), nb, cb, world, True, slot)
    peer = b.hbm_needed(ShardedCompressJob.memory_needed(nb, 65536, slot, world, cb, False), nb, cb, world, False, slot)
    assert root > world * nb * 65536 + world * nb * slot            # the generated input AND the gathered regions
    assert 140e9 < root < 200e9 < 288e9 and peer < 40e9 and peer < root / 4
    half = b.hbm_needed(ShardedCompressJob.memory_needed(nb // 2, 65536, slot, world, cb, True), nb // 2, cb, world, True, slot)
    assert 0.45 * root < half < 0.6 * root
