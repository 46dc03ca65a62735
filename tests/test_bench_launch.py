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
