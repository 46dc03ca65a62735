"""Sanitizers on the CPU builds.  Never on the GPU: the pool has no GPU
AddressSanitizer, and the reference's own latent issues (uninitialised table reads, a one-byte over-read:
c/src/liblzs/lzs-compression.c:329,349,437) are the reason to look.

* The product's UNCHANGED host sources -- lzs_host.c, lzs_stream.c, lzs_incremental.c, lzs_pipeline.c, lzs_hostcodec.c:
  per-thread staging, pinned-piece rings filled by four threads, carry state across pieces, dirty-segment re-entry, the
  small calls' own codec -- linked with tests/cpu_shim/lzs_cpu_shim.c (a CPU implementation of csrc/lzs_hip_shim.h:
  device memory is heap memory, the launches are backed by the oracle and by serial restatements of the kernels'
  contracts) and driven by tests/cpu_shim/san_driver.c through the ragged-batch, truncation, >= 24 MiB pipeline,
  one-stream-in-segments and incremental-pieces cases, under -fsanitize=address,undefined and under -fsanitize=thread.
* The checkers themselves (oracle/lzs_oracle.c, cpu_bench.c, csrc/lzs_workload.c): tests/test_oracle.py against their
  ASan + UBSan builds.
A report from a sanitizer fails the run by itself (non-zero exit, text on stderr)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SHIM = os.path.join(ROOT, "tests", "cpu_shim")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", SHIM, "all"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return os.path.join(SHIM, "_build")


def _run(exe, cases, env_extra, timeout):
    env = {k: v for k, v in os.environ.items() if not k.startswith("LZS_")}
    env.update(env_extra)
    for case in cases:
        r = subprocess.run([exe, case], capture_output=True, text=True, timeout=timeout, env=env)
        assert r.returncode == 0 and "0 failure(s)" in r.stdout, (case, r.stdout[-2000:], r.stderr[-6000:])
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (case, r.stderr[-6000:])


def test_host_sources_under_address_and_undefined_behaviour_sanitizers(built):
    # ("release": what a thread keeps between calls goes back on lzs_release_thread_cache(), on thread exit, and above
    # LZS_KEEP_MAX_MB -- the shim counts its outstanding "device" bytes; leak detection is on)
    _run(os.path.join(built, "san_asan"), ["ragged", "streams", "incremental", "pipeline", "release"],
         {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}, 900)


def test_the_harness_is_alive(built):
    """One byte written past a "device" buffer of the shim is a heap overflow the address sanitizer stops."""
    r = subprocess.run([os.path.join(built, "san_asan"), "canary"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "heap-buffer-overflow" in r.stderr and "went unnoticed" not in r.stdout, r.stderr[-2000:]


def test_host_sources_under_thread_sanitizer(built):
    """The cases with threads in them: the pipeline's four workers per batch call, two calling threads at once, the
    per-thread staging and environment (the one-stream case is single-threaded and the oracle's brute-force search behind
    the shim's segments takes 100 s under this sanitizer: the address run covers it)."""
    _run(os.path.join(built, "san_tsan"), ["ragged", "pipeline", "release"], {"TSAN_OPTIONS": "halt_on_error=1"}, 900)


def test_the_checkers_under_address_and_undefined_behaviour_sanitizers():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan.so beside this gcc")
    build = os.path.join(ROOT, "oracle", "_build")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               LZS_ORACLE_SO=os.path.join(build, "liblzs_oracle_asan.so"), LZS_WORKLOAD_SO=os.path.join(build, "liblzs_workload_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
