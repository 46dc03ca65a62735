import json
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# liblzs reads its development switches (LZS_FORCE_STREAM, LZS_DEC_SEG, LZS_ONE_WAVE, ...) ONCE per process; tests flip
# them between calls, which LZS_DEV_ENV -- seen at that first read -- allows (csrc/lzs_internal.h: lzs_env).
# LZS_TEST_CACHED_ENV=1 runs the suite the way a PRODUCTION process runs the library instead (the tests ran
# another path through lzs_env() than production): the environment is read once, at the first call, and every test that
# would flip a library switch is skipped (tools/gpu_round.sh runs the GPU suite a second time this way, with
# LZS_ROUTE=device fixed for the whole process, which is what tests/test_gpu_parity.py asks for test by test).
CACHED_ENV = bool(os.environ.get("LZS_TEST_CACHED_ENV"))
if CACHED_ENV:
    os.environ.pop("LZS_DEV_ENV", None)
else:
    os.environ.setdefault("LZS_DEV_ENV", "1")


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's own, except that with LZS_TEST_CACHED_ENV a test that sets or removes a switch of the library (LZS_*) to
    something the process does not already have is skipped: the library would not see the change."""
    if CACHED_ENV:
        setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

        def guarded_setenv(name, value, *a, **k):
            if name.startswith("LZS_") and os.environ.get(name) != str(value):
                pytest.skip(f"flips {name}: needs LZS_DEV_ENV (the library reads its switches once)")
            return setenv(name, value, *a, **k)

        def guarded_delenv(name, *a, **k):
            if name.startswith("LZS_") and name in os.environ:
                pytest.skip(f"flips {name}: needs LZS_DEV_ENV (the library reads its switches once)")
            return delenv(name, *a, **k)

        monkeypatch.setenv, monkeypatch.delenv = guarded_setenv, guarded_delenv
    return monkeypatch


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def golden_path(name):
    return os.path.join(GOLDEN, name)


def golden_bytes(name):
    with open(golden_path(name), "rb") as f:
        return f.read()


def golden_json(name):
    with open(golden_path(name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def edge_vectors():
    return golden_json("edge_vectors.json")


@pytest.fixture(scope="session")
def class_digests():
    return golden_json("class_digests.json")


def uncompressible_sequence():
    """22 rows x 23 letters with no repeated digram: row k, column i (1-based) is
    letter (k*i - 1) mod 23 -- the construction behind the table at
    c/src/test/test-lzs.c:44-66 (23 is prime, so every row is a permutation)."""
    return bytes(97 + (k * i - 1) % 23 for k in range(1, 23) for i in range(1, 24))


def length_bits(repeated):
    """Bit cost of the length field(s) of one match of `repeated` bytes
    (c/src/test/test-lzs.c:73-87)."""
    if repeated == 0:
        return 0
    if repeated == 1:
        return 9
    if repeated <= 4:
        return 2
    if repeated <= 7:
        return 4
    return ((repeated + 22) // 15) * 4
