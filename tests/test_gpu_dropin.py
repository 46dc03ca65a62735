"""Drop-in proof: the REFERENCE's own compression unit test (c/src/test/test-lzs.c, built by
oracle/Makefile against OUR header and linked with OUR liblzs.so -> oracle/_ref/dropin-test-lzs)
passes: 507 incompressible prefixes and 1001 repeated-byte lengths, exact compressed
sizes and round trips, all through the 4-argument lzs_compress()/lzs_decompress() -- on the GPU
(LZS_ROUTE=device: every call a launch), on the small calls' host route (LZS_ROUTE=host, which needs no
device and runs in the CPU suite too), and by size as a caller gets it (the default)."""
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
EXE = os.path.join(ROOT, "oracle", "_ref", "dropin-test-lzs")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/dropin-test-lzs was not built (needs /root/reference)")
@pytest.mark.parametrize("route", [pytest.param("device", marks=pytest.mark.gpu), pytest.param("", marks=pytest.mark.gpu), "host"])
def test_reference_unit_test_passes_against_our_library(route):
    env = {k: v for k, v in os.environ.items() if k != "LZS_ROUTE"}
    if route:
        env["LZS_ROUTE"] = route
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "2 Tests 0 Failures 0 Ignored" in r.stdout, r.stdout
