"""The chain build relies on a measured gfx950 property: returning LDS atomics of ONE wave
instruction that hit the same address are applied in ascending lane order (each lane gets
the value left by the nearest lower lane).  It is not an ISA guarantee, so it is re-checked
on the device the tests run on, over 262144 conflict patterns (few/many distinct addresses,
same-bank strides, random, runs) for ds_wrxchg_rtn and ds_max_rtn."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROBE = os.path.join(ROOT, "tools", "probes", "lds_order_probe")


def test_same_address_lds_atomics_apply_in_lane_order():
    if not os.path.exists(PROBE):
        subprocess.run(["make", "-C", os.path.dirname(PROBE), PROBE], check=True)
    out = subprocess.run([PROBE], check=True, capture_output=True, text=True, timeout=300).stdout
    m = re.search(r"atomicMax lane-order violations (\d+), atomicExch violations (\d+)", out)
    assert m, out
    assert m.group(1) == "0" and m.group(2) == "0", out
