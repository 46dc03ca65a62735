"""Bounded run of the randomized cross-check (tests/dev/fuzz_all.py: every host-visible path --
one-shot calls with random segment sizes, cut capacities, truncated and garbage streams,
concatenated streams, the incremental interface in random pieces, device stream calls at odd
alignments, ragged batches -- against the oracle and, where it was built, the compiled reference).

The seed list starts with every seed that ever failed during development (DESIGN.md section 7 says
which commit fixed which) and goes on with fresh ones until the time is up."""
import importlib.util
import os
import time

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
HERE = os.path.dirname(os.path.abspath(__file__))

# found by the fuzzer in round 1: 100067, 100358 (all-0xFF segment shortcut of the stream decoder,
# fixed in 447e0bd); 500000, 700000, 702200 (device stream wrappers vs the caller's torch stream and
# output clearing, fixed in ed8ad5f)
REGRESSION_SEEDS = [100067, 100358, 500000, 700000, 702200]


def test_fuzz_regression_seeds_then_fresh_ones_for_a_minute():
    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (no fallback exists)")
    spec = importlib.util.spec_from_file_location("fuzz_all", os.path.join(HERE, "dev", "fuzz_all.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    t0, done = time.time(), 0
    for seed in REGRESSION_SEEDS:
        fz.check_seed(seed)
        done += 1
    seed = 2_000_000
    while time.time() - t0 < 60.0:
        fz.check_seed(seed)
        seed += 1
        done += 1
    assert done >= len(REGRESSION_SEEDS)
