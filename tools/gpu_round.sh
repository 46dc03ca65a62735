#!/bin/bash
# One GPU-box session: parity tests, bench lines for the three classes, rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -5 | tee $OUT/pytest_gpu.txt
# ... and once more the way a production process runs the library: the environment read once (no LZS_DEV_ENV), every call on the
# device; the tests that flip a switch of the library skip themselves (tests/conftest.py)
LZS_TEST_CACHED_ENV=1 LZS_ROUTE=device timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -5 | tee $OUT/pytest_gpu_cached_env.txt
timeout 600 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 | tee $OUT/bench_text.json
timeout 300 python bench.py --steps 10 --warmup 2 --workload lowent --no-config5 --no-other-classes 2>/dev/null | tail -1 | tee $OUT/bench_lowent.json
timeout 300 python bench.py --steps 10 --warmup 2 --workload random --no-config5 --no-other-classes 2>/dev/null | tail -1 | tee $OUT/bench_random.json
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5 --no-other-classes > $OUT/rocprof_bench.log 2>&1
find $OUT/rocprof_stats -name '*stats*.csv' | head -5
for f in $(find $OUT/rocprof_stats -name '*kernel_stats.csv'); do head -8 $f; done
# the driver's acceptance hook, LAST: whatever changed during the session, smoke() ran after it
cd $GRAFT_REPO_ROOT
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3 | tee $OUT/smoke.txt
# keep the trace itself small: drop the per-dispatch csv if huge
find $OUT -size +8M -delete
