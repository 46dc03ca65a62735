#!/bin/bash
# round 5, session 10: PMC passes for the other two classes (a tag each: the passes of one tag share directories), the suite's summary lines, a fuzz run
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for WL in lowent random; do bash tools/gpu_pmc.sh pmc5_$WL $WL 2>&1 | tee $OUT/pmc5_$WL.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -5 | tee $OUT/pytest_gpu_s10.txt
LZS_TEST_CACHED_ENV=1 LZS_ROUTE=device timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -5 | tee $OUT/pytest_gpu_cached_env_s10.txt
timeout 700 python tests/dev/fuzz_all.py 600 5001 2>&1 | tail -5 | tee $OUT/fuzz_600s_seed5001.txt
