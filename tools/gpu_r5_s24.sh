#!/bin/bash
# round 5, session 24: the 96-bit feed of the block decoder as masks: the decoder's parity tests, a fuzz run, the times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/pytest_wide_masks.txt
timeout 500 python tests/dev/fuzz_all.py 360 9001 2>&1 | tail -2 | tee $OUT/fuzz_360s_seed9001.txt
timeout 300 python tests/dev/dectime.py 2>&1 | tail -3 | tee $OUT/decompress_blocks_times.txt
