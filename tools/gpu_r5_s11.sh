#!/bin/bash
# round 5, session 11: the round's script on the last tree, then half an hour of the randomized cross-check
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r05c 2>&1 | tail -12
timeout 1900 python tests/dev/fuzz_all.py 1800 6001 2>&1 | tail -5 | tee $OUT/fuzz_1800s_seed6001.txt
