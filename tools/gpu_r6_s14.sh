#!/bin/bash
# round 6, session 14: the round's whole validation on the final tree (tools/gpu_r6_s7.sh: both GPU suites, the three bench lines, kernel
# trace, smoke, counters for profiles/pmc_*.json), then forty minutes of fresh fuzz seeds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_s7.sh 2>&1 | cut -c1-400
cd $GRAFT_REPO_ROOT
timeout 2600 python tests/dev/fuzz_all.py 2400 17001 > $OUT/fuzz_2400s_seed17001.txt 2>&1
tail -3 $OUT/fuzz_2400s_seed17001.txt
