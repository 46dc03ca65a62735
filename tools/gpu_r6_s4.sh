#!/bin/bash
# round 6, session 4: SQ_INSTS_* of the kernel with a phase switched off (wrong output on purpose), to check the ledger's shares
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc_ab.sh "$@" > $OUT/pmc_ab_s4.txt 2>&1
cat $OUT/pmc_ab_s4.txt
