#!/bin/bash
# round 5, session 32: run mode's pool size and rounds once more, on the low-entropy variant (six workgroups per CU), through the launcher
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run lcur 1 AB_X=1
for rep in 1 2; do for v in lcur rp64 rp256 rr1 rr3 rr4; do run $v 1 AB_NOHASH=1; done; done
for v in lcur rp64 rp256 rr3; do run $v 0 AB_NOHASH=1; done
} 2>&1 | tee $OUT/ab_s32_run_mode_low_entropy.txt
