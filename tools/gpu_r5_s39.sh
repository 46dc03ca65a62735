#!/bin/bash
# round 5, session 39: six workgroups per CU for the high-entropy variant (a small 3-byte table: its chains hold collisions anyway)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run lit640 2 AB_X=1
for rep in 1 2; do for v in lfin lit640 lit512 lit256; do run $v 2 AB_NOHASH=1; done; done
} 2>&1 | tee $OUT/ab_s39_high_entropy_six_workgroups.txt
