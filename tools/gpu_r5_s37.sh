#!/bin/bash
# round 5, session 37: the low-entropy variant without quick-reject hops; its loop's shape around that
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run lnew 1 AB_X=1
for rep in 1 2; do for v in lcur lnew lnews1 lnews3 lnewr24 lnewr48; do run $v 1 AB_NOHASH=1; done; done
run lnew 0 AB_X=1; run lnew 2 AB_X=1
} 2>&1 | tee $OUT/ab_s37_low_entropy_without_hops.txt
