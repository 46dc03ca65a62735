#!/bin/bash
# round 5, session 34: more quick-reject hops per full step on the high-entropy variant
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run h4 2 AB_X=1
for rep in 1 2; do for v in lcur h3 h4 h5 h6; do run $v 2 AB_NOHASH=1; done; done
} 2>&1 | tee $OUT/ab_s34_hops_high_entropy.txt
