#!/bin/bash
# round 5, session 36: quick-reject hops on the low-entropy variant
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run lcur 1 AB_X=1
for rep in 1 2; do for v in lcur h0 h1 h3 h4; do run $v 1 AB_NOHASH=1; done; done
} 2>&1 | tee $OUT/ab_s36_hops_low_entropy.txt
