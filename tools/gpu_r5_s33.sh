#!/bin/bash
# round 5, session 33: the SEARCH loop's shape on the high-entropy variant (one full step per pass), through the launcher
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run lcur 2 AB_X=1
for rep in 1 2; do for v in lcur h0 h1 h3 r24 r40; do run $v 2 AB_NOHASH=1; done; done
} 2>&1 | tee $OUT/ab_s33_loop_shape_high_entropy.txt
