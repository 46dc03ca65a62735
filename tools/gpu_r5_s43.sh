#!/bin/bash
# round 5, session 43: the high-entropy variant's loop shape once more at six workgroups per CU
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 40 ./ab_$1 $2; }
{
for v in l6 l6h2 l6h4 l6s2 l6s2h2; do run $v 2 AB_NOHASH=1; done
} 2>&1 | tee $OUT/ab_s43_high_entropy_shape_at_six.txt
