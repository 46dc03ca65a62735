#!/bin/bash
# round 5, session 40: the high-entropy variant at six workgroups per CU by the size of its 3-byte table
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for v in lfin lit512 lit384 lit448 lit480 lit544 lit576; do run $v 2 AB_X=1; done
for v in lfin lit512 lit448 lit576; do run $v 2 AB_NOHASH=1; done
} 2>&1 | tee $OUT/ab_s40_high_entropy_table_size.txt
