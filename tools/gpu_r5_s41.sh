#!/bin/bash
# round 5, session 41: six workgroups per CU for the DEFAULT variant by how the LDS is taken from the tables (text); the high-entropy variant with 512 buckets once more
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for v in lfin2 t6a t6b t6c t6d; do run $v 0 AB_X=1; done
for v in lfin2 t6a t6b t6c t6d; do run $v 0 AB_NOHASH=1; done
run lfin2 2 AB_X=1; run lfin2 1 AB_X=1
} 2>&1 | tee $OUT/ab_s41_text_six_workgroups.txt
