#!/bin/bash
# round 5, session 1: per-role phase profile of the compress kernel + first A/B set
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
for c in 0 1 2; do timeout 120 ./prof_compress $c; done 2>&1 | tee $OUT/prof_compress_classes.txt
cd $GRAFT_REPO_ROOT
{
bash tools/gpu_ab_class.sh 0 base nosafe defer early all3
bash tools/gpu_ab_class.sh 1 base early all3 le6 le6b
bash tools/gpu_ab_class.sh 2 base early all3 he1
} 2>&1 | tee $OUT/ab_s1.txt
