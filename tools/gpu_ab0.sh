#!/bin/bash
# usage (GPU box): tools/gpu_ab0.sh TAG variant...   -- the variants on the text class, twice, into gpurun_out/r06/ab_TAG.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
tag=$1; shift
bash tools/gpu_ab_class.sh 0 "$@" > $OUT/ab_$tag.txt 2>&1
cat $OUT/ab_$tag.txt
