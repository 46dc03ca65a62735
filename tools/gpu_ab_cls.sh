#!/bin/bash
# usage (GPU box): tools/gpu_ab_cls.sh TAG CLASS variant...   -- the variants on one class, twice, into gpurun_out/r06/ab_TAG.txt (appended)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
tag=$1; shift
bash tools/gpu_ab_class.sh "$@" >> $OUT/ab_$tag.txt 2>&1
