#!/bin/bash
# round 5, session 29: the final compress kernel by optimisation level and scheduling strategy (text)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for v in cur ilp memcl o2nu o3nu; do run $v 0 AB_X=1; done
for v in cur ilp memcl o2nu o3nu; do run $v 0 AB_NOHASH=1; done
} 2>&1 | tee $OUT/ab_s29_compress_flags.txt
