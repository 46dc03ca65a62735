#!/bin/bash
# round 5, session 22: the SEARCH loop's shape once more, on the kernel with priorities and the 16-bit wrap (text)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run cur 0 AB_X=1
for rep in 1 2; do for v in cur rm24 rm40 rm48 hops1 hops3 sub3; do run $v 0 AB_NOHASH=1; done; done
} 2>&1 | tee $OUT/ab_s22_loop_shape_resweep.txt
