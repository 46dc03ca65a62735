#!/bin/bash
# round 5, session 7: the classifier with 16-byte loads (disp4); the loop's shape and the optimisation level re-swept on round 5's kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  for v in text2 disp3 disp4 rm24 rm40 hops1 hops3 sub3 o3 o2; do run $v 0 $H; done
  run disp3 1 $H; run disp4 1 $H; run disp3 2 $H; run disp4 2 $H
done
} 2>&1 | tee $OUT/ab_s7.txt
