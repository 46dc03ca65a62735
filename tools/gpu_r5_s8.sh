#!/bin/bash
# round 5, session 8: PACK in one pass per wave (pack1); the classifier that looks before it sets a bit (disp5)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  for v in text2 pack1 disp4 disp5; do run $v 0 $H; done
  run pack1 1 $H; run disp4 1 $H; run disp5 1 $H; run disp5 1 "$H LZS_VARIANT=few"
  run pack1 2 $H; run disp4 2 $H; run disp5 2 $H; run disp5 2 "$H LZS_VARIANT=lit"
done
} 2>&1 | tee $OUT/ab_s8.txt
