#!/bin/bash
# round 5, session 13: wave priorities by phase, more settings (SEARCH / the other phases / CHAIN), on all three classes through the launcher
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2 3; do
  H=$([ $rep != 1 ] && echo AB_NOHASH=1 || echo AB_X=1)
  for v in text4 p02 p03 p01 p023 p003; do run $v 0 $H; done
done
for v in text4 p02 p03 p023 p003; do
  run $v 1 "AB_NOHASH=1 AB_VIA_LAUNCHER=1"; run $v 2 "AB_NOHASH=1 AB_VIA_LAUNCHER=1"
  run $v 1 "AB_NOHASH=1 AB_VIA_LAUNCHER=1"; run $v 2 "AB_NOHASH=1 AB_VIA_LAUNCHER=1"
done
} 2>&1 | tee $OUT/ab_s13_priorities.txt
