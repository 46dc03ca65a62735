#!/bin/bash
# round 5, session 14: wave priorities as a variant option (text: on); the same for the low- and high-entropy variants, through the launcher
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
run text4 0 AB_X=1; run text5 0 AB_X=1; run text4 0 AB_NOHASH=1; run text5 0 AB_NOHASH=1
for rep in 1 2 3; do
  for v in l0 lfew llit; do run $v 0 AB_NOHASH=1; run $v 1 AB_NOHASH=1; run $v 2 AB_NOHASH=1; done
done
run l0 1 AB_X=1; run lfew 1 AB_X=1; run l0 2 AB_X=1; run llit 2 AB_X=1
run l0 0 "AB_NOHASH=1 LZS_VARIANT=few"; run l0 0 "AB_NOHASH=1 LZS_VARIANT=lit"
} 2>&1 | tee $OUT/ab_s14_priorities.txt
