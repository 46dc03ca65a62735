#!/bin/bash
# round 5, session 9: the round's script on the current tree (tests twice, bench x3, rocprof stats, smoke), A/B of the final kernels, PMC passes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r05b 2>&1 | tail -30
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  run base 0 $H; run text3 0 $H; run disp6 0 $H
  run base 1 $H; run disp6 1 $H; run disp6 1 "$H LZS_VARIANT=few"
  run base 2 $H; run disp6 2 $H; run disp6 2 "$H LZS_VARIANT=lit"
done
} 2>&1 | tee $OUT/ab_s9.txt
cd $GRAFT_REPO_ROOT
for WL in text lowent random; do bash tools/gpu_pmc.sh pmc5 $WL 2>&1 | tee $OUT/pmc5_$WL.txt; done
bash tools/gpu_traffic.sh 2>&1 | tail -5
