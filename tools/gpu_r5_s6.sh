#!/bin/bash
# round 5, session 6: launch set by sampled hints (disp3); block decoder with four lanes a stream (sixteen streams a wavefront)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  run text1 0 $H; run disp 0 $H; run disp3 0 $H
  run disp 1 $H; run disp3 1 $H; run disp3 1 "$H LZS_VARIANT=few"
  run disp 2 $H; run disp3 2 $H; run disp3 2 "$H LZS_VARIANT=lit"
done
} 2>&1 | tee $OUT/ab_s6.txt
cd $GRAFT_REPO_ROOT
bash tools/gpu_abdec.sh "0 1 2" l8 l4 2>&1 | tee $OUT/abdec_s6_lanes.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -8 | tee $OUT/pytest_gpu_s6.txt
