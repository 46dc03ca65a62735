#!/bin/bash
# round 5, session 5: the launch set by hint (disp2) against all three variants every launch (disp) and one variant forced
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  run text1 0 $H; run disp 0 $H; run disp2 0 $H
  run disp 1 $H; run disp2 1 $H; run disp2 1 "$H LZS_VARIANT=few"
  run disp 2 $H; run disp2 2 $H; run disp2 2 "$H LZS_VARIANT=lit"
done
} 2>&1 | tee $OUT/ab_s5.txt
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -8 | tee $OUT/pytest_gpu_s5.txt
timeout 600 python -u tests/dev/route_crossover.py text 2>&1 | tail -4 | tee $OUT/route_default_s5.txt
