#!/bin/bash
# round 5, session 2: the variants dispatched per block, the deferred result store; the variant tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  for v in base nosafe text1 disp defer; do run $v 0 $H; done
  for v in base disp; do run $v 1 $H; done
  run disp 1 "$H LZS_VARIANT=few"; run disp 1 "$H LZS_VARIANT=text"
  for v in base disp; do run $v 2 $H; done
  run disp 2 "$H LZS_VARIANT=lit"; run disp 2 "$H LZS_VARIANT=text"
done
} 2>&1 | tee $OUT/ab_s2.txt
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variant or classes_get or class_digests or golden_vector or tiny or ring_does_not_fit" 2>&1 | tail -5 | tee $OUT/pytest_s2.txt
timeout 600 python -m pytest tests/test_routes.py tests/test_gpu_dropin.py tests/test_gpu_incremental.py -x -q 2>&1 | tail -5 | tee -a $OUT/pytest_s2.txt
