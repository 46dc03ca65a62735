#!/bin/bash
# round 5, session 3: what is slow in stream decompression; the CLI single-stream test; route crossovers; few more A/Bs
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tests/dev/stream_dec_stages.py text 2>&1 | tail -40 | tee $OUT/stream_dec_stages_s3.txt
timeout 300 python -m pytest tests/test_gpu_cli.py -x -q --timeout 120 2>&1 | tail -15 | tee $OUT/pytest_cli_s3.txt
timeout 600 python tests/dev/route_crossover.py text 2>&1 | tee $OUT/route_crossover_text.txt
cd tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2; do
  H=$([ $rep = 2 ] && echo AB_NOHASH=1 || echo AB_X=1)
  run text1 0 $H; run disp 0 $H; run disp 0 "$H LZS_VARIANT=few"; run disp 0 "$H LZS_VARIANT=lit"
  run disp 1 $H; run disp 1 "$H LZS_VARIANT=few"; run disp 2 $H; run disp 2 "$H LZS_VARIANT=lit"
done
} 2>&1 | tee $OUT/ab_s3.txt
