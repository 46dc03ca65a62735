#!/bin/bash
# round 5, session 4: after the idle-lane fix: the whole GPU suite; stream decode stages with and without kept buffers; crossovers
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -8 | tee $OUT/pytest_gpu_s4.txt
for k in "" 16384; do
  echo "== LZS_KEEP_MAX_MB=$k" | tee -a $OUT/stream_dec_stages_s4.txt
  for i in 1 2; do LZS_KEEP_MAX_MB=$k timeout 300 python tests/dev/stream_dec_stages.py text 2>&1 | grep "bytes decoded\|round trip" | tee -a $OUT/stream_dec_stages_s4.txt; done
done
timeout 900 python -u tests/dev/route_crossover.py text 2>&1 | tee $OUT/route_crossover_text.txt
