#!/bin/bash
# round 5, session 42: the three bench lines once more, on the committed final tree (the counters' stamp now covers lzs_kernels.hip)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05l; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_text.json
timeout 90 python bench.py --steps 10 --warmup 2 --workload lowent --no-config5 --no-other-classes 2>/dev/null | tail -1 > $OUT/bench_lowent.json
timeout 90 python bench.py --steps 10 --warmup 2 --workload random --no-config5 --no-other-classes 2>/dev/null | tail -1 > $OUT/bench_random.json
ls -la $OUT
