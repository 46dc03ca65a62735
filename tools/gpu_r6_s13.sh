#!/bin/bash
# round 6, session 13: one-byte exits as a parameter of the default variant only (LZS_WGV_EXIT8) -- the three classes against the commit
# before them through the library's launcher, then the round's whole validation (tools/gpu_r6_s7.sh) and ten minutes of fresh fuzz seeds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for c in 2 1 0; do bash tools/gpu_ab_cls.sh s50 $c lhead ltree2; done
cat $OUT/ab_s50.txt
bash tools/gpu_r6_s7.sh 2>&1 | cut -c1-400
cd $GRAFT_REPO_ROOT
timeout 800 python tests/dev/fuzz_all.py 600 16001 > $OUT/fuzz_600s_seed16001.txt 2>&1
tail -3 $OUT/fuzz_600s_seed16001.txt
