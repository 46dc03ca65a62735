#!/bin/bash
# round 6, session 2: the profiling build's trip counts for the instruction ledger (tools/ledger.py), then the round's first kernel
# experiments side by side (tools/probes/ab.sh: b0 the tree, e1 fallback in the tail of the pass, e2 exit walk in registers, e12 both)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for c in 0 1 2; do timeout 300 tools/probes/prof_compress $c; done > $OUT/prof_compress_classes.txt 2>&1
bash tools/gpu_ab_class.sh 0 "$@" > $OUT/ab_s2.txt 2>&1
cat $OUT/ab_s2.txt; grep -A1 "ledger trips" $OUT/prof_compress_classes.txt | head -8
