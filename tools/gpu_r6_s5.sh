#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for c in 0 1 2; do timeout 300 tools/probes/prof_compress_counts $c; done > $OUT/prof_counts_classes.txt 2>&1
grep "ledger trips" $OUT/prof_counts_classes.txt
for c in 0 1 2; do timeout 300 tools/probes/prof_compress $c; done > $OUT/prof_compress_classes.txt 2>&1
