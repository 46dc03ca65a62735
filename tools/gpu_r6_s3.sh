#!/bin/bash
# round 6, session 3: the product kernel's own trip counts (counts-only profiling build), the pool-of-256 / six-workgroup variants
# (VERDICT r05 item 1c), the aligned candidate reads (item 2) with their LDS counters
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 300 tools/probes/prof_compress_counts 0 > $OUT/prof_counts_text.txt 2>&1
grep -m1 -A0 "ledger trips" $OUT/prof_counts_text.txt; grep "ledger trips" $OUT/prof_counts_text.txt | tail -1
bash tools/gpu_ab_class.sh 0 "$@" > $OUT/ab_s3.txt 2>&1
cat $OUT/ab_s3.txt
bash tools/gpu_pmc_ab.sh b0 a128 a64 p256h1536 > $OUT/pmc_ab_s3.txt 2>&1
cat $OUT/pmc_ab_s3.txt
