#!/bin/bash
# round 5, session 27: the final block decoder by optimisation level; its counters
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
VARIANTS="fin o3 o2 o3nu" TAG=abdec_s27_optimisation_levels bash tools/gpu_r5_s16.sh
bash tools/gpu_pmc_dec.sh 2>&1 | grep -v "^[EW]2026" | tee $OUT/pmc_decompress_blocks.txt
