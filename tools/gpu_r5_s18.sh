#!/bin/bash
# round 5, session 18: the counters of the final kernels -- compress per class (each class its own tag), traffic, the block decoder
cd $GRAFT_REPO_ROOT
for WL in text lowent random; do
  bash tools/gpu_pmc.sh pmc5f_$WL $WL > gpurun_out/pmc5f_$WL.txt 2>&1
  tail -30 gpurun_out/pmc5f_$WL.txt | grep "per-dispatch" | head -24
done
bash tools/gpu_traffic.sh 2>&1 | tail -5
mkdir -p gpurun_out/r05
timeout 300 python tests/dev/dectime.py 2>&1 | tail -3 | tee gpurun_out/r05/decompress_blocks_times.txt
bash tools/gpu_pmc_dec.sh 2>&1 | tee gpurun_out/r05/pmc_decompress_blocks.txt
