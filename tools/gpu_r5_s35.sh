#!/bin/bash
# round 5, session 35: three hops in the high-entropy variant -- the counters of the final compress kernels again, then the round's session
cd $GRAFT_REPO_ROOT
for WL in text lowent random; do
  bash tools/gpu_pmc.sh pmc5g_$WL $WL > gpurun_out/pmc5g_$WL.txt 2>&1
done
bash tools/gpu_traffic.sh 2>&1 | tail -2
python3 tools/pmc_limiter.py gpurun_out/pmc5g 2>&1 | tail -4
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
bash tools/gpu_round.sh r05i
