#!/bin/bash
# round 5, session 35 (and 38): three hops in the high-entropy variant, none in the low-entropy one -- the counters of the final compress kernels again, then the round's session
cd $GRAFT_REPO_ROOT
for WL in text lowent random; do
  bash tools/gpu_pmc.sh pmc5i_$WL $WL > gpurun_out/pmc5i_$WL.txt 2>&1
done
bash tools/gpu_traffic.sh 2>&1 | tail -2
python3 tools/pmc_limiter.py gpurun_out/pmc5i 2>&1 | tail -4
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
bash tools/gpu_round.sh r05k
