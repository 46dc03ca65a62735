#!/bin/bash
# round 6, session 7: the whole GPU suite on the round's tree (twice: as the tests flip switches, and as production reads them), the
# three bench lines, the kernel trace, smoke; then the counters for profiles/pmc_limiter.json / pmc_traffic.json (the kernel sources'
# stamp moved with the round's edits)
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh r06
for wl in text lowent random; do bash tools/gpu_pmc.sh pmc6_$wl $wl > gpurun_out/pmc6_$wl.txt 2>&1; done
python3 tools/pmc_limiter.py gpurun_out/pmc6 gpurun_out/r06
