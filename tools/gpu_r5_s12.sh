#!/bin/bash
# round 5, session 12: wave priorities by phase (experiment); the incremental interface by piece size; the incremental suite with the route by size
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2 3; do
  H=$([ $rep != 1 ] && echo AB_NOHASH=1 || echo AB_X=1)
  for v in base text4 prio1 prio2; do run $v 0 $H; done
done
run prio1 1 AB_X=1; run prio1 2 AB_X=1; run prio2 1 AB_X=1; run prio2 2 AB_X=1
} 2>&1 | tee $OUT/ab_s12_priorities.txt
cd $GRAFT_REPO_ROOT
timeout 900 python tests/dev/inc_time.py 2>&1 | tail -8 | tee $OUT/incremental_times.txt
timeout 900 python -m pytest tests/test_gpu_incremental.py tests/test_routes.py -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/pytest_incremental_s12.txt
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2 | tee $OUT/smoke_s12.txt
