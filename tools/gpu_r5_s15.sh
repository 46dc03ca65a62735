#!/bin/bash
# round 5, session 15: instruction selection by issue cost (profiles/r05/valu_rates.txt): 16-bit minimum for the link ring's wrap and the length, the ring's masks in registers
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
run() { echo -n "$1 [$3]: "; env $3 timeout 60 ./ab_$1 $2; }
{
for rep in 1 2 3; do
  H=$([ $rep != 1 ] && echo AB_NOHASH=1 || echo AB_X=1)
  for v in ${VARIANTS:-text5 u16}; do run $v 0 $H; done
done
for v in ${VARIANTS:-text5 u16}; do run $v 1 AB_X=1; run $v 2 AB_X=1; done
} 2>&1 | tee $OUT/${TAG:-ab_s15_issue_costs}.txt
