#!/bin/bash
# round 5, session 16: the block decoder with its flags as masks (profiles/r05/valu_rates.txt), against the compare-and-select form
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools/probes
{
for v in ${VARIANTS:-base5 masks}; do for c in 0 1 2; do echo -n "$v: "; timeout 120 ./dec_$v $c; done; done
for v in ${VARIANTS:-base5 masks}; do echo -n "$v: "; timeout 120 ./dec_$v 0; done
} 2>&1 | tee $OUT/${TAG:-abdec_s16_masks}.txt
