#!/bin/bash
# usage (on the GPU box): tools/gpu_ab.sh variant...  -- every ab_<variant> on the three classes, twice, interleaved
cd $GRAFT_REPO_ROOT/tools/probes
for rep in 1 2; do
  for c in 0 1 2; do
    for v in "$@"; do
      echo -n "$v: "; timeout 120 ./ab_$v $c
    done
  done
done
