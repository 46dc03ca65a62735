#!/bin/bash
# usage (on the GPU box): tools/gpu_ab_class.sh CLASS variant...  -- every ab_<variant> on one class (0 text, 1 low entropy, 2 high entropy), twice
# (the second time without hashing the output: the first run's fnv says whether a variant changed the bytes)
cd $GRAFT_REPO_ROOT/tools/probes
c=$1; shift
for rep in 1 2; do for v in "$@"; do echo -n "$v: "; env $( [ $rep = 2 ] && echo AB_NOHASH=1 ) timeout 120 ./ab_$v $c; done; done
