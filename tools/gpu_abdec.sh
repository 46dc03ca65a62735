#!/bin/bash
# usage (on the GPU box): tools/gpu_abdec.sh "CLASSES" variant...  -- every probes/dec_<variant> (tools/probes/abdec.sh) on the given classes (0 text, 1 low entropy, 2 high entropy), twice
cd $GRAFT_REPO_ROOT/tools/probes
cs=$1; shift
for rep in 1 2; do for c in $cs; do for v in "$@"; do echo -n "$v: "; timeout 120 ./dec_$v $c; done; done; done
