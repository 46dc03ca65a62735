#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/probes
for rep in 1 2; do for v in "$@"; do echo -n "$v: "; timeout 120 ./ab_$v 0; done; done
