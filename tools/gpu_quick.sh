#!/bin/bash
# Quick check of a kernel change on the 1-GPU box: parity diagnostics + three bench lines.
# Everything runs under a short timeout: a kernel that hangs must not eat the GPU budget.
cd $GRAFT_REPO_ROOT
timeout 90 python tests/dev/diag.py 2>&1 | tail -4
for w in text lowent random; do
  timeout 60 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-other-classes --workload $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
