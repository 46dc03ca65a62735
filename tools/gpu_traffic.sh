#!/bin/bash
# HBM traffic of the compress kernel from the PMC counters, per class: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (kernel trace only), as MI355X_MICROARCH.md prescribes, written with
# the identity of the kernel sources they were measured on to gpurun_out/pmc_traffic.json.  Copy that
# file to profiles/pmc_traffic.json: bench.py reports `roofline.traffic` from it only while the
# identity matches the tree it runs in.
# Usage (GPU box, repo root):  bash tools/gpu_traffic.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for WL in text lowent random; do
  # (one variant of the kernel per launch, the one that class's blocks get: see tools/gpu_pmc.sh)
  case $WL in lowent) export LZS_VARIANT=few;; random) export LZS_VARIANT=lit;; *) export LZS_VARIANT=text;; esac
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${WL}_$C
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${WL}_$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-stream --no-config5 --no-other-classes --workload $WL > $OUT/${WL}_$C.log 2>&1
  done
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $OUT gpurun_out/pmc_traffic.json
find $OUT -size +4M -delete
