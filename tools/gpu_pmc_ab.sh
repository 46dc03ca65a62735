#!/bin/bash
# PMC counters for ab_bench variants: tools/gpu_pmc_ab.sh variant...   (one rocprofv3 pass each)
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for v in "$@"; do
  out=$ROOT/gpurun_out/pmc_ab_$v
  rm -rf $out; mkdir -p $out
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out -- $ROOT/tools/probes/ab_$v 0 > $out.log 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'lzs_compress' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("  " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(agg.items())))
PY
  find $out -size +2M -delete
done
