#!/bin/bash
# PMC counters for the decompress kernel (tests/dev/dectime.py, all three classes in one process).
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
out=$ROOT/gpurun_out/pmc_dec
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out -- python3 $ROOT/tests/dev/dectime.py > $out.log 2>&1
tail -3 $out.log
f=$(find $out -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'lzs_decompress' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, ["%.3g" % x for x in v[::6]], "(dispatches: text x6, lowent x6, random x6)")
PY
find $out -size +2M -delete
