#!/bin/bash
# PMC counters for the kernels of single-stream decompression (tests/dev/decseg_big_sweep.py, text, 4 KiB segments).
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
cd /tmp
out=$ROOT/gpurun_out/pmc_stream
rm -rf $out; mkdir -p $out
SEGS=4096 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out -- python3 $ROOT/tests/dev/decseg_big_sweep.py text > $out.log 2>&1
tail -2 $out.log
f=$(find $out -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    for key in ('lzs_scan_stream_g8_kernel', 'lzs_decode_stream_g8_kernel', 'lzs_resolve_chunks_kernel', 'lzs_resolve_tails_kernel', 'lzs_resolve_stream_kernel'):
        if key in n:
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-20s dispatches %3d  largest %.4g  mean %.4g" % (c, len(v), max(v), sum(v) / len(v)))
PY
find $out -size +2M -delete
