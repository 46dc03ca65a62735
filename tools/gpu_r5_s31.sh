#!/bin/bash
# round 5, session 31: instruction-cache and instruction-fetch counters of the text compress kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
export TMPDIR=/tmp LZS_VARIANT=text
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_WAIT_IFETCH|SQC_.*(REQ|MISS|HIT)" | head -30 > $OUT/icache_counters_available.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_IFETCH_LEVEL"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/ic_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-stream --no-config5 --no-other-classes --workload text > $OUT/ic_$name.log 2>&1
  f=$(find $OUT/ic_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'lzs_compress_blocks' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items(): print(f"  {k}: per-dispatch mean {sum(v)/len(v):.6g} over {len(v)} dispatches")
PY
  [ -z "$f" ] && tail -3 $OUT/ic_$name.log
done 2>&1 | tee $OUT/icache_text.txt
find $OUT -size +4M -delete
