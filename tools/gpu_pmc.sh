#!/bin/bash
# PMC passes for the compress kernel (separate runs per counter set, kernel-trace only).
TAG=${1:-pmc}
WL=${2:-text}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# one variant of the kernel per launch, the one that class's blocks get (round 5: a launch of the library runs the classifier and
# up to three variants over the same grid; the counters are the dominant kernel's)
case $WL in lowent) export LZS_VARIANT=few;; random) export LZS_VARIANT=lit;; *) export LZS_VARIANT=text;; esac
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-stream --no-config5 --no-other-classes --workload $WL > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'lzs_compress_blocks' in r['Kernel_Name']:   # (not the segment kernels of bench.py's single_stream line)
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"  {k}: per-dispatch mean {sum(v)/len(v):.6g} over {len(v)} dispatches")
PY
  else echo "no counter csv for $name"; tail -5 $OUT/$name.log; fi
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
find $OUT -size +4M -delete
head -2 /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; lscpu | grep -E "Model name|Socket|Core|Thread" | head -5
