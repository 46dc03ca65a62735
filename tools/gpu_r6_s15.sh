#!/bin/bash
# round 6, session 15: the incremental interface at 512-byte calls from C on the GPU box's own CPU -- this library (by size: the default
# route, device probed; and LZS_ROUTE=host) beside the compiled reference --, then twenty-five more minutes of fresh fuzz seeds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tests/dev/inc_time_c.sh 16 > $OUT/inc_time_c_gpu_box.txt 2>&1
# the product library itself (liblzs.so, default routes) under the same driver
gcc -O2 -std=c11 -Iinclude -Ilzs_compression_amd/csrc tests/dev/inc_time_c.c -Llzs_compression_amd -llzs -llzs_workload -Wl,-rpath,$PWD/lzs_compression_amd -pthread -o /tmp/inc_time_c_product 2>>$OUT/inc_time_c_gpu_box.txt
for c in 0 1 2; do echo -n "liblzs.so      "; /tmp/inc_time_c_product $c 16 | tail -1; done >> $OUT/inc_time_c_gpu_box.txt 2>&1
cat $OUT/inc_time_c_gpu_box.txt
timeout 1700 python tests/dev/fuzz_all.py 1500 18001 > $OUT/fuzz_1500s_seed18001.txt 2>&1
tail -3 $OUT/fuzz_1500s_seed18001.txt
