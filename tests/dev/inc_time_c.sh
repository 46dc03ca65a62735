#!/bin/bash
# tests/dev/inc_time_c.sh -- builds tests/dev/inc_time_c.c twice (this library's host sources over tests/cpu_shim with LZS_ROUTE=host: the
# route small pieces take on a GPU box too; and the compiled reference, oracle/_ref/liblzs_ref.so, where /root/reference was present to
# build it) and runs both on the three classes: the incremental interface at the reference tools' 512-byte calls, one core, same driver.
cd "$(dirname "$0")/../.." && B=${TMPDIR:-/tmp}/inc_time_c && mkdir -p $B && C=lzs_compression_amd/csrc
gcc -O2 -std=c11 -D_POSIX_C_SOURCE=200809L -Iinclude -I$C tests/dev/inc_time_c.c $C/lzs_host.c $C/lzs_stream.c $C/lzs_incremental.c $C/lzs_pipeline.c \
    $C/lzs_hostcodec.c $C/lzs_workload.c tests/cpu_shim/lzs_cpu_shim.c oracle/lzs_oracle.c -pthread -o $B/ours || exit 1
[ -f oracle/_ref/liblzs_ref.so ] && gcc -O2 -std=c11 -Iinclude -I$C tests/dev/inc_time_c.c $C/lzs_workload.c -Loracle/_ref -llzs_ref -Wl,-rpath,$PWD/oracle/_ref -pthread -o $B/ref
echo "# $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | xargs), one core; last of three passes"
for c in 0 1 2; do
  echo -n "this library   "; LZS_ROUTE=host $B/ours $c ${1:-16} | tail -1
  [ -x $B/ref ] && { echo -n "the reference  "; $B/ref $c ${1:-16} | tail -1; }
done
