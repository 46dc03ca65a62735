#!/bin/bash
# usage: tools/probes/ab.sh name=file.hip[:extra -D flags] ...   builds one ab_bench binary per variant
# (optimisation level as the library's Makefile; ABOPT=-O3 overrides)
cd "$(dirname "$0")"
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; file=${rest%%:*}; flags=""; [ "$rest" != "$file" ] && flags=${rest#*:}
  /opt/rocm/bin/hipcc ${ABOPT:--Os -fno-unroll-loops} --offload-arch=gfx950 -w -I../../lzs_compression_amd/csrc $flags -DKSRC="\"$file\"" ab_bench.hip -o ab_$name -L../../lzs_compression_amd -llzs_workload -Wl,-rpath,'$ORIGIN/../../lzs_compression_amd' &
done
wait
