#!/bin/bash
# usage: tools/probes/abdec.sh name=file.hip[:extra -D flags] ...   builds one ab_dec binary (dec_<name>) per variant
cd "$(dirname "$0")"
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; file=${rest%%:*}; flags=""; [ "$rest" != "$file" ] && flags=${rest#*:}
  /opt/rocm/bin/hipcc ${ABOPT:--Os -fno-unroll-loops} --offload-arch=gfx950 -w -I../../lzs_compression_amd/csrc $flags -DKSRC="\"$file\"" ab_dec.hip -o dec_$name -L../../lzs_compression_amd -llzs_workload -Wl,-rpath,'$ORIGIN/../../lzs_compression_amd' &
done
wait
