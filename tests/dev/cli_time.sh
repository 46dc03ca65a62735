#!/bin/bash
# dev aid: time the file tools on a 256 MiB text file (plain stream, no index)
cd "$(dirname "$0")/../.."
python - <<'PY'
from lzs_compression_amd import workload
open("/tmp/t.bin","wb").write(workload.fill("text", 4096).tobytes())
PY
B=lzs_compression_amd/bin
time $B/lzs-compress -b 0 /tmp/t.bin /tmp/t.lzs
ls -l /tmp/t.bin /tmp/t.lzs
time $B/lzs-decompress /tmp/t.lzs /tmp/t.back
cmp /tmp/t.bin /tmp/t.back && echo round trip ok
