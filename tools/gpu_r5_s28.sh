#!/bin/bash
# round 5, session 28: the library with its decompress half compiled at -O3 -fno-unroll-loops: the GPU suite, decoder and stream-decode times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/pytest_two_units.txt
timeout 300 python tests/dev/dectime.py 2>&1 | tail -3 | tee $OUT/decompress_blocks_times.txt
{
for c in text lowent random; do echo "== $c"; timeout 200 python tests/dev/stream_dec_stages.py $c 2>&1 | grep -E "decoded in|round trip|scanned|resolve" | tail -8; done
} 2>&1 | tee $OUT/stream_decode_masks_stages.txt
