#!/bin/bash
# round 5, session 21: the segment decoder's step with its flags as masks: stage times on the three classes, the parity tests that decode streams, a short fuzz
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for c in text lowent random; do echo "== $c"; timeout 200 python tests/dev/stream_dec_stages.py $c 2>&1 | grep -E "decoded in|round trip|scanned|resolve" | tail -8; done
} 2>&1 | tee $OUT/stream_decode_masks_stages.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_dropin.py tests/test_gpu_incremental.py -x -q -k "stream or concat or file or incremental or segment or decode or decompress" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/pytest_stream_masks.txt
timeout 400 python tests/dev/fuzz_all.py 240 8001 2>&1 | tail -2 | tee $OUT/fuzz_240s_seed8001.txt
