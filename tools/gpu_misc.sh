#!/bin/bash
# Launcher-path and PCIe-inclusive sanity checks on the 1-GPU box.
cd $GRAFT_REPO_ROOT
echo "== torchrun nproc=1 (driver's launch form, WORLD_SIZE=1)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
echo "== nccl world of 1: sharding helpers on device tensors"
timeout 300 python - <<'PY'
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29512")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
import lzs_compression_amd as lzs
from lzs_compression_amd import sharding, workload
x = torch.from_numpy(workload.fill("text", 64)).cuda()
mine = sharding.scatter_blocks(x, 64, 65536, x.device)
slots, lens = lzs.compress_blocks(mine)
dense, offs = lzs.compact(slots, lens)
got, counts = sharding.gather_streams(dense, int(offs[-1].item()))
alll = sharding.gather_lengths(lens)
torch.cuda.synchronize()
print("gathered", got.numel(), counts, int(alll.sum().item()) == got.numel())
dist.destroy_process_group()
PY
echo "== host-buffer API (PCIe inclusive, pageable memory)"
timeout 300 python - <<'PY'
import time, numpy as np, lzs_compression_amd as lzs
from lzs_compression_amd import workload
for nb in (256, 4096):
    blocks = workload.fill("text", nb)
    lzs.compress_batch(blocks[:8])
    t = time.perf_counter(); out, n = lzs.compress_batch(blocks); dt = time.perf_counter() - t
    print(f"lzs_compress_batch host buffers: {nb} blocks {nb*65536/dt/1e9:.2f} GB/s ({dt*1e3:.1f} ms)")
PY
