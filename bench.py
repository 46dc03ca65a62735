#!/usr/bin/env python3
"""bench.py -- input GB/s of the LZS block-compression hot path on MI355X.

A "step" is one pass of lzs_compress_batch_device over one batch of independent 64 KiB
blocks already resident in HBM (BASELINE.json configs[1]: 1 GiB = 16384 blocks of
enwik-style ASCII text per GPU).  Blocks shard across GPUs with no data-path collective
(weak scaling: every rank compresses its own 1 GiB shard); `value` is the whole-job rate.

    python bench.py                       # 1 GPU, text class
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 2

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import lzs_compression_amd as lzs          # noqa: E402
from lzs_compression_amd import workload   # noqa: E402

BLOCK = 65536
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level table)


def usable_cores() -> int:
    """Host cores this process may really use: the smaller of the affinity mask and the
    cgroup CPU quota (the GPU box advertises 256 CPUs but grants a 16-CPU quota)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def pmc_traffic(cls: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None.
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 wide streaming reads."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))[cls]
        return {"hbm_bytes": rec["fetch_bytes_corrected"] + rec["write_bytes"], **rec}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(blocks_host: np.ndarray, gpu_len: np.ndarray, gpu_slots) -> dict:
    """The reference C compressor (oracle/_ref, kind "reference") or, where it did not
    travel, our C restatement (kind "port"), one block per task on the host cores, on a
    bounded sample of the same workload.  Also cross-checks the GPU output on that sample."""
    import oracle
    cores = usable_cores()
    kind = "reference" if oracle.have_ref() else "port"
    codec = oracle.ref() if kind == "reference" else oracle.oracle()
    # size the sample for roughly 15 core-seconds of work: probe 64 blocks on one thread first
    _, _, probe = oracle.run_blocks(codec, blocks_host[:64], threads=1)
    per_block = max(probe / 64.0, 1e-6)
    nsample = int(min(len(blocks_host), max(256, 15.0 / per_block)))
    out, out_len, secs = oracle.run_blocks(codec, blocks_host[:nsample], threads=cores)
    _, _, secs1 = oracle.run_blocks(codec, blocks_host[:max(64, nsample // cores)], threads=1)
    one_core = max(64, nsample // cores) * BLOCK / secs1 / 1e9
    exact = bool((out_len == gpu_len[:nsample]).all())
    if exact:
        g = gpu_slots[:nsample].cpu().numpy()
        for b in range(0, nsample, max(1, nsample // 64)):
            exact = exact and g[b, :out_len[b]].tobytes() == out[b, :out_len[b]].tobytes()
    return {"value": nsample * BLOCK / secs / 1e9, "unit": "GB/s", "cores": cores, "kind": kind,
            "host": f"{os.cpu_count()} logical CPUs visible, {cores} usable (affinity/cgroup quota)",
            "one_core_GBps": one_core,
            "sample": f"first {nsample} of the same 64 KiB blocks ({nsample * BLOCK >> 20} MiB), "
                      f"one block per task, {cores} threads",
            "gpu_output_bit_exact_on_sample": exact}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="text", choices=workload.CLASS_NAMES)
    ap.add_argument("--blocks", type=int, default=16384, help="64 KiB blocks per GPU (16384 = 1 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", torch.cuda.current_device())

    # ---- this rank's shard: blocks [rank*nb, (rank+1)*nb) of the seeded class, into HBM
    nb = args.blocks
    host = workload.fill(args.workload, nb, BLOCK, first_block=rank * nb)
    x = torch.from_numpy(host).to(dev)
    slot_stride = (lzs.compressed_max(BLOCK) + 15) // 16 * 16
    slots = torch.empty((nb, slot_stride), dtype=torch.uint8, device=dev)
    lens = torch.empty(nb, dtype=torch.int32, device=dev)
    cap = lzs.compressed_max(BLOCK)

    def step():
        lzs.compress_blocks(x, None, cap, slots, lens)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # per-launch durations with HIP events on the stream the kernel is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    # N > 1: one compressed-output gather to rank 0 (compaction + RCCL gather-v), outside
    # the timed steps and reported separately -- the compute path itself has no collective
    gather = None
    if dist is not None:
        from lzs_compression_amd import sharding
        try:
            barrier()
            g0 = time.perf_counter()
            dense, offsets = lzs.compact(slots, lens)
            nbytes = int(offsets[-1].item())
            got, counts = sharding.gather_streams(dense, nbytes)
            barrier()
            gsec = time.perf_counter() - g0
            gather = {"ms": gsec * 1e3, "bytes": int(sum(counts)), "GBps": sum(counts) / gsec / 1e9}
            del dense, got
        except Exception as e:          # the headline number must survive a gather failure
            gather = {"error": repr(e)}

    total_in = world * nb * BLOCK * args.steps
    lens_h = lens.cpu().numpy()
    ratio = float(lens_h.sum()) / (nb * BLOCK)

    if rank == 0:
        avg_ms = float(np.mean(kernel_ms))
        traffic = pmc_traffic(args.workload) if nb == 16384 else None
        in_bytes = nb * BLOCK
        algo_bytes = in_bytes + int(lens_h.sum()) + 4 * nb
        achieved = in_bytes / (avg_ms * 1e-3) / 1e9
        result = {
            "metric": "input GB/s on 64KiB blocks, bit-exact vs C ref",
            "value": total_in / elapsed / 1e9,
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{nb} independent 64 KiB blocks per GPU ({nb * BLOCK >> 20} MiB), "
                                   f"class '{args.workload}' (seeded generator), device-resident",
                       "class": args.workload, "blocks_per_gpu": nb, "block_bytes": BLOCK,
                       "sharding": f"blocks/{world} per rank, no data-path collective",
                       "compression_ratio": ratio},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("hbm_bytes"), "traffic_detail": traffic,
                         "kernel": "lzs_compress_blocks_wg_kernel",
                         "algorithmic_bytes_per_launch": {"read_input": in_bytes,
                                                          "total_read_plus_written": algo_bytes},
                         "avg_kernel_ms": avg_ms, "min_kernel_ms": float(np.min(kernel_ms)),
                         "total_GBps": algo_bytes / (avg_ms * 1e-3) / 1e9},
        }
        if gather is not None:
            result["compressed_output_gather"] = gather
        if world == 1:
            # secondary, outside the timed region: the same bytes as ONE stream through
            # lzs_compress_stream_device, wall clock around the synchronous call
            try:
                flat = x.reshape(-1)
                buf, nbytes = lzs.compress_stream(flat)
                t = time.perf_counter()
                buf, nbytes = lzs.compress_stream(flat, buf)
                dt = time.perf_counter() - t
                result["single_stream"] = {"entry": "lzs_compress_stream_device", "input_bytes": int(flat.numel()),
                                           "compressed_bytes": nbytes, "ms": dt * 1e3,
                                           "value": flat.numel() / dt / 1e9, "unit": "GB/s"}
                back, got = lzs.decompress_stream(buf[:nbytes], flat.numel() + 16)
                t = time.perf_counter()
                back, got = lzs.decompress_stream(buf[:nbytes], flat.numel() + 16, back)
                dt = time.perf_counter() - t
                result["single_stream"]["decompress"] = {
                    "entry": "lzs_decompress_stream_device", "ms": dt * 1e3, "value": got / dt / 1e9,
                    "unit": "GB/s of output", "round_trip": bool(got == flat.numel() and torch.equal(back[:got], flat))}
            except Exception as exc:                      # never let the extra line spoil the contract line
                result["single_stream"] = {"error": str(exc)}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(host, lens_h, slots)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
