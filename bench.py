#!/usr/bin/env python3
"""bench.py -- input GB/s of the LZS block-compression hot path on MI355X.

N = 1: a "step" is one pass of lzs_compress_batch_device over one batch of independent 64 KiB
blocks already resident in HBM (BASELINE.json configs[1]: 1 GiB = 16384 blocks of
enwik-style ASCII text).

N > 1 (BASELINE.json configs[4]): 131072 blocks (8 GiB) per GPU --
64 GiB at 8 GPUs -- generated in HBM on the root GPU by lzs_gen_blocks_kernel; a "step" is the
whole job: SCATTER (root -> ranks, one batched group of RCCL point-to-point sends over xGMI),
COMPRESS (every rank its shard; blocks are independent, no collective), GATHER (compaction +
gather-v of the compressed streams and their lengths to the root).  `value` is the end-to-end
rate of the whole job; the three phases are timed separately and the compute-only rate is
reported next to it.  Weak scaling: per-GPU work is fixed.

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


def kernel_identity() -> str:
    """SHA-256 over the sources the measured kernel, lzs_compress_blocks_wg_kernel, is compiled from
    (kernels/common.inc and kernels/compress_wg.inc: its code, its LDS layout and its launch
    geometry -- one workgroup of kWgThreads per block): what a counter measurement belongs to.
    (The decoders' sources are not part of it: the traffic figure is the compress kernel's.)"""
    import hashlib
    csrc = os.path.join(ROOT, "lzs_compression_amd", "csrc", "kernels")
    h = hashlib.sha256()
    for name in ("common.inc", "compress_wg.inc"):
        h.update(name.encode() + b"\0" + open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()


def pmc_traffic(cls: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/gpu_traffic.sh), or (None, why).  The file carries the identity of the kernel
    sources it was measured on; with other sources in the tree the number is stale and not reported.
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 wide streaming reads."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
        rec = doc[cls]
    except (OSError, KeyError, ValueError):
        return None, "profiles/pmc_traffic.json has no entry for this class"
    if doc.get("kernel_source_sha256") != kernel_identity():
        return None, ("profiles/pmc_traffic.json was measured on other kernel sources (its kernel_source_sha256 differs "
                      "from this tree's): stale, not reported; regenerate with tools/gpu_traffic.sh")
    return {"hbm_bytes": rec["fetch_bytes_corrected"] + rec["write_bytes"], "kernel_source_sha256": doc["kernel_source_sha256"], **rec}, None


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


def run_sharded(args, dist, rank: int, world: int, dev) -> None:
    """N > 1: the config-5 job (lzs_compression_amd/sharded_job.py), K timed passes."""
    from lzs_compression_amd.sharded_job import ShardedCompressJob
    nb = args.blocks if args.blocks is not None else 131072
    cap = lzs.compressed_max(BLOCK)
    slot_stride = (cap + 15) // 16 * 16
    kernel_ev = []

    def compress(x, slots, lens):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lzs.compress_blocks(x, None, cap, slots, lens)
        b.record()
        kernel_ev.append((a, b))

    def compact(slots, lens, dense, offsets):
        lzs.compact(slots, lens, dense=dense, offsets=offsets)
        return int(offsets[-1].item())

    job = ShardedCompressJob(nb, BLOCK, slot_stride, dev, compress, compact, torch.cuda.synchronize)
    # ---- the root generates every rank's blocks in HBM, one piece (<= 8 GiB) per rank
    pieces = None
    t_gen = time.perf_counter()
    if rank == 0:
        pieces = [workload.fill_device(args.workload, nb, BLOCK, first_block=r * nb, device=dev) for r in range(world)]
        torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.step(pieces)
    kernel_ev.clear()
    barrier()
    t0 = time.perf_counter()
    phases = [job.step(pieces) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    # MAX over ranks of the job time and of every phase's mean
    mean = {k: float(np.mean([p[k] for p in phases])) for k in ("scatter", "compress", "gather", "total")}
    t = torch.tensor([elapsed, mean["scatter"], mean["compress"], mean["gather"], mean["total"]], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, sc, co, ga, tot = (float(v) for v in t.tolist())
    kernel_ms = [a.elapsed_time(b) for a, b in kernel_ev]

    # ---- checks outside the timed region: every rank decodes its own shard on the device ...
    x = pieces[0] if rank == 0 else job.mine
    back, back_len = lzs.decompress_blocks(job.slots, job.lens, BLOCK)
    ok = bool((back_len == BLOCK).all()) and torch.equal(back[:, :BLOCK], x)
    del back
    okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    lens_h = job.lens.cpu().numpy()
    if rank == 0:
        # ... and the root compares sampled blocks of EVERY rank's part of the gathered bytes with the
        # CPU oracle (first, middle and last block of each shard), at the gathered offsets
        import oracle
        O = oracle.oracle()
        all_lens = job.all_lens.cpu().numpy().astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(all_lens)])
        gathered_ok = int(offs[-1]) == sum(job.counts) == int(job.out.numel())
        for r in range(world):
            for b in (0, nb // 2, nb - 1):
                g = r * nb + b
                got = bytes(job.out[int(offs[g]):int(offs[g + 1])].cpu().numpy())
                gathered_ok = gathered_ok and got == O.compress(bytes(pieces[r][b].cpu().numpy()))
        total_in = world * nb * BLOCK
        in_bytes = nb * BLOCK
        avg_ms = float(np.mean(kernel_ms))
        achieved = in_bytes / (avg_ms * 1e-3) / 1e9
        algo_bytes = in_bytes + int(lens_h.sum()) + 4 * nb
        result = {
            "metric": "input GB/s on 64KiB blocks, bit-exact vs C ref",
            "value": total_in * args.steps / elapsed / 1e9,
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{world * nb} independent 64 KiB blocks ({world * nb * BLOCK >> 30} GiB / {world} GPUs), class "
                                   f"'{args.workload}', generated in HBM on the root GPU, scattered {nb} blocks "
                                   f"({nb * BLOCK >> 30} GiB) per rank over RCCL/xGMI, compressed, compacted, gathered to the root",
                       "class": args.workload, "blocks_per_gpu": nb, "block_bytes": BLOCK,
                       "sharding": f"contiguous block ranges, {nb} per rank; a step = scatter + compress + gather",
                       "compression_ratio": float(all_lens.sum()) / total_in},
            "phases_ms": {"scatter": sc * 1e3, "compress": co * 1e3, "gather": ga * 1e3, "step": tot * 1e3,
                          "note": "mean over the timed steps, MAX over ranks; each phase ends with a device synchronize"},
            "end_to_end_GBps": total_in * args.steps / elapsed / 1e9,
            "compute_only_GBps": total_in / co / 1e9,
            "scatter_GBps": (world - 1) * nb * BLOCK / sc / 1e9 if sc > 0 else None,
            "gather_GBps": (sum(job.counts) - job.counts[0]) / ga / 1e9 if ga > 0 else None,
            "gathered_bytes": int(sum(job.counts)),
            "generate_on_root_s": t_gen,
            "checks": {"every_rank_round_trip_on_device": bool(okt.item()), "gathered_samples_equal_oracle": bool(gathered_ok)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "lzs_compress_blocks_wg_kernel (rank 0's launches)",
                         "algorithmic_bytes_per_launch": {"read_input": in_bytes, "total_read_plus_written": algo_bytes},
                         "avg_kernel_ms": avg_ms, "min_kernel_ms": float(np.min(kernel_ms))},
        }
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        emit(result)


def emit(result: dict) -> None:
    """The ONE JSON line, last on stdout: RCCL writes a version banner through C stdio, which a pipe
    holds back until exit -- flush it out first."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(result), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="text", choices=workload.CLASS_NAMES)
    ap.add_argument("--blocks", type=int, default=None,
                    help="64 KiB blocks per GPU (default: 16384 = 1 GiB at N = 1, 131072 = 8 GiB at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true", help="skip the secondary single_stream line (profiling runs)")
    ap.add_argument("--sharded-job", action="store_true",
                    help="run the N > 1 job (scatter / compress / gather over torch.distributed) even at N = 1: "
                         "exercises that code path with the nccl backend on a one-GPU box")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.sharded_job:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist = None
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1 or args.sharded_job:
        run_sharded(args, dist, rank, world, dev)
        return
    if args.blocks is None:
        args.blocks = 16384

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

    def barrier():                   # (N = 1: no other rank to wait for)
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
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    total_in = world * nb * BLOCK * args.steps
    lens_h = lens.cpu().numpy()
    ratio = float(lens_h.sum()) / (nb * BLOCK)

    if rank == 0:
        avg_ms = float(np.mean(kernel_ms))
        traffic, traffic_note = pmc_traffic(args.workload) if nb == 16384 else (None, "the counters were taken at 16384 blocks per launch")
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
                         "traffic": (traffic or {}).get("hbm_bytes"), "traffic_detail": traffic if traffic else traffic_note,
                         "kernel": "lzs_compress_blocks_wg_kernel",
                         "algorithmic_bytes_per_launch": {"read_input": in_bytes,
                                                          "total_read_plus_written": algo_bytes},
                         "avg_kernel_ms": avg_ms, "min_kernel_ms": float(np.min(kernel_ms)),
                         "median_kernel_ms": float(np.median(kernel_ms)),
                         "total_GBps": algo_bytes / (avg_ms * 1e-3) / 1e9},
        }
        if world == 1:
            # the box's own streaming figure next to the 8 TB/s of the data sheet (report
            # the fraction against both): a device-to-device copy of the same 1 GiB, read + written
            try:
                y = torch.empty_like(x)
                for _ in range(2):
                    y.copy_(x)
                c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c0.record()
                for _ in range(10):
                    y.copy_(x)
                c1.record()
                torch.cuda.synchronize()
                copy_gbps = 2.0 * x.numel() * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
                del y
                result["roofline"]["measured_device_copy_GBps"] = copy_gbps
                result["roofline"]["frac_of_measured_copy"] = result["roofline"]["total_GBps"] / copy_gbps
                result["roofline"]["measured_copy_note"] = ("torch device-to-device copy of the input tensor, bytes read + written per second; "
                                                             "frac_of_measured_copy = the kernel's read + written bytes per second over it")
            except Exception as exc:                       # noqa: BLE001
                result["roofline"]["measured_device_copy_GBps"] = None
                result["roofline"]["measured_copy_note"] = f"copy measurement failed: {exc}"
        if world == 1 and not args.no_single_stream:
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
    emit(result)


if __name__ == "__main__":
    main()
