/*
 * lzs_shard.h -- independent blocks sharded over the GPUs of one node, from a C host (additive; not in the reference).
 *
 * The reference has no multi-GPU code; its blocks are independent by construction (every lzs_compress() call starts with
 * an empty history and ends with its own end marker: c/src/liblzs/lzs-compression.c:291-299, 449-466), so the compute
 * path needs no exchange: rank r of G compresses blocks [r N / G, (r + 1) N / G) with lzs_compress_batch_device().  Data
 * moves only where a job says so (BASELINE.json configs[4]: input scatter from a root GPU, compressed-output gather to
 * it), and these three calls are those moves over RCCL -- one process per GPU, the communicator made by the caller
 * (ncclCommInitRank), every rank calling each of them in the same order:
 *
 *     lzs_shard_range(N, rank, world, &lo, &hi);                               // my blocks
 *     lzs_rccl_scatter_blocks(comm, d_mine, d_all_on_root, N, block_len, rank, world, root, stream);
 *     lzs_compress_batch_device(d_slots, stride, cap, d_len, d_mine, block_len, NULL, block_len, hi - lo, stream);
 *     lzs_compact_device(d_dense, d_offs, d_slots, stride, d_len, hi - lo, stream);      // d_offs[hi - lo] = my byte count
 *     lzs_rccl_gather_streams(comm, d_out_on_root, counts, d_counts, d_dense, d_offs + (hi - lo), rank, world, root, stream);
 *
 * librccl is not a link-time dependency of liblzs: it is opened when the first of the two transfer calls runs
 * (LZS_RCCL_LIBRARY names another file), and a process that never calls them never loads it.  The Python job
 * (lzs_compression_amd/sharded_job.py, bench.py --gpus N) issues the same operations through torch.distributed, pipelined
 * over chunks; INTEGRATION.md section 4 has both side by side.
 */
#ifndef LZS_SHARD_H
#define LZS_SHARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Contiguous block range [*lo, *hi) of `rank` among `world` ranks; remainders go to the low ranks.  Pure arithmetic. */
void lzs_shard_range(size_t nblocks, int rank, int world, size_t *lo, size_t *hi);

/* Root -> all: rank r receives rows [lo_r, hi_r) of the root's `nblocks` x `block_len` bytes (contiguous, device memory)
 * into d_mine (device, (hi - lo) * block_len bytes; on the root a device copy).  The root's sends are ONE RCCL group
 * (ncclGroupStart, one ncclSend per peer and piece of <= 1 GiB, ncclGroupEnd), so all its xGMI links run at once.
 * `comm` is an ncclComm_t, `hip_stream` a hipStream_t; asynchronous on the stream.  LZS_OK or LZS_E_* (lzs_last_error()). */
int lzs_rccl_scatter_blocks(void *comm, void *d_mine, const void *d_all_on_root, size_t nblocks, size_t block_len,
                            int rank, int world, int root, void *hip_stream);

/* All -> root, variable length: every rank's d_dense[0 .. its count) lands in d_out_on_root, concatenated in rank order.
 * d_my_count: the rank's byte count as a uint64 in DEVICE memory (lzs_compact_device leaves it in d_offsets[nblocks]);
 * d_counts: device scratch for `world` uint64; counts: host array of `world` entries that receives every rank's count.
 * An ncclAllGather of the counts, a wait for it (the root must know the extents before it can post its receives), then
 * one RCCL group of ncclRecv (root) / ncclSend (peers) in pieces of <= 1 GiB.  d_out_on_root must hold the sum. */
int lzs_rccl_gather_streams(void *comm, void *d_out_on_root, uint64_t *counts, uint64_t *d_counts, const void *d_dense,
                            const uint64_t *d_my_count, int rank, int world, int root, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
