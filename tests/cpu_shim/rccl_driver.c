/*
 * tests/cpu_shim/rccl_driver.c -- the C half of the sharded job (include/lzs/lzs_shard.h: lzs_rccl_scatter_blocks, the batch
 * compressor, lzs_compact_device, lzs_rccl_gather_streams -- csrc/lzs_rccl.c UNCHANGED but for a smaller piece size) run by
 * `world` forked processes over tests/cpu_shim/fake_rccl.c, which the library opens through LZS_RCCL_LIBRARY.  The device is
 * tests/cpu_shim/lzs_cpu_shim.c (heap memory, the oracle behind the launches).  What the root gathers must be the oracle's
 * streams of all blocks, back to back (blocks are independent: reference lzs-compression.c:291-299, 449-466).
 *   usage: rccl_driver WORLD ROOT NBLOCKS BLOCK_LEN        exit code 0 = every rank passed
 * TEST INFRASTRUCTURE ONLY.
 */
#define _DEFAULT_SOURCE            /* MAP_ANONYMOUS */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "lzs/lzs.h"
#include "lzs/lzs_batch.h"
#include "lzs/lzs_shard.h"
#include "fake_rccl.h"

size_t lzs_oracle_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);
int lzs_workload_fill(uint8_t *dst, unsigned cls, uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_len, int nthreads);

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL rank %d %s:%d: ", rank, __func__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); return 1; } } while (0)

static int run_rank(int rank, int world, int root, size_t nblocks, size_t block_len, fake_shm_t *shm)
{
    fake_comm_t comm = { rank, world, shm };
    size_t lo, hi;
    lzs_shard_range(nblocks, rank, world, &lo, &hi);
    const size_t mine = hi - lo, cap = LZS_COMPRESSED_MAX(block_len), stride = (cap + 15) / 16 * 16;
    /* "device" buffers: the shim's device memory is the heap */
    uint8_t *all = NULL;
    if (rank == root) {
        all = (uint8_t *)malloc(nblocks * block_len + 1);
        /* text, low-entropy and high-entropy blocks in turn: shards of very different compressed sizes */
        for (size_t b = 0; b < nblocks; b++) lzs_workload_fill(all + b * block_len, (unsigned)(b % 3), 0x4C5A5331ull, b, 1, block_len, 1);
    }
    uint8_t *d_mine = (uint8_t *)malloc(mine * block_len + 1);
    memset(d_mine, 0xEE, mine * block_len + 1);
    int rc = lzs_rccl_scatter_blocks(&comm, d_mine, all, nblocks, block_len, rank, world, root, NULL);
    CHECK(rc == LZS_OK, "lzs_rccl_scatter_blocks: %s", lzs_last_error());
    CHECK(d_mine[mine * block_len] == 0xEE, "scatter wrote past the shard");
    /* every rank can make its own rows again: what arrived must be them */
    uint8_t *want_rows = (uint8_t *)malloc(mine * block_len + 1);
    for (size_t b = 0; b < mine; b++) lzs_workload_fill(want_rows + b * block_len, (unsigned)((lo + b) % 3), 0x4C5A5331ull, lo + b, 1, block_len, 1);
    CHECK(memcmp(d_mine, want_rows, mine * block_len) == 0, "the scattered rows differ from blocks [%zu, %zu)", lo, hi);

    uint8_t *d_slots = (uint8_t *)malloc(mine * stride + 1);
    uint32_t *d_len = (uint32_t *)malloc((mine + 1) * sizeof(uint32_t));
    uint8_t *d_dense = (uint8_t *)malloc(mine * stride + 1);
    uint64_t *d_offs = (uint64_t *)malloc((mine + 1) * sizeof(uint64_t));
    if (mine) {
        rc = lzs_compress_batch_device(d_slots, stride, cap, d_len, d_mine, block_len, NULL, block_len, mine, NULL);
        CHECK(rc == LZS_OK, "lzs_compress_batch_device: %s", lzs_last_error());
    }
    rc = lzs_compact_device(d_dense, d_offs, d_slots, stride, d_len, mine, NULL);
    CHECK(rc == LZS_OK, "lzs_compact_device: %s", lzs_last_error());

    uint64_t counts[FAKE_RCCL_MAX_WORLD], *d_counts = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)world);
    memset(counts, 0xFF, sizeof counts);
    /* the root's output: sized by the worst case, with a guard behind what will be used */
    const size_t out_cap = nblocks * cap + 64;
    uint8_t *d_out = rank == root ? (uint8_t *)malloc(out_cap) : NULL;
    if (d_out) memset(d_out, 0x5A, out_cap);
    rc = lzs_rccl_gather_streams(&comm, d_out, counts, d_counts, d_dense, d_offs + mine, rank, world, root, NULL);
    CHECK(rc == LZS_OK, "lzs_rccl_gather_streams: %s", lzs_last_error());
    CHECK(counts[rank] == d_offs[mine], "my own count came back as %llu, not %llu", (unsigned long long)counts[rank], (unsigned long long)d_offs[mine]);

    int bad = 0;
    if (rank == root) {
        uint8_t *want = (uint8_t *)malloc(out_cap), *one = (uint8_t *)malloc(cap);
        size_t total = 0, at_rank = 0;
        for (int r = 0; r < world; r++) {
            size_t rlo, rhi, bytes = 0;
            lzs_shard_range(nblocks, r, world, &rlo, &rhi);
            for (size_t b = rlo; b < rhi; b++) { const size_t w = lzs_oracle_compress(one, cap, all + b * block_len, block_len); memcpy(want + total + bytes, one, w); bytes += w; }
            if (counts[r] != bytes) { fprintf(stderr, "FAIL rank %d: rank %d's count %llu, the oracle's %zu\n", rank, r, (unsigned long long)counts[r], bytes); bad = 1; }
            total += bytes; at_rank += bytes;
        }
        if (!bad && memcmp(d_out, want, total) != 0) { fprintf(stderr, "FAIL rank %d: the gathered bytes differ from the oracle's concatenation (%zu bytes)\n", rank, total); bad = 1; }
        for (size_t i = total; i < total + 64 && !bad; i++) if (d_out[i] != 0x5A) { fprintf(stderr, "FAIL rank %d: the gather wrote past the total\n", rank); bad = 1; }
        free(want); free(one);
    }
    free(all); free(d_mine); free(want_rows); free(d_slots); free(d_len); free(d_dense); free(d_offs); free(d_counts); free(d_out);
    lzs_release_thread_cache();
    return bad;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: rccl_driver WORLD ROOT NBLOCKS BLOCK_LEN\n"); return 2; }
    const int world = atoi(argv[1]), root = atoi(argv[2]);
    const size_t nblocks = (size_t)atol(argv[3]), block_len = (size_t)atol(argv[4]);
    if (world < 1 || world > FAKE_RCCL_MAX_WORLD || root < 0 || root >= world) return 2;
    fake_shm_t *shm = (fake_shm_t *)mmap(NULL, sizeof *shm, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (shm == MAP_FAILED) { perror("mmap"); return 2; }
    memset(shm, 0, sizeof *shm);
    pid_t pids[FAKE_RCCL_MAX_WORLD];
    for (int r = 0; r < world; r++) {
        pids[r] = fork();
        if (pids[r] < 0) { perror("fork"); return 2; }
        if (pids[r] == 0) _exit(run_rank(r, world, root, nblocks, block_len, shm) ? 1 : 0);   /* (_exit: the parent's atexit work is the parent's; leaks are checked in-process below) */
    }
    int failures = 0;
    for (int r = 0; r < world; r++) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "rank %d: exit status 0x%x\n", r, st); failures++; }
    }
    printf("world %d root %d: %zu blocks of %zu bytes; %llu sends, %llu receives in %llu groups, %llu bytes between ranks, %llu operations outside a group; %d failure(s)\n",
           world, root, nblocks, block_len, (unsigned long long)shm->sends, (unsigned long long)shm->recvs, (unsigned long long)shm->groups,
           (unsigned long long)shm->bytes_moved, (unsigned long long)shm->ops_outside_group, failures);
    return failures ? 1 : 0;
}
