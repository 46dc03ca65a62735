/*
 * lzs_rccl.c -- the two data moves of a sharded job (input scatter from a root GPU, compressed-output gather to it) over
 * RCCL, callable from a C host (include/lzs/lzs_shard.h).  The compute path between
 * them has no collective: blocks are independent (reference lzs-compression.c:291-299, 449-466).
 *
 * librccl is opened at run time (dlopen), not linked: liblzs.so keeps its dependencies, and a process that never
 * shards never loads it.  xGMI is point to point (7 links x ~153 GB/s per GPU), so the root's transfers to all its
 * peers go into ONE group and run on all links at once; nothing is relayed peer to peer.
 */
#include <dlfcn.h>

#include "lzs_internal.h"
#include "lzs/lzs_shard.h"

/* the few RCCL entry points used, by their C signatures (rccl.h: ncclResult_t is an int-sized enum, ncclSuccess = 0;
 * ncclUint8 = 1, ncclUint64 = 5) */
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_sendrecv_fn)(void *buf, size_t count, int dtype, int peer, void *comm, void *stream);
typedef int (*nccl_allgather_fn)(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream);
typedef const char *(*nccl_strerror_fn)(int);
#define NCCL_UINT8  1
#define NCCL_UINT64 5
#ifdef LZS_RCCL_PIECE                      /* (tests/cpu_shim: a few KB, so that a shard of a test spans several pieces) */
#define RCCL_PIECE ((size_t)LZS_RCCL_PIECE)
#else
#define RCCL_PIECE ((size_t)1 << 30)
#endif

static struct {
    pthread_once_t once;
    void *lib;
    nccl_group_fn group_start, group_end;
    nccl_sendrecv_fn send, recv;
    nccl_allgather_fn all_gather;
    nccl_strerror_fn strerror;
    char why[256];
} g_rccl = { PTHREAD_ONCE_INIT, NULL, NULL, NULL, NULL, NULL, NULL, NULL, "" };

static void rccl_open(void)
{
    const char *name = getenv("LZS_RCCL_LIBRARY");
    const char *tries[] = { name, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (size_t i = 0; i < sizeof tries / sizeof tries[0] && !g_rccl.lib; i++)
        if (tries[i] && tries[i][0]) g_rccl.lib = dlopen(tries[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.lib) { snprintf(g_rccl.why, sizeof g_rccl.why, "librccl could not be opened (%s)", dlerror()); return; }
    g_rccl.group_start = (nccl_group_fn)dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.group_end = (nccl_group_fn)dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.send = (nccl_sendrecv_fn)dlsym(g_rccl.lib, "ncclSend");
    g_rccl.recv = (nccl_sendrecv_fn)dlsym(g_rccl.lib, "ncclRecv");
    g_rccl.all_gather = (nccl_allgather_fn)dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.strerror = (nccl_strerror_fn)dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.group_start || !g_rccl.group_end || !g_rccl.send || !g_rccl.recv || !g_rccl.all_gather) {
        snprintf(g_rccl.why, sizeof g_rccl.why, "librccl lacks ncclGroupStart / ncclSend / ncclRecv / ncclAllGather");
        g_rccl.lib = NULL;
    }
}

static int rccl_ready(const char *who)
{
    pthread_once(&g_rccl.once, rccl_open);
    return g_rccl.lib ? LZS_OK : fail(LZS_E_NO_DEVICE, "%s: %s", who, g_rccl.why);
}

static int rccl_fail(const char *who, const char *what, int e)
{
    return fail(LZS_E_HIP, "%s: %s: %s", who, what, g_rccl.strerror ? g_rccl.strerror(e) : "RCCL error");
}

void lzs_shard_range(size_t nblocks, int rank, int world, size_t *lo, size_t *hi)
{
    if (world < 1) world = 1;
    if (rank < 0) rank = 0;
    const size_t base = nblocks / (size_t)world, rem = nblocks % (size_t)world, r = (size_t)rank;
    const size_t a = r * base + (r < rem ? r : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (r < rem ? 1u : 0u);
}

int lzs_rccl_scatter_blocks(void *comm, void *d_mine, const void *d_all_on_root, size_t nblocks, size_t block_len,
                            int rank, int world, int root, void *hip_stream)
{
    const char *who = "lzs_rccl_scatter_blocks";
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return fail(LZS_E_ARG, "%s: rank %d / world %d / root %d", who, rank, world, root);
    size_t lo, hi;
    lzs_shard_range(nblocks, rank, world, &lo, &hi);
    if ((hi > lo && block_len && !d_mine) || (rank == root && nblocks && block_len && !d_all_on_root)) return fail(LZS_E_ARG, "%s: NULL buffer", who);
    int rc = require_device();
    if (rc != LZS_OK) return rc;
    int e;
    if (rank == root) {
        /* my own rows: a device copy, queued on the same stream */
        const uint8_t *mine = (const uint8_t *)d_all_on_root + lo * block_len;
        if (mine != (const uint8_t *)d_mine && (e = lzs_hip_d2d(d_mine, mine, (hi - lo) * block_len, hip_stream)) != 0) return hip_fail(e, "hipMemcpy D2D");
    }
    if (world == 1) return LZS_OK;
    if (!comm) return fail(LZS_E_ARG, "%s: no communicator", who);
    if ((rc = rccl_ready(who)) != LZS_OK) return rc;
    if ((e = g_rccl.group_start()) != 0) return rccl_fail(who, "ncclGroupStart", e);
    if (rank == root) {
        for (int r = 0; r < world && !e; r++) {
            if (r == root) continue;
            size_t rlo, rhi;
            lzs_shard_range(nblocks, r, world, &rlo, &rhi);
            const uint8_t *rows = (const uint8_t *)d_all_on_root + rlo * block_len;
            const size_t bytes = (rhi - rlo) * block_len;
            for (size_t at = 0; at < bytes && !e; at += RCCL_PIECE)
                e = g_rccl.send((void *)(rows + at), bytes - at < RCCL_PIECE ? bytes - at : RCCL_PIECE, NCCL_UINT8, r, comm, hip_stream);
        }
    } else {
        const size_t bytes = (hi - lo) * block_len;
        for (size_t at = 0; at < bytes && !e; at += RCCL_PIECE)
            e = g_rccl.recv((uint8_t *)d_mine + at, bytes - at < RCCL_PIECE ? bytes - at : RCCL_PIECE, NCCL_UINT8, root, comm, hip_stream);
    }
    const int e2 = g_rccl.group_end();
    if (e) return rccl_fail(who, rank == root ? "ncclSend" : "ncclRecv", e);
    if (e2) return rccl_fail(who, "ncclGroupEnd", e2);
    return LZS_OK;
}

int lzs_rccl_gather_streams(void *comm, void *d_out_on_root, uint64_t *counts, uint64_t *d_counts, const void *d_dense,
                            const uint64_t *d_my_count, int rank, int world, int root, void *hip_stream)
{
    const char *who = "lzs_rccl_gather_streams";
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return fail(LZS_E_ARG, "%s: rank %d / world %d / root %d", who, rank, world, root);
    if (!counts || !d_counts || !d_my_count) return fail(LZS_E_ARG, "%s: NULL count buffer", who);
    int rc = require_device();
    if (rc != LZS_OK) return rc;
    int e;
    /* every rank's byte count, on every rank: the root cannot post a receive before it knows the extent */
    if (world == 1 && !comm) {
        if ((e = lzs_hip_d2d(d_counts, d_my_count, sizeof(uint64_t), hip_stream)) != 0) return hip_fail(e, "hipMemcpy D2D");
    } else {                                           /* (a communicator of one rank goes through RCCL too: the one-GPU test of this binding) */
        if (!comm) return fail(LZS_E_ARG, "%s: no communicator", who);
        if ((rc = rccl_ready(who)) != LZS_OK) return rc;
        if ((e = g_rccl.all_gather(d_my_count, d_counts, 1, NCCL_UINT64, comm, hip_stream)) != 0) return rccl_fail(who, "ncclAllGather", e);
    }
    if ((e = lzs_hip_d2h(counts, d_counts, sizeof(uint64_t) * (size_t)world, hip_stream)) != 0) return hip_fail(e, "hipMemcpy D2H");
    if ((e = lzs_hip_stream_sync(hip_stream)) != 0) return hip_fail(e, "hipStreamSynchronize");
    if (rank == root) {
        uint64_t at = 0;
        for (int r = 0; r < root; r++) at += counts[r];
        if (counts[root] && !d_out_on_root) return fail(LZS_E_ARG, "%s: NULL output on the root", who);
        if ((const uint8_t *)d_dense != (const uint8_t *)d_out_on_root + at &&
            (e = lzs_hip_d2d((uint8_t *)d_out_on_root + at, d_dense, (size_t)counts[root], hip_stream)) != 0) return hip_fail(e, "hipMemcpy D2D");
    }
    if (world == 1) return LZS_OK;
    if ((e = g_rccl.group_start()) != 0) return rccl_fail(who, "ncclGroupStart", e);
    if (rank == root) {
        uint64_t at = 0;
        for (int r = 0; r < world && !e; r++) {
            if (r != root)
                for (uint64_t o = 0; o < counts[r] && !e; o += RCCL_PIECE)
                    e = g_rccl.recv((uint8_t *)d_out_on_root + at + o, (size_t)(counts[r] - o < RCCL_PIECE ? counts[r] - o : RCCL_PIECE), NCCL_UINT8, r, comm, hip_stream);
            at += counts[r];
        }
    } else {
        for (uint64_t o = 0; o < counts[rank] && !e; o += RCCL_PIECE)
            e = g_rccl.send((void *)((const uint8_t *)d_dense + o), (size_t)(counts[rank] - o < RCCL_PIECE ? counts[rank] - o : RCCL_PIECE), NCCL_UINT8, root, comm, hip_stream);
    }
    const int e2 = g_rccl.group_end();
    if (e) return rccl_fail(who, rank == root ? "ncclRecv" : "ncclSend", e);
    if (e2) return rccl_fail(who, "ncclGroupEnd", e2);
    return LZS_OK;
}
