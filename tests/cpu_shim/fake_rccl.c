/*
 * tests/cpu_shim/fake_rccl.c -- a stand-in for librccl (ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv / ncclAllGather /
 * ncclGetErrorString, by their C signatures) over a shared-memory mailbox between forked processes, so that the product's
 * lzs_rccl_scatter_blocks / lzs_rccl_gather_streams (csrc/lzs_rccl.c, unchanged) run at world 2, 3 and 8 on the CPU, under the
 * sanitizers.  Built as a shared library and loaded through LZS_RCCL_LIBRARY -- the switch exists for this.
 *
 * What it keeps of RCCL's contract, so that a misuse shows: operations inside a group are only QUEUED and all run at
 * ncclGroupEnd, making progress together (a root's sends to seven peers and their receives complete in any order); a receive
 * whose count differs from the matching send's is an error; counts are in elements of the given type.
 * TEST INFRASTRUCTURE ONLY.
 */
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fake_rccl.h"

enum { OP_SEND = 1, OP_RECV = 2 };
typedef struct { int kind, peer; uint8_t *buf; size_t bytes; fake_comm_t *comm; int done; } op_t;
static __thread op_t *g_ops;
static __thread size_t g_nops, g_cap;
static __thread int g_depth;

static size_t type_size(int dtype) { return dtype == 1 /* ncclUint8 */ ? 1 : dtype == 5 /* ncclUint64 */ ? 8 : 0; }

const char *ncclGetErrorString(int e) { return e == 4 ? "invalid argument (fake rccl)" : e == 5 ? "invalid usage (fake rccl)" : e ? "error (fake rccl)" : "no error"; }

/* one attempt at an operation: 1 = done, 0 = not yet, < 0 = error */
static int try_op(op_t *o)
{
    fake_comm_t *c = o->comm;
    if (o->kind == OP_SEND) {
        fake_channel_t *ch = &c->shm->ch[c->rank][o->peer];
        if (__atomic_load_n(&ch->posted, __ATOMIC_ACQUIRE) != __atomic_load_n(&ch->taken, __ATOMIC_ACQUIRE)) return 0;   /* slot busy */
        if (o->bytes > FAKE_RCCL_SLOT) return -4;
        memcpy(ch->data, o->buf, o->bytes);
        ch->len = o->bytes;
        __atomic_add_fetch(&c->shm->sends, 1, __ATOMIC_RELAXED);
        __atomic_add_fetch(&c->shm->bytes_moved, o->bytes, __ATOMIC_RELAXED);
        __atomic_add_fetch(&ch->posted, 1, __ATOMIC_RELEASE);
        return 1;
    }
    fake_channel_t *ch = &c->shm->ch[o->peer][c->rank];
    if (__atomic_load_n(&ch->posted, __ATOMIC_ACQUIRE) == __atomic_load_n(&ch->taken, __ATOMIC_ACQUIRE)) return 0;       /* nothing there */
    if (ch->len != o->bytes) return -4;                                     /* send and receive must agree on the count */
    memcpy(o->buf, ch->data, o->bytes);
    __atomic_add_fetch(&c->shm->recvs, 1, __ATOMIC_RELAXED);
    __atomic_add_fetch(&ch->taken, 1, __ATOMIC_RELEASE);
    return 1;
}

/* all queued operations, together: sends and receives of one peer keep their order, different peers overlap */
static int run_ops(void)
{
    int err = 0;
    size_t left = g_nops;
    unsigned long spins = 0;
    while (left && !err) {
        int moved = 0;
        for (size_t i = 0; i < g_nops && !err; i++) {
            op_t *o = &g_ops[i];
            if (o->done) continue;
            int blocked = 0;                                                /* an earlier operation of the same kind and peer goes first */
            for (size_t j = 0; j < i; j++) if (!g_ops[j].done && g_ops[j].kind == o->kind && g_ops[j].peer == o->peer) { blocked = 1; break; }
            if (blocked) continue;
            const int r = try_op(o);
            if (r < 0) err = -r;
            else if (r) { o->done = 1; left--; moved = 1; }
        }
        if (!moved) { sched_yield(); if (++spins > 200000000ul) err = 1; }   /* (a hung peer must not hang the test run for ever) */
    }
    g_nops = 0;
    if (g_depth == 0) { free(g_ops); g_ops = NULL; g_cap = 0; }
    return err;
}

static int enqueue(int kind, void *buf, size_t count, int dtype, int peer, void *comm)
{
    fake_comm_t *c = (fake_comm_t *)comm;
    const size_t ts = type_size(dtype);
    if (!c || !ts || peer < 0 || peer >= c->world || peer == c->rank || (count && !buf)) return 4;
    if (g_nops == g_cap) { g_cap = g_cap ? 2 * g_cap : 64; g_ops = (op_t *)realloc(g_ops, g_cap * sizeof *g_ops); if (!g_ops) return 2; }
    g_ops[g_nops++] = (op_t){ kind, peer, (uint8_t *)buf, count * ts, c, 0 };
    if (g_depth == 0) { __atomic_add_fetch(&c->shm->ops_outside_group, 1, __ATOMIC_RELAXED); return run_ops(); }
    return 0;
}

int ncclGroupStart(void) { g_depth++; return 0; }
int ncclGroupEnd(void)
{
    if (g_depth <= 0) return 5;
    if (--g_depth) return 0;
    if (g_nops) __atomic_add_fetch(&g_ops[0].comm->shm->groups, 1, __ATOMIC_RELAXED);
    return run_ops();
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *stream) { (void)stream; return enqueue(OP_SEND, (void *)buf, count, dtype, peer, comm); }
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *stream) { (void)stream; return enqueue(OP_RECV, buf, count, dtype, peer, comm); }

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream)
{
    (void)stream;
    fake_comm_t *c = (fake_comm_t *)comm;
    const size_t bytes = count * type_size(dtype);
    if (!c || !bytes || bytes > sizeof c->shm->gather_data[0]) return 4;
    fake_shm_t *s = c->shm;
    /* nobody writes round g + 1 before everybody has read round g */
    const uint64_t g = s->gather_gen[c->rank] + 1;
    for (int r = 0; r < c->world; r++) while (__atomic_load_n(&s->gather_left[r], __ATOMIC_ACQUIRE) < g - 1) sched_yield();
    memcpy(s->gather_data[c->rank], send, bytes);
    __atomic_store_n(&s->gather_gen[c->rank], g, __ATOMIC_RELEASE);
    for (int r = 0; r < c->world; r++) {
        while (__atomic_load_n(&s->gather_gen[r], __ATOMIC_ACQUIRE) < g) sched_yield();
        memcpy((uint8_t *)recv + (size_t)r * bytes, s->gather_data[r], bytes);
    }
    __atomic_store_n(&s->gather_left[c->rank], g, __ATOMIC_RELEASE);
    return 0;
}
