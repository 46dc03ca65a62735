/* tests/cpu_shim/fake_rccl.h -- the "communicator" of tests/cpu_shim/fake_rccl.c: a stand-in for librccl over shared memory between
 * forked processes, loaded by the product's lzs_rccl.c through LZS_RCCL_LIBRARY.  TEST INFRASTRUCTURE ONLY. */
#ifndef FAKE_RCCL_H
#define FAKE_RCCL_H
#include <stddef.h>
#include <stdint.h>

enum { FAKE_RCCL_MAX_WORLD = 8, FAKE_RCCL_SLOT = 4096 };

/* One single-slot mailbox per ordered pair of ranks: a message is there while posted != taken. */
typedef struct {
    volatile uint64_t posted, taken;
    volatile uint64_t len;
    uint8_t data[FAKE_RCCL_SLOT];
} fake_channel_t;

typedef struct {
    fake_channel_t ch[FAKE_RCCL_MAX_WORLD][FAKE_RCCL_MAX_WORLD];          /* [from][to] */
    volatile uint64_t gather_gen[FAKE_RCCL_MAX_WORLD];                    /* all-gather rounds a rank has entered */
    volatile uint64_t gather_left[FAKE_RCCL_MAX_WORLD];                   /* ... and left */
    uint8_t gather_data[FAKE_RCCL_MAX_WORLD][64];
    volatile uint64_t sends, recvs, groups, bytes_moved;                  /* what went over the "wire": the test reads them */
    volatile uint64_t ops_outside_group;
} fake_shm_t;

typedef struct { int rank, world; fake_shm_t *shm; } fake_comm_t;

#endif
