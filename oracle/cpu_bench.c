/*
 * oracle/cpu_bench.c -- thread-pool driver that times a CPU one-shot codec
 * (our restatement, or the compiled reference in oracle/_ref/) over independent
 * blocks: one block per task, `nthreads` workers.
 *
 * TEST INFRASTRUCTURE ONLY (see lzs_oracle.c).  Used by tests/ as a fast way to
 * run the checker over many blocks and by bench.py's cpu_baseline leg.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <time.h>

typedef size_t (*codec_fn)(uint8_t *out, size_t out_cap, const uint8_t *in, size_t in_len);

typedef struct {
    codec_fn        fn;
    const uint8_t  *in;
    size_t          in_stride;
    const uint32_t *in_len;      /* per block, or NULL -> in_len_uniform */
    size_t          in_len_uniform;
    uint8_t        *out;
    size_t          out_stride;
    size_t          out_cap;
    uint32_t       *out_len;
    size_t          nblocks;
    size_t          next;        /* work counter */
    pthread_mutex_t lock;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    for (;;) {
        pthread_mutex_lock(&j->lock);
        size_t b = j->next;
        size_t e = b + 8 < j->nblocks ? b + 8 : j->nblocks;
        j->next = e;
        pthread_mutex_unlock(&j->lock);
        if (b >= j->nblocks)
            return NULL;
        for (; b < e; b++) {
            size_t n = j->in_len ? j->in_len[b] : j->in_len_uniform;
            size_t got = j->fn(j->out + b * j->out_stride, j->out_cap,
                               j->in + b * j->in_stride, n);
            if (j->out_len)
                j->out_len[b] = (uint32_t)got;
        }
    }
}

/* Returns wall seconds for the whole job (threads included), or -1.0. */
double lzs_cpu_run_blocks(codec_fn fn, uint8_t *out, size_t out_stride, size_t out_cap,
                          uint32_t *out_len, const uint8_t *in, size_t in_stride,
                          const uint32_t *in_len, size_t in_len_uniform, size_t nblocks,
                          int nthreads)
{
    job_t j = { fn, in, in_stride, in_len, in_len_uniform, out, out_stride, out_cap,
                out_len, nblocks, 0, PTHREAD_MUTEX_INITIALIZER };
    pthread_t tid[256];
    struct timespec t0, t1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nthreads; i++)
        if (pthread_create(&tid[i], NULL, worker, &j) != 0)
            return -1.0;
    for (int i = 0; i < nthreads; i++)
        pthread_join(tid[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
