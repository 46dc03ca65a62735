/*
 * lzs_pipeline.c -- large host-buffer batches (lzs_compress_batch / lzs_decompress_batch) with their
 * three stages overlapped.  What a drop-in caller of the reference's one-shot calls in a loop
 * (c/src/test/test-lzs.c:111-114) actually hands over are HOST buffers; round 3 copied them in, ran the
 * kernel and copied the results back one after the other on one stream: 12.1 / 15.7 GB/s at 1 GiB against
 * 64 GB/s in the kernel and 57 GB/s of pinned DMA on the same box.
 *
 * Here the batch is cut into chunks of blocks, and per chunk k
 *
 *     FILL(k)    host threads copy the caller's blocks into a pinned piece, laid out at the device stride
 *     H2D(k)     one DMA from the pinned piece                                      (one stream)
 *     RUN(j)     ONE launch for a group of G chunks: the kernel, slot compaction, the lengths to the host
 *                                                                                    (two streams in turn: two launches
 *                                                                                    on the device at once)
 *     D2H(k)     one DMA of exactly the bytes the chunk produced, into a pinned piece   (one stream)
 *     DRAIN(k)   host threads lay the blocks out in the caller's array (nothing past out_len[b] is touched)
 *
 * run as a pipeline driven by what is ready (events are asked, not waited for, as long as there is anything else to
 * do): while launches are on the device, later chunks are filled and copied in and earlier ones copied out and
 * drained.  Copies go by the chunk (~45 MiB), launches by the group: a compress launch wants 1280 blocks to fill the
 * device (G = 2), a DEcompress launch takes as long as its slowest block whatever its size (a wavefront walks eight
 * streams token by token: 7.4 ms for 88 blocks of 64 KiB, 8.2 ms for 5632 -- tests/dev/dectime_chunks.py), so only
 * thousands of blocks at once reach the rate of one big launch (G = 8, two launches in flight).  Two pinned pieces on
 * the way in, three on the way out.
 *
 * Measured (1 GiB of text in 64 KiB blocks, the caller's buffers touched beforehand, MI355X box with a 16-CPU quota;
 * tests/dev/hostbatch_time2.py, hostbatch_sweep.py; profiles/r04/hostbatch_*.txt): lzs_compress_batch 11-12.6 GB/s one
 * after the other -> 45 GB/s; lzs_decompress_batch 16-17 -> 29 GB/s (what is left: the first launch's 9 ms before any
 * byte can leave, then 1.07 GB over the copy engine).
 * Pageable memory cannot be read by the copy engines at more than 13-14 GB/s and pinning the caller's buffers
 * in place costs 79 ms per GiB (tools/probes/host_copy_probe.hip), hence the pinned pieces and the threads
 * (one thread's memcpy is 31 GB/s there: the single-thread variant of round 3 lost).
 *
 * Nothing here knows the codec: `launch` is lzs_hip_launch_compress or lzs_hip_launch_decompress.
 */
#include "lzs_internal.h"

#define PIPE_MAX_THREADS 8
#define PIPE_IN(k)   ((k) & 1)
#define PIPE_OUT(k)  (2 + (k) % 3)
#define PIPE_RUN     2              /* run streams: with the two copy streams that is four streams, what the runtime gives
                                     * hardware queues of their own (with twelve streams a 46 MB copy-out stood in line
                                     * behind another stream's 9 ms kernel) */
#define PIPE_SLOTS   3              /* launches whose blocks, slots and dense piece are on the device at once: one being
                                     * copied in, one running, one being copied out */
#define PIPE_GROUP_MAX 8            /* chunks per launch, at most */
#define PIPE_MIN_MB  24             /* the smallest batch (MiB of the wider side) and ... */
#define PIPE_MIN_BLOCKS 256         /* ... the fewest blocks that take this route (see host_batch_pipelined) */

/* events */
enum { EV_IN = 0,                   /* [2] H2D of chunk k has left pinned piece k & 1 */
       EV_INS = 2,                  /* [PIPE_SLOTS] all of launch j's chunks are on the device */
       EV_RUN = 5,                  /* [PIPE_SLOTS] launch j (kernel, compaction, lengths to the host) is through */
       EV_OUT = 8,                  /* [3] D2H of chunk k is in pinned piece k % 3 */
       EV_OUTS = 11,                /* [PIPE_SLOTS] every chunk of launch j has been copied out of its dense piece */
       EV_COUNT = 14 };

typedef struct {
    /* the batch */
    uint8_t *out; size_t out_stride; uint32_t *out_len;
    const uint8_t *in; size_t in_stride; const uint32_t *in_len_each; size_t in_len;
    size_t d_in_stride, nblocks, chunk;
    /* this round's work: chunk numbers, or -1 */
    long fill, drain;
    uint8_t *pin_in, *pin_out;
    const uint64_t *drain_off;      /* drain: offsets of the chunk's blocks in pin_out (chunk + 1 entries) */
    /* the workers */
    pthread_t th[PIPE_MAX_THREADS];
    int nth, started;
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    unsigned gen;
    int pending, quit;
} pipe_t;

/* worker w's share of this round: its slice of the blocks to fill and of the blocks to drain */
static void pipe_do_share(const pipe_t *p, int w)
{
    if (p->fill >= 0) {
        const size_t b0 = (size_t)p->fill * p->chunk;
        const size_t nb = p->nblocks - b0 < p->chunk ? p->nblocks - b0 : p->chunk;
        const size_t lo = nb * (size_t)w / (size_t)p->nth, hi = nb * (size_t)(w + 1) / (size_t)p->nth;
        if (!p->in_len_each && p->in_stride == p->d_in_stride) {
            /* contiguous blocks of the device's own stride: one copy (the last block of the batch may be short of its stride) */
            if (hi > lo) {
                const size_t bytes = (hi - lo - 1) * p->d_in_stride + (b0 + hi == p->nblocks ? p->in_len : p->d_in_stride);
                memcpy(p->pin_in + lo * p->d_in_stride, p->in + (b0 + lo) * p->in_stride, bytes);
            }
        } else {
            for (size_t b = lo; b < hi; b++)
                memcpy(p->pin_in + b * p->d_in_stride, p->in + (b0 + b) * p->in_stride, p->in_len_each ? p->in_len_each[b0 + b] : p->in_len);
        }
    }
    if (p->drain >= 0) {
        const size_t b0 = (size_t)p->drain * p->chunk;
        const size_t nb = p->nblocks - b0 < p->chunk ? p->nblocks - b0 : p->chunk;
        const size_t lo = nb * (size_t)w / (size_t)p->nth, hi = nb * (size_t)(w + 1) / (size_t)p->nth;
        for (size_t b = lo; b < hi; b++)
            memcpy(p->out + (b0 + b) * p->out_stride, p->pin_out + p->drain_off[b], p->out_len[b0 + b]);
    }
}

typedef struct { pipe_t *p; int w; } pipe_arg_t;

static void *pipe_worker(void *argp)
{
    pipe_arg_t *a = (pipe_arg_t *)argp;
    pipe_t *p = a->p;
    const int w = a->w;
    unsigned seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->quit && p->gen == seen) pthread_cond_wait(&p->cv_work, &p->mu);
        if (p->quit) break;
        seen = p->gen;
        pthread_mutex_unlock(&p->mu);
        pipe_do_share(p, w);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

/* one round of host copies on all workers; the calling thread takes share 0 itself */
static void pipe_round(pipe_t *p)
{
    if (p->fill < 0 && p->drain < 0) return;
    pthread_mutex_lock(&p->mu);
    p->pending = p->started;
    p->gen++;
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    pipe_do_share(p, 0);
    pthread_mutex_lock(&p->mu);
    while (p->pending) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

static size_t round16(size_t v) { return (v + 15u) & ~(size_t)15u; }

LZS_HIDDEN int host_batch_pipelined(const char *who, launch_fn launch, uint8_t *out, size_t out_stride, uint32_t cap32, uint32_t *out_len,
                                    const uint8_t *in, size_t in_stride, const uint32_t *in_len_each, size_t in_len, size_t nblocks,
                                    int *taken)
{
    *taken = 0;
    const lzs_env_t *env = lzs_env();
    const size_t d_in_stride = round16(in_len ? in_len : 1), d_out_stride = round16(cap32 ? cap32 : 1);
    const size_t widest = d_in_stride > d_out_stride ? d_in_stride : d_out_stride;
    /* chunks of about 45 MiB of the wider side (640 blocks of 64 KiB), at least 64 blocks; worth it from four chunks on */
    size_t chunk = ((size_t)(env->pipe_chunk_mb > 0 ? env->pipe_chunk_mb : 46) << 20) / widest;
    chunk = chunk < 64 ? 64 : chunk & ~(size_t)63;
    /* From 24 MiB on (batches of streams up to 64 MiB of output are decompressed in segments before this is asked:
     * lzs_host.c); a batch of fewer than four chunks is cut into four all the same: pinned pieces and host threads beat
     * the runtime's path for pageable memory from there on (1024 blocks of 64 KiB: 9-28 ms -> 3.3; 384 blocks: 2.3 -> 2.0,
     * 512: 2.7 -> 2.2, equal at 256: profiles/r04/hostbatch_small_routes.txt). */
    const unsigned long long least = (unsigned long long)(env->pipe_min_mb > 0 ? env->pipe_min_mb : PIPE_MIN_MB) << 20;
    if (env->overlap_off || !cap32 || nblocks < PIPE_MIN_BLOCKS || (unsigned long long)nblocks * widest < least ||
        widest > ((size_t)4 << 20)) return LZS_OK;
    if (nblocks < 4 * chunk) chunk = ((nblocks + 3) / 4 + 63) & ~(size_t)63;
    const size_t K = (nblocks + chunk - 1) / chunk;
    size_t G = env->pipe_group > 0 && env->pipe_group <= PIPE_GROUP_MAX ? (size_t)env->pipe_group
             : launch == lzs_hip_launch_compress ? 2 : 8;                  /* chunks per launch ... */
    if (G > K) G = K;                                                       /* ... of the chunks there are (a 1024-block
                                                                             * decompress batch is ONE launch of four chunks, not a ring
                                                                             * of three launches of eight) */
    const size_t J = (K + G - 1) / G;                                       /* launches */
    const size_t slots = J < PIPE_SLOTS ? J : PIPE_SLOTS;                   /* launches' worth of device memory in the ring */
    if ((unsigned long long)G * chunk * d_out_stride > 0xF0000000ull) return LZS_OK;
    staging_t *st = staging_get();
    if (!st) return fail(LZS_E_NOMEM, "%s: out of host memory", who);
    *taken = 1;

    int rc = LZS_OK, e = 0;
    pipe_t P;
    pipe_arg_t args[PIPE_MAX_THREADS];
    uint64_t *doff[3] = {NULL, NULL, NULL};
    memset(&P, 0, sizeof P);
    P.out = out; P.out_stride = out_stride; P.out_len = out_len;
    P.in = in; P.in_stride = in_stride; P.in_len_each = in_len_each; P.in_len = in_len;
    P.d_in_stride = d_in_stride; P.nblocks = nblocks; P.chunk = chunk;
    P.fill = P.drain = -1;
    P.nth = env->copy_threads > 0 ? env->copy_threads : 4;
    if (P.nth > PIPE_MAX_THREADS) P.nth = PIPE_MAX_THREADS;
    pthread_mutex_init(&P.mu, NULL);
    pthread_cond_init(&P.cv_work, NULL);
    pthread_cond_init(&P.cv_done, NULL);

#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto done; } } while (0)
    for (size_t i = 0; i < 2 + PIPE_RUN; i++)
        if (!st->pipe_stream[i]) HIP_TRY(lzs_hip_stream_create(&st->pipe_stream[i]), "hipStreamCreate");
    for (size_t i = 0; i < EV_COUNT; i++)
        if (!st->pipe_event[i]) HIP_TRY(lzs_hip_event_create(&st->pipe_event[i]), "hipEventCreate");
    void *const s_in = st->pipe_stream[0], *const s_out = st->pipe_stream[1];
    void **const ev = st->pipe_event;
#define S_RUN(j)   (st->pipe_stream[2 + (j) % PIPE_RUN])
#define E_IN(k)    (ev[EV_IN + ((k) & 1)])
#define E_INS(j)   (ev[EV_INS + (j) % PIPE_SLOTS])
#define E_RUN(j)   (ev[EV_RUN + (j) % PIPE_SLOTS])
#define E_OUT(k)   (ev[EV_OUT + (k) % 3])
#define E_OUTS(j)  (ev[EV_OUTS + (j) % PIPE_SLOTS])

    /* device memory: a ring of PIPE_SLOTS launches' worth of blocks, slots, lengths and dense bytes (sizes that stay with
     * the thread between calls: no allocation per call) */
    void *d_in = NULL, *d_out = NULL, *d_len = NULL, *d_in_len = NULL, *d_dense = NULL, *d_offs = NULL;
    const size_t group = G * chunk;                                         /* blocks per launch */
    const size_t dense_piece = group * d_out_stride + 32;
    e = staging_reserve(st, BUF_IN, slots * group * d_in_stride, &d_in);
    if (!e) e = staging_reserve(st, BUF_OUT, slots * group * d_out_stride, &d_out);
    if (!e) e = staging_reserve(st, BUF_LEN, slots * sizeof(uint32_t) * group, &d_len);
    if (!e && in_len_each) e = staging_reserve(st, BUF_INLEN, sizeof(uint32_t) * nblocks, &d_in_len);
    if (!e) e = staging_reserve(st, BUF_KEEP, slots * dense_piece + 64, &d_dense);
    if (!e) e = staging_reserve(st, BUF_AUX, slots * sizeof(uint64_t) * (group + 1), &d_offs);
    /* (no room for the ring or the pinned pieces -- a small cgroup, many threads at once: nothing has been queued yet, the
     * batch takes the one-after-the-other route, which needs neither.  The runtime keeps a failed allocation as its "last
     * error" until somebody fetches it, and every launcher returns the last error after its launch: fetch it here, or the
     * route that is to save the call reports this out-of-memory as its own -- ADVICE r04) */
    if (e) { lzs_hip_clear_error(); *taken = 0; goto done; }
    uint8_t *pin_in[2], *pin_out[3], *pin_len = NULL;
    const size_t len_piece = sizeof(uint32_t) * group + sizeof(uint64_t);
    for (int i = 0; i < 2 && !e; i++) e = staging_pin_reserve(st, PIPE_IN(i), chunk * d_in_stride, &pin_in[i]);
    for (int i = 0; i < 3 && !e; i++) e = staging_pin_reserve(st, PIPE_OUT(i), chunk * d_out_stride, &pin_out[i]);
    if (!e) e = staging_pin_reserve(st, PIN_LENGTHS, slots * len_piece, &pin_len);
    if (e) { lzs_hip_clear_error(); *taken = 0; goto done; }
    for (int i = 0; i < 3; i++) doff[i] = (uint64_t *)malloc(sizeof(uint64_t) * (chunk + 1));
    if (!doff[0] || !doff[1] || !doff[2]) { *taken = 0; goto done; }

    for (int w = 1; w < P.nth; w++) {
        args[w].p = &P; args[w].w = w;
        if (pthread_create(&P.th[w], NULL, pipe_worker, &args[w]) != 0) { P.nth = w; break; }
        P.started++;
    }
    if (in_len_each) {
        HIP_TRY(lzs_hip_h2d(d_in_len, in_len_each, sizeof(uint32_t) * nblocks, s_in), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_stream_sync(s_in), "hipStreamSynchronize");
    }

    const int debug = env->stream_debug, trace = env->stream_debug && env->pipe_trace;
#define TRACE(...) do { if (trace) { fprintf(stderr, "  %8.2f ms  ", now_ms() - t_all); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
    double t_copy = 0, t_wait = 0, t_queue = 0, t_mark = 0, t_all = debug ? now_ms() : 0;
#define MARK() (t_mark = debug ? now_ms() : 0)
#define SINCE(acc) do { if (debug) { const double n_ = now_ms(); acc += n_ - t_mark; t_mark = n_; } } while (0)
    /* The four hands of the pipeline, each a counter of what it has done: chunks filled + copied in, launches queued,
     * chunks whose copy out is queued, chunks drained. */
    size_t n_fill = 0, n_run = 0, n_fin = 0, n_drain = 0, n_done = 0;       /* (n_done: launches the host has seen finished) */
    size_t j_known = (size_t)-1;                        /* the launch whose lengths the host has read */
    uint64_t chunk_at[PIPE_GROUP_MAX + 1];              /* of that launch: where its chunks begin in the dense piece */
    unsigned long rounds = 0;
    while (n_drain < K) {
        int progressed = 0;
        MARK();
        /* ---- launches whose chunks are all on their way in, and whose dense piece and length slot are free again
         * (launch j - PIPE_RUN used them: all its chunks must have been handed to the copy-out stream) */
        while (n_run < J && n_fill >= ((n_run + 1) * G < K ? (n_run + 1) * G : K) &&
               (n_run < PIPE_SLOTS || n_fin >= ((n_run - PIPE_SLOTS + 1) * G < K ? (n_run - PIPE_SLOTS + 1) * G : K))) {
            const size_t j = n_run, q = j % PIPE_SLOTS, b0 = j * group, nb = nblocks - b0 < group ? nblocks - b0 : group;
            uint8_t *const dense = (uint8_t *)d_dense + q * dense_piece;
            uint64_t *const offs = (uint64_t *)d_offs + q * (group + 1);
            uint8_t *const lens = pin_len + q * len_piece;
            uint8_t *const c_in = (uint8_t *)d_in + q * group * d_in_stride, *const c_out = (uint8_t *)d_out + q * group * d_out_stride;
            uint32_t *const c_len = (uint32_t *)d_len + q * group;
            HIP_TRY(lzs_hip_stream_wait_event(S_RUN(j), E_INS(j)), "hipStreamWaitEvent");
            HIP_TRY(launch(c_out, d_out_stride, cap32, c_len, c_in, d_in_stride,
                           d_in_len ? (const uint32_t *)d_in_len + b0 : NULL, (uint32_t)in_len, (uint32_t)nb, S_RUN(j)), who);
            if (j >= PIPE_SLOTS) HIP_TRY(lzs_hip_stream_wait_event(S_RUN(j), E_OUTS(j)), "hipStreamWaitEvent");   /* (the dense piece is read out) */
            HIP_TRY(lzs_hip_launch_compact(dense, offs, c_out, d_out_stride, c_len, (uint32_t)nb, S_RUN(j)), who);
            /* (the lengths and the byte total reach the host by a kernel, not by a copy engine: see lzs_hip_words_to_host) */
            HIP_TRY(lzs_hip_words_to_host((uint32_t *)lens, c_len, nb, S_RUN(j)), who);
            HIP_TRY(lzs_hip_words_to_host((uint32_t *)(lens + sizeof(uint32_t) * group), (const uint32_t *)(offs + nb), 2, S_RUN(j)), who);
            HIP_TRY(lzs_hip_event_record(E_RUN(j), S_RUN(j)), "hipEventRecord");
            TRACE("launch %zu queued (%zu blocks)", j, nb);
            n_run++;
            progressed = 1;
        }
        /* ---- chunks of a finished launch: exactly their bytes to a free pinned piece (three: one being drained, one
         * being copied into, one waiting) */
        while (n_fin < K && n_fin / G < n_run && n_fin - n_drain < 3) {
            const size_t k = n_fin, j = k / G, b0 = k * chunk, nb = nblocks - b0 < chunk ? nblocks - b0 : chunk;
            if (j != j_known) {
                const int ready = lzs_hip_event_done(E_RUN(j));
                if (ready < 0) HIP_TRY(-ready, "hipEventQuery");
                if (!ready) break;
                /* the launch is through: its lengths, and where each of its chunks begins in the dense piece */
                const size_t g0 = j * group, gn = nblocks - g0 < group ? nblocks - g0 : group;
                const uint8_t *const lp = pin_len + (j % PIPE_SLOTS) * len_piece;
                const uint32_t *lens = (const uint32_t *)lp;
                uint64_t total = 0, at = 0;
                memcpy(&total, lp + sizeof(uint32_t) * group, sizeof total);
                for (size_t b = 0; b < gn; b++) {
                    if (b % chunk == 0) chunk_at[b / chunk] = at;
                    out_len[g0 + b] = lens[b];
                    at += lens[b];
                }
                chunk_at[(gn + chunk - 1) / chunk] = at;
                if (at != total || total > (uint64_t)group * d_out_stride) {
                    rc = fail(LZS_E_HIP, "%s: inconsistent lengths from the device (launch %zu: %llu bytes by the lengths, %llu by the offsets)",
                              who, j, (unsigned long long)at, (unsigned long long)total);
                    goto done;
                }
                j_known = j;
                n_done = j + 1;
                TRACE("launch %zu seen finished", j);
            }
            const size_t c = k - j * G;                 /* the chunk's number inside its launch */
            uint64_t *o = doff[k % 3], at = 0;
            for (size_t b = 0; b < nb; b++) { o[b] = at; at += out_len[b0 + b]; }
            o[nb] = at;
            HIP_TRY(lzs_hip_d2h(pin_out[k % 3], (const uint8_t *)d_dense + (j % PIPE_SLOTS) * dense_piece + chunk_at[c],
                                (size_t)(chunk_at[c + 1] - chunk_at[c]), s_out), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_event_record(E_OUT(k), s_out), "hipEventRecord");
            if (c + 1 == G || k + 1 == K) HIP_TRY(lzs_hip_event_record(E_OUTS(j), s_out), "hipEventRecord");
            TRACE("chunk %zu: copy out queued (%llu bytes)", k, (unsigned long long)(chunk_at[c + 1] - chunk_at[c]));
            n_fin++;
            progressed = 1;
        }
        SINCE(t_queue);
        /* ---- the host threads: the next chunk in (its pinned piece is free once the copy before last has left it), and
         * the oldest chunk whose bytes have arrived out */
        /* (a chunk goes where launch j - PIPE_SLOTS read its blocks from: that launch must be through) */
        const int do_fill_possible = n_fill < K && (n_fill / G < PIPE_SLOTS || n_done + PIPE_SLOTS > n_fill / G);
        int do_fill = do_fill_possible, do_drain = 0;
        if (do_fill && n_fill >= 2) {
            const int ready = lzs_hip_event_done(E_IN(n_fill));
            if (ready < 0) HIP_TRY(-ready, "hipEventQuery");
            do_fill = ready;
        }
        if (n_drain < n_fin) {
            const int ready = lzs_hip_event_done(E_OUT(n_drain));
            if (ready < 0) HIP_TRY(-ready, "hipEventQuery");
            do_drain = ready;
        }
        if (do_fill || do_drain) {
            P.fill = do_fill ? (long)n_fill : -1;
            P.drain = do_drain ? (long)n_drain : -1;
            P.pin_in = pin_in[n_fill & 1];
            P.pin_out = pin_out[n_drain % 3];
            P.drain_off = doff[n_drain % 3];
            pipe_round(&P);
            SINCE(t_copy);
            rounds++;
            TRACE("host copies done: fill %ld, drain %ld", P.fill, P.drain);
            if (do_fill) {
                const size_t k = n_fill, b0 = k * chunk, nb = nblocks - b0 < chunk ? nblocks - b0 : chunk;
                uint8_t *const to = (uint8_t *)d_in + (((k / G) % PIPE_SLOTS) * group + (k % G) * chunk) * d_in_stride;
                HIP_TRY(lzs_hip_h2d(to, pin_in[k & 1], nb * d_in_stride, s_in), "hipMemcpy H2D");
                HIP_TRY(lzs_hip_event_record(E_IN(k), s_in), "hipEventRecord");
                if ((k + 1) % G == 0 || k + 1 == K) HIP_TRY(lzs_hip_event_record(E_INS(k / G), s_in), "hipEventRecord");
                n_fill++;
            }
            if (do_drain) n_drain++;
            SINCE(t_queue);
            progressed = 1;
        }
        if (!progressed) {
            /* nothing is ready: wait for what the pipeline needs next -- the copy out that is under way, else the
             * launch that is on the device; but first of all the copy in whose pinned piece the next chunk wants */
            if (do_fill_possible) HIP_TRY(lzs_hip_event_sync(E_IN(n_fill)), "hipEventSynchronize");     /* (the shortest wait first) */
            else if (n_drain < n_fin) HIP_TRY(lzs_hip_event_sync(E_OUT(n_drain)), "hipEventSynchronize");
            else if (n_fin < K && n_fin / G < n_run) HIP_TRY(lzs_hip_event_sync(E_RUN(n_fin / G)), "hipEventSynchronize");
            else { rc = fail(LZS_E_HIP, "%s: the pipeline has nothing to wait for (filled %zu, launched %zu, copied out %zu, drained %zu of %zu)",
                             who, n_fill, n_run, n_fin, n_drain, K); goto done; }
            SINCE(t_wait);
        }
    }
    if (debug)
        fprintf(stderr, "liblzs pipeline: %s, %zu blocks in %zu chunks of %zu, %zu chunks a launch, %d host threads: %.1f ms = host copies %.1f "
                        "(%lu rounds) + waiting %.1f + queueing %.1f\n",
                who, nblocks, K, chunk, G, P.started + 1, now_ms() - t_all, t_copy, rounds, t_wait, t_queue);
#undef MARK
#undef SINCE
#undef HIP_TRY
#undef S_RUN
#undef E_IN
#undef E_INS
#undef E_RUN
#undef E_OUT
#undef E_OUTS

done:
    if (rc != LZS_OK)                                       /* nothing of ours may still be in flight */
        for (int i = 0; i < PIPE_STREAMS; i++) if (st->pipe_stream[i]) lzs_hip_stream_sync(st->pipe_stream[i]);
    pthread_mutex_lock(&P.mu);
    P.quit = 1;
    pthread_cond_broadcast(&P.cv_work);
    pthread_mutex_unlock(&P.mu);
    for (int w = 1; w <= P.started; w++) pthread_join(P.th[w], NULL);
    pthread_mutex_destroy(&P.mu); pthread_cond_destroy(&P.cv_work); pthread_cond_destroy(&P.cv_done);
    free(doff[0]); free(doff[1]); free(doff[2]);
    staging_trim(st);
    return rc;
}
