/*
 * lzs_host.c -- the C host side of liblzs (MI355X build): argument checking, error
 * reporting, host<->device staging.  All codec work happens in lzs_kernels.hip, reached
 * through the extern-"C" shim in lzs_hip_shim.h.  There is deliberately NO CPU codec
 * here: without a HIP device every entry point fails loudly.
 *
 * Public surface: include/lzs/lzs.h (the reference's one-shot calls,
 * c/src/liblzs/lzs.h:218,229) and include/lzs/lzs_batch.h (additive batch calls).
 */
#include "lzs_internal.h"

LZS_HIDDEN _Thread_local char tls_error[512];

LZS_HIDDEN int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_error, sizeof(tls_error), fmt, ap);
    va_end(ap);
    return code;
}

LZS_HIDDEN int hip_fail(int hip_error, const char *what)
{
    return fail(LZS_E_HIP, "%s: %s", what, lzs_hip_strerror(hip_error));
}

const char *lzs_last_error(void) { return tls_error; }

LZS_HIDDEN int require_device(void)
{
    /* (asked once per process: a device that was there and answered the ordering check does not go away, and a call on
     * the small calls' host route -- 5 us of work at the reference tools' 512-byte reads -- should not pay two runtime
     * calls to hear it again; a device that fails later fails the HIP call that meets it) */
    static int device_ok;
    if (__atomic_load_n(&device_ok, __ATOMIC_ACQUIRE)) return LZS_OK;
    int n = 0;
    int e = lzs_hip_device_count(&n);
    if (e != 0 || n <= 0)
        return fail(LZS_E_NO_DEVICE, "no HIP device available (%s); liblzs has no CPU codec",
                    e ? lzs_hip_strerror(e) : "device count is 0");
    /* the device is asked how it orders same-address LDS exchanges HERE, on a stream of its own,
     * once per device and process: the first asynchronous launch then finds the answer and neither
     * allocates nor waits */
    int mode = 0;
    if ((e = lzs_hip_chain_mode(NULL, &mode)) != 0) return hip_fail(e, "LDS ordering check");
    __atomic_store_n(&device_ok, 1, __ATOMIC_RELEASE);
    return LZS_OK;
}

int lzs_backend_info(char *buf, size_t cap)
{
    if (!buf || cap == 0) return fail(LZS_E_ARG, "lzs_backend_info: no buffer");
    int rc = require_device();
    if (rc != LZS_OK) { snprintf(buf, cap, "%s", tls_error); return rc; }
    int e = lzs_hip_describe(buf, cap);
    if (e) { snprintf(buf, cap, "%s", lzs_hip_strerror(e)); return hip_fail(e, "hipGetDeviceProperties"); }
    return LZS_OK;
}

/* ------------------------------------------------------------------ device batches */

static int check_batch(const char *who, const void *out_len, const void *in, size_t in_len,
                       size_t nblocks)
{
    if (nblocks == 0) return LZS_OK;
    if (!out_len) return fail(LZS_E_ARG, "%s: out_len is NULL", who);
    if (!in && in_len) return fail(LZS_E_ARG, "%s: input is NULL", who);
    if (in_len > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "%s: block of %zu bytes exceeds LZS_BLOCK_MAX", who, in_len);
    if (nblocks > 0x7FFFFFFFu) return fail(LZS_E_ARG, "%s: too many blocks (%zu)", who, nblocks);
    return LZS_OK;
}

static int device_batch(const char *who, launch_fn launch, void *d_out, size_t out_stride,
                        size_t out_cap, uint32_t *d_out_len, const void *d_in, size_t in_stride,
                        const uint32_t *d_in_len, size_t in_len, size_t nblocks, void *stream)
{
    int rc = check_batch(who, d_out_len, d_in, in_len, nblocks);
    if (rc != LZS_OK || nblocks == 0) return rc;
    if (!d_out && out_cap) return fail(LZS_E_ARG, "%s: output is NULL", who);
    /* the lengths written are not the lengths read: a compress launch leaves every block's class in d_out_len[] before any
     * workgroup reads d_in_len[] */
    if (d_in_len && (const void *)d_in_len == (const void *)d_out_len) return fail(LZS_E_ARG, "%s: d_out_len and d_in_len are the same array", who);
    uint32_t cap32 = out_cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)out_cap;
    int e = launch(d_out, out_stride, cap32, d_out_len, d_in, in_stride, d_in_len,
                   (uint32_t)in_len, (uint32_t)nblocks, stream);
    return e ? hip_fail(e, who) : LZS_OK;
}

int lzs_compress_batch_device(void *d_out, size_t out_stride, size_t out_cap, uint32_t *d_out_len,
                              const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                              size_t in_len, size_t nblocks, void *hip_stream)
{
    return device_batch("lzs_compress_batch_device", lzs_hip_launch_compress, d_out, out_stride,
                        out_cap, d_out_len, d_in, in_stride, d_in_len, in_len, nblocks, hip_stream);
}

int lzs_decompress_batch_device(void *d_out, size_t out_stride, size_t out_cap, uint32_t *d_out_len,
                                const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                                size_t in_len, size_t nblocks, void *hip_stream)
{
    return device_batch("lzs_decompress_batch_device", lzs_hip_launch_decompress, d_out, out_stride,
                        out_cap, d_out_len, d_in, in_stride, d_in_len, in_len, nblocks, hip_stream);
}

int lzs_decompress_batch_device_sync(void *d_out, size_t out_stride, size_t out_cap, uint32_t *out_len,
                                     const void *d_in, size_t in_stride, const uint32_t *in_len_each,
                                     size_t in_len, size_t nblocks)
{
    const char *who = "lzs_decompress_batch_device_sync";
    int rc = check_batch(who, out_len, d_in, in_len, nblocks);
    if (rc != LZS_OK || nblocks == 0) return rc;
    if (!d_out && out_cap) return fail(LZS_E_ARG, "%s: output is NULL", who);
    size_t total_in = 0, longest = 0;
    for (size_t b = 0; b < nblocks; b++) {
        const size_t len = in_len_each ? in_len_each[b] : in_len;
        total_in += len;
        if (len > longest) longest = len;
    }
    if (longest > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "%s: block exceeds LZS_BLOCK_MAX", who);
    if ((rc = require_device()) != LZS_OK) return rc;
    staging_t *st = staging_get();
    if (!st) return fail(LZS_E_NOMEM, "%s: out of host memory", who);
    int e = 0;
    if (!st->stream && (e = lzs_hip_stream_create(&st->stream))) return hip_fail(e, "hipStreamCreate");
    const uint32_t cap32 = out_cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)out_cap;
    void *d_len = NULL, *d_in_len = NULL;
    if ((e = staging_reserve(st, BUF_LEN, sizeof(uint32_t) * nblocks, &d_len)))
        return fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e));
    if (cap32 && nblocks <= BATCH_SEG_MAX_BLOCKS && total_in >= STREAM_DEC_MIN && total_in / nblocks >= 1024u &&
        (unsigned long long)nblocks * out_stride <= BATCH_SEG_MAX_EXTENT && (unsigned long long)nblocks * in_stride < 0xF0000000ull &&
        cap32 <= out_stride) {
        rc = batch_decompress_segments(st, st->stream, who, d_out, out_stride, cap32, out_len, (uint32_t *)d_len, d_in, in_stride,
                                       in_len_each, (uint32_t)in_len, nblocks);
    } else {
        if (in_len_each) {
            if ((e = staging_reserve(st, BUF_INLEN, sizeof(uint32_t) * nblocks, &d_in_len)))
                return fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e));
            if ((e = lzs_hip_h2d(d_in_len, in_len_each, sizeof(uint32_t) * nblocks, st->stream))) return hip_fail(e, "hipMemcpy H2D");
        }
        e = lzs_hip_launch_decompress(d_out, out_stride, cap32, (uint32_t *)d_len, d_in, in_stride, (const uint32_t *)d_in_len,
                                      (uint32_t)in_len, (uint32_t)nblocks, st->stream);
        if (!e) e = lzs_hip_d2h(out_len, d_len, sizeof(uint32_t) * nblocks, st->stream);
        if (!e) e = lzs_hip_stream_sync(st->stream);
        if (e) rc = hip_fail(e, who);
    }
    staging_trim(st);
    return rc;
}

int lzs_compact_device(void *d_dense, uint64_t *d_offsets, const void *d_slots, size_t slot_stride,
                       const uint32_t *d_len, size_t nblocks, void *hip_stream)
{
    if (!d_offsets) return fail(LZS_E_ARG, "lzs_compact_device: d_offsets is NULL");
    if (nblocks && (!d_slots || !d_len || !d_dense))
        return fail(LZS_E_ARG, "lzs_compact_device: NULL buffer");
    if (nblocks > 0x7FFFFFFFu) return fail(LZS_E_ARG, "lzs_compact_device: too many blocks");
    int e = lzs_hip_launch_compact(d_dense, d_offsets, d_slots, slot_stride, d_len,
                                   (uint32_t)nblocks, hip_stream);
    return e ? hip_fail(e, "lzs_compact_device") : LZS_OK;
}

/* ------------------------------------------------- per-thread staging (host batches) */
/* Each host thread keeps one HIP stream and grow-only device buffers for the host-buffer
 * entry points, so a small one-shot call costs copies + a launch, not allocations.  What a thread keeps between
 * calls is at most keep_max() bytes of device memory IN SUM (staging_trim, after every call); all of it goes back when
 * the thread exits or calls lzs_release_thread_cache().  (The reference keeps nothing after return -- it has nothing
 * to keep: lzs.h:218,229 work on the caller's two buffers and ~12 KiB of stack; SURVEY.md 8(b) Ownership.) */
#define KEEP_MAX ((size_t)640 << 20)


static pthread_key_t  staging_key;
static pthread_once_t staging_once = PTHREAD_ONCE_INIT;

static void staging_destroy(void *p)
{
    staging_t *st = (staging_t *)p;
    if (!st) return;
    for (int i = 0; i < BUF_COUNT; i++)
        if (st->buf[i]) lzs_hip_free(st->buf[i]);
    if (st->stream) lzs_hip_stream_destroy(st->stream);
    if (st->host_box) lzs_hip_host_free(st->host_box);     /* pinned (lzs_incremental.c) */
    if (st->host_tab) lzs_hip_host_free(st->host_tab);
    for (int i = 0; i < PIPE_STREAMS; i++) if (st->pipe_stream[i]) lzs_hip_stream_destroy(st->pipe_stream[i]);
    for (int i = 0; i < PIPE_EVENTS; i++) if (st->pipe_event[i]) lzs_hip_event_destroy(st->pipe_event[i]);
    for (int i = 0; i < 6; i++) if (st->pin[i]) lzs_hip_host_free(st->pin[i]);
    hostcodec_free(st->hostcodec);
    free(st);
}

static void staging_make_key(void) { pthread_key_create(&staging_key, staging_destroy); }

/* Everything the calling thread's earlier calls left behind -- device staging, its streams and events, pinned pieces, the
 * host route's tables -- given back now instead of when the thread exits.  The thread's next call starts from nothing. */
void lzs_release_thread_cache(void)
{
    pthread_once(&staging_once, staging_make_key);
    staging_t *st = (staging_t *)pthread_getspecific(staging_key);
    if (!st) return;
    if (st->stream) lzs_hip_stream_sync(st->stream);          /* (nothing of this thread's is in flight between its calls; a failed call may have left work) */
    pthread_setspecific(staging_key, NULL);
    staging_destroy(st);
}

LZS_HIDDEN staging_t *staging_get(void)
{
    pthread_once(&staging_once, staging_make_key);
    staging_t *st = (staging_t *)pthread_getspecific(staging_key);
    if (!st) {
        st = (staging_t *)calloc(1, sizeof(*st));
        if (st) pthread_setspecific(staging_key, st);
    }
    return st;
}

/* 0 on success, else a hipError_t */
LZS_HIDDEN int staging_reserve(staging_t *st, int which, size_t bytes, void **out)
{
    if (st->cap[which] < bytes) {
        if (st->buf[which]) { lzs_hip_free(st->buf[which]); st->buf[which] = NULL; st->cap[which] = 0; }
        size_t want = (bytes + 65535u) & ~(size_t)65535u;
        /* (tests: LZS_STAGING_FAIL_MB=n makes a reservation above n MiB fail the way a full device fails it -- by a real
         * hipMalloc that cannot succeed, so that the runtime's sticky last error is set as it would be) */
        const int fail_mb = lzs_env()->staging_fail_mb;
        if (fail_mb > 0 && want > ((size_t)fail_mb << 20)) want = (size_t)1 << 60;
        int e = lzs_hip_malloc(&st->buf[which], want);
        if (e) { st->buf[which] = NULL; return e; }
        st->cap[which] = want;
    }
    *out = st->buf[which];
    return 0;
}

/* ---- the environment, once per process (lzs_internal.h) */
static lzs_env_t g_env;
static int g_env_dev;                                   /* LZS_DEV_ENV: read afresh on every call (tests) */
static pthread_once_t g_env_once = PTHREAD_ONCE_INIT;

static const char *get(const char *name) { return getenv(name); }      /* the host library's one reader of the environment */

static void env_read(lzs_env_t *e)
{
    const char *v;
    memset(e, 0, sizeof *e);
    v = get("LZS_KEEP_MAX_MB");
    { const unsigned long mb = v ? strtoul(v, NULL, 10) : 0; e->keep_max = (size_t)mb << 20; }   /* (0: keep_max() below has the default) */
    e->one_wave = get("LZS_ONE_WAVE") != NULL;
    e->one_workgroup = get("LZS_ONE_WORKGROUP") != NULL;
    e->force_stream = get("LZS_FORCE_STREAM") != NULL;
    v = get("LZS_STREAM_SEG"); e->stream_seg = v ? (uint32_t)strtoul(v, NULL, 10) : 0;
    v = get("LZS_DEC_SEG");    e->dec_seg = v ? (uint32_t)strtoul(v, NULL, 10) : 0;
    e->stream_debug = get("LZS_STREAM_DEBUG") != NULL;
    e->no_marks = get("LZS_NO_MARKS") != NULL;
    e->no_ones = get("LZS_NO_ONES") != NULL;
    e->verify_scan = get("LZS_VERIFY_SCAN") != NULL;
    e->no_tails = get("LZS_NO_TAILS") != NULL;
    e->no_chunks = get("LZS_NO_CHUNKS") != NULL;
    e->overlap_off = get("LZS_HOST_SERIAL") != NULL;
    v = get("LZS_COPY_THREADS"); e->copy_threads = v ? (int)strtol(v, NULL, 10) : 0;
    e->pipe_trace = get("LZS_PIPE_TRACE") != NULL;
    v = get("LZS_PIPE_GROUP"); e->pipe_group = v ? (int)strtol(v, NULL, 10) : 0;
    v = get("LZS_PIPE_CHUNK_MB"); e->pipe_chunk_mb = v ? (int)strtol(v, NULL, 10) : 0;
    v = get("LZS_BATCH_SEG_MB"); e->batch_seg_mb = v ? (int)strtol(v, NULL, 10) : 0;
    v = get("LZS_PIPE_MIN_MB"); e->pipe_min_mb = v ? (int)strtol(v, NULL, 10) : 0;
    v = get("LZS_STAGING_FAIL_MB"); e->staging_fail_mb = v ? (int)strtol(v, NULL, 10) : 0;
    v = get("LZS_ROUTE"); e->route = !v ? LZS_ROUTE_AUTO : (v[0] == 'd' ? LZS_ROUTE_DEVICE : (v[0] == 'h' ? LZS_ROUTE_HOST : LZS_ROUTE_AUTO));
}

/* (g_env_dev is written here and nowhere else: after pthread_once every thread only reads it -- ADVICE r04) */
static void env_once(void) { env_read(&g_env); g_env_dev = get("LZS_DEV_ENV") != NULL; }

LZS_HIDDEN const lzs_env_t *lzs_env(void)
{
    static _Thread_local lzs_env_t fresh;
    pthread_once(&g_env_once, env_once);
    if (!g_env_dev) return &g_env;
    env_read(&fresh);
    return &fresh;
}

/* What a thread may keep between calls, device bytes IN SUM over its staging buffers (round 6; per buffer until then, which
 * let one thread sit on 7 x 4.5 GiB): the largest buffers go first until the rest fits.  The default is 1/32 of the device's
 * memory and no less than 640 MiB -- 9 GiB on a 288 GiB MI355X, so that what a 1 GiB stream's decode uses (its table of
 * origins, 4 GiB, the stream and the output: 5.7 GiB) stays: a hipMalloc of gigabytes is not a cheap call, 0.2 ms most of
 * the time and 0.24-1.3 s now and then (profiles/r05/malloc_time.txt, stream_decode_glitch.txt), which a program that decodes
 * large streams over and over would pay again and again.  LZS_KEEP_MAX_MB overrides it (include/lzs/lzs_batch.h). */
static size_t keep_max(void)
{
    const size_t from_env = lzs_env()->keep_max;
    if (from_env) return from_env;
    static size_t dflt;                                  /* (the same value whichever thread gets there first) */
    size_t d = __atomic_load_n(&dflt, __ATOMIC_RELAXED);
    if (!d) {
        size_t total = 0;
        d = KEEP_MAX;
        if (lzs_hip_total_memory(&total) == 0 && total / 32 > d) d = total / 32;
        __atomic_store_n(&dflt, d, __ATOMIC_RELAXED);
    }
    return d;
}

LZS_HIDDEN int staging_pin_reserve(staging_t *st, int which, size_t bytes, uint8_t **out)
{
    if (st->pin_cap[which] < bytes) {
        if (st->pin[which]) { lzs_hip_host_free(st->pin[which]); st->pin[which] = NULL; st->pin_cap[which] = 0; }
        const size_t want = (bytes + 65535u) & ~(size_t)65535u;
        /* Pieces only the copy engines and the host touch are non-coherent (56 against 35-39 GB/s on a quiet device); the
         * piece a KERNEL writes (PIN_LENGTHS: lzs_hip_words_to_host) is coherent host memory -- HIP does not promise that a
         * kernel's stores to non-coherent host memory are visible to the host at an event. */
        const int e = which == PIN_LENGTHS ? lzs_hip_host_malloc(&st->pin[which], want) : lzs_hip_host_malloc_staging(&st->pin[which], want);
        if (e) { st->pin[which] = NULL; return e; }
        st->pin_cap[which] = want;
    }
    *out = (uint8_t *)st->pin[which];
    return 0;
}

LZS_HIDDEN void *staging_host_tables(staging_t *st, size_t bytes)
{
    if (st->host_tab_cap < bytes) {
        if (st->host_tab) { lzs_hip_host_free(st->host_tab); st->host_tab = NULL; st->host_tab_cap = 0; }
        const size_t want = (bytes + 65535u) & ~(size_t)65535u;
        if (lzs_hip_host_malloc(&st->host_tab, want)) { st->host_tab = NULL; return NULL; }
        st->host_tab_cap = want;
    }
    return st->host_tab;
}

LZS_HIDDEN void staging_trim(staging_t *st)
{
    const size_t limit = keep_max();
    size_t sum = 0;
    for (int i = 0; i < BUF_COUNT; i++) sum += st->cap[i];
    while (sum > limit) {                                     /* the largest first: the fewest reallocations for the bytes given back */
        int big = 0;
        for (int i = 1; i < BUF_COUNT; i++) if (st->cap[i] > st->cap[big]) big = i;
        sum -= st->cap[big];
        lzs_hip_free(st->buf[big]); st->buf[big] = NULL; st->cap[big] = 0;
    }
    /* the pinned pieces of the overlapped host batches (up to 6 x 46 MiB x the group) and the tables of the stream paths
     * follow the same limit, on the host's side */
    size_t pinned = st->host_tab_cap;
    for (int i = 0; i < 6; i++) pinned += st->pin_cap[i];
    if (pinned > limit) {
        for (int i = 0; i < 6; i++) if (st->pin[i]) { lzs_hip_host_free(st->pin[i]); st->pin[i] = NULL; st->pin_cap[i] = 0; }
        if (st->host_tab) { lzs_hip_host_free(st->host_tab); st->host_tab = NULL; st->host_tab_cap = 0; }
    }
}

/* -------------------------------------------------------------------- host batches */
int lzs_hip_launch_decompress(void *, size_t, uint32_t, uint32_t *, const void *, size_t, const uint32_t *, uint32_t, uint32_t, void *);

/* Stage host buffers through this thread's device memory.  Blocks are packed on the device
 * with 16-byte-aligned strides so the kernels take their aligned paths, whatever the
 * caller's strides are. */
static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int host_batch(const char *who, launch_fn launch, uint8_t *out, size_t out_stride,
                      size_t out_cap, uint32_t *out_len, const uint8_t *in, size_t in_stride,
                      const uint32_t *in_len_each, size_t in_len, size_t nblocks)
{
    int rc = check_batch(who, out_len, in, in_len, nblocks);
    if (rc != LZS_OK || nblocks == 0) return rc;
    if (!out && out_cap) return fail(LZS_E_ARG, "%s: output is NULL", who);
    if (in_len_each) {
        in_len = 0;
        for (size_t b = 0; b < nblocks; b++) {
            if (in_len_each[b] > in_len) in_len = in_len_each[b];
        }
        if (in_len > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "%s: block exceeds LZS_BLOCK_MAX", who);
    }
    if ((rc = require_device()) != LZS_OK) return rc;

    const uint32_t cap32 = out_cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)out_cap;
    /* a large batch: copy in, run and copy back overlapped, chunk by chunk (lzs_pipeline.c) */
    const unsigned long long seg_extent = lzs_env()->batch_seg_mb > 0 ? (unsigned long long)lzs_env()->batch_seg_mb << 20 : BATCH_SEG_MAX_EXTENT;
    const int by_segments = launch == lzs_hip_launch_decompress && cap32 && !lzs_env()->one_wave && nblocks <= BATCH_SEG_MAX_BLOCKS &&
                            (unsigned long long)nblocks * round_up(cap32, 16) <= seg_extent;
    if (!by_segments) {
        int taken = 0;
        rc = host_batch_pipelined(who, launch, out, out_stride, cap32, out_len, in, in_stride, in_len_each, in_len, nblocks, &taken);
        if (taken || rc != LZS_OK) return rc;
    }
    const size_t d_in_stride = round_up(in_len ? in_len : 1, 16);
    const size_t d_out_stride = round_up(cap32 ? cap32 : 1, 16);
    void *stream = NULL, *d_in = NULL, *d_out = NULL, *d_len = NULL, *d_in_len = NULL;
    uint8_t *bounce[2] = {NULL, NULL};
    int e = 0;
    rc = LZS_OK;
    staging_t *st = staging_get();
    if (!st) return fail(LZS_E_NOMEM, "%s: out of host memory", who);

#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto done; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    stream = st->stream;
    e = staging_reserve(st, BUF_IN, d_in_stride * nblocks, &d_in);
    if (!e) e = staging_reserve(st, BUF_OUT, d_out_stride * nblocks, &d_out);
    if (!e) e = staging_reserve(st, BUF_LEN, sizeof(uint32_t) * nblocks, &d_len);
    if (!e && in_len_each) e = staging_reserve(st, BUF_INLEN, sizeof(uint32_t) * nblocks, &d_in_len);
    if (e) { rc = fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto done; }

    const int debug = lzs_env()->stream_debug;
    double t0 = debug ? now_ms() : 0, t1;
#define STAGE(name) do { if (debug) { lzs_hip_stream_sync(stream); t1 = now_ms(); fprintf(stderr, "liblzs %s (%zu blocks): %s %.2f ms\n", who, nblocks, name, t1 - t0); t0 = t1; } } while (0)
    if (in_stride == d_in_stride && !in_len_each) {
        HIP_TRY(lzs_hip_h2d(d_in, in, d_in_stride * (nblocks - 1) + in_len, stream), "hipMemcpy H2D");
    } else if (nblocks < 16) {
        for (size_t b = 0; b < nblocks; b++) {
            size_t len_b = in_len_each ? in_len_each[b] : in_len;
            HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + b * d_in_stride, in + b * in_stride, len_b, stream),
                    "hipMemcpy H2D");
        }
    } else {
        /* ragged or strided blocks: laid out at the device stride by the CPU, 32 MiB at a time
         * (thousands of small copies cost more than the data) */
        size_t per = ((size_t)32 << 20) / d_in_stride;
        if (per == 0) per = 1;
        /* (in one of the thread's pinned pieces: no pages to fault in call after call, and the copy engine reads it
         * directly; plain memory only if there is no pinned memory to be had) */
        const size_t lay = (nblocks < per ? nblocks : per) * d_in_stride;
        uint8_t *piece_in = NULL;
        if (staging_pin_reserve(st, 0, lay, &piece_in)) {
            bounce[0] = (uint8_t *)malloc(lay);
            if (!bounce[0]) { rc = fail(LZS_E_NOMEM, "%s: out of host memory", who); goto done; }
            piece_in = bounce[0];
        }
        for (size_t b0 = 0; b0 < nblocks; b0 += per) {
            const size_t nb = nblocks - b0 < per ? nblocks - b0 : per;
            for (size_t b = 0; b < nb; b++)
                memcpy(piece_in + b * d_in_stride, in + (b0 + b) * in_stride, in_len_each ? in_len_each[b0 + b] : in_len);
            HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + b0 * d_in_stride, piece_in, nb * d_in_stride, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        }
        free(bounce[0]); bounce[0] = NULL;
    }
    if (in_len_each)
        HIP_TRY(lzs_hip_h2d(d_in_len, in_len_each, sizeof(uint32_t) * nblocks, stream), "hipMemcpy H2D");

    STAGE("copy in");
    /* A small batch of blocks to decompress does not fill the device with one wavefront per block
     * (a wavefront takes 9 ms for a 64 KiB block, whatever the batch): then the blocks are cut
     * into segments for many wavefronts, like one long stream. */
    int segmented = 0;
    if (launch == lzs_hip_launch_decompress && cap32 && !lzs_env()->one_wave) {
        size_t total_in = 0;
        for (size_t b = 0; b < nblocks; b++) total_in += in_len_each ? in_len_each[b] : in_len;
        if (nblocks <= BATCH_SEG_MAX_BLOCKS && total_in >= STREAM_DEC_MIN && total_in / nblocks >= 1024u &&
            (unsigned long long)nblocks * d_out_stride <= seg_extent &&
            (unsigned long long)nblocks * d_in_stride < 0xF0000000ull) {      /* segment tables hold 32-bit input offsets */
            rc = batch_decompress_segments(st, stream, who, d_out, d_out_stride, cap32, out_len, (uint32_t *)d_len, d_in, d_in_stride,
                                           in_len_each, (uint32_t)in_len, nblocks);
            if (rc != LZS_OK) goto done;
            segmented = 1;
        }
    }
    if (!segmented) {
        HIP_TRY(launch(d_out, d_out_stride, cap32, (uint32_t *)d_len, d_in, d_in_stride,
                       (const uint32_t *)d_in_len, (uint32_t)in_len, (uint32_t)nblocks, stream), who);
        HIP_TRY(lzs_hip_d2h(out_len, d_len, sizeof(uint32_t) * nblocks, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
    }
    STAGE("kernel + lengths");
    /* copy back only what each block produced: nothing past out_len[b] is touched */
    if (nblocks < 16) {
        for (size_t b = 0; b < nblocks; b++)
            HIP_TRY(lzs_hip_d2h(out + b * out_stride, (uint8_t *)d_out + b * d_out_stride, out_len[b], stream),
                    "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
    } else {
        /* Thousands of small copies cost more than the data: the results are gathered into one
         * dense string on the device, come back in pieces of 32 MiB, and are laid out by the CPU.
         * (Full blocks at the device's own stride need neither: one copy.) */
        size_t total = 0, full = 0;
        for (size_t b = 0; b < nblocks; b++) { total += out_len[b]; full += (b + 1 < nblocks) && out_len[b] == out_stride; }
        if (out_stride == d_out_stride && full == nblocks - 1) {
            HIP_TRY(lzs_hip_d2h(out, d_out, total, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            goto done;
        }
        void *d_dense = NULL, *d_offs = NULL;
        e = staging_reserve(st, BUF_KEEP, total + 64, &d_dense);
        if (!e) e = staging_reserve(st, BUF_AUX, sizeof(uint64_t) * (nblocks + 1), &d_offs);
        if (e) { rc = fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto done; }
        HIP_TRY(lzs_hip_launch_compact(d_dense, (uint64_t *)d_offs, d_out, d_out_stride, (const uint32_t *)d_len,
                                       (uint32_t)nblocks, stream), who);
        const size_t piece = (size_t)32 << 20;
        /* (pinned pieces of the thread, as on the way in) */
        uint8_t *piece_out[2] = {NULL, NULL};
        if (staging_pin_reserve(st, 2, total < piece ? total + 1 : piece, &piece_out[0]) ||
            (total > piece && staging_pin_reserve(st, 3, piece, &piece_out[1]))) {
            bounce[0] = (uint8_t *)malloc(total < piece ? total + 1 : piece);
            bounce[1] = total > piece ? (uint8_t *)malloc(piece) : NULL;
            if (!bounce[0] || (total > piece && !bounce[1])) { rc = fail(LZS_E_NOMEM, "%s: out of host memory", who); goto done; }
            piece_out[0] = bounce[0]; piece_out[1] = bounce[1];
        }
        size_t b = 0, within = 0;                              /* next block to lay out, bytes of it already done */
        for (size_t at = 0, k = 0; at < total || k == 0; k++) {
            const size_t len = total - at < piece ? total - at : piece;
            HIP_TRY(lzs_hip_d2h(piece_out[k & 1], (uint8_t *)d_dense + at, len, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            const uint8_t *src = piece_out[k & 1];
            size_t left = len;
            while (left) {
                while (b < nblocks && within == out_len[b]) { b++; within = 0; }
                const size_t m = out_len[b] - within < left ? out_len[b] - within : left;
                memcpy(out + b * out_stride + within, src, m);
                src += m; left -= m; within += m;
            }
            at += len;
            if (len == 0) break;
        }
    }
    STAGE("copy out + layout");
#undef HIP_TRY
#undef STAGE

done:
    if (rc != LZS_OK && stream) lzs_hip_stream_sync(stream);   /* nothing of ours may still be in flight */
    free(bounce[0]); free(bounce[1]);
    staging_trim(st);
    return rc;
}

int lzs_compress_batch(uint8_t *out, size_t out_stride, size_t out_cap, uint32_t *out_len,
                       const uint8_t *in, size_t in_stride, const uint32_t *in_len_each,
                       size_t in_len, size_t nblocks)
{
    return host_batch("lzs_compress_batch", lzs_hip_launch_compress, out, out_stride, out_cap,
                      out_len, in, in_stride, in_len_each, in_len, nblocks);
}

int lzs_decompress_batch(uint8_t *out, size_t out_stride, size_t out_cap, uint32_t *out_len,
                         const uint8_t *in, size_t in_stride, const uint32_t *in_len_each,
                         size_t in_len, size_t nblocks)
{
    return host_batch("lzs_decompress_batch", lzs_hip_launch_decompress, out, out_stride, out_cap,
                      out_len, in, in_stride, in_len_each, in_len, nblocks);
}

/* ---------------------------------------------- the reference's one-shot entry points */
static size_t one_shot(const char *who, launch_fn launch, uint8_t *out, size_t cap,
                       const uint8_t *in, size_t n)
{
    uint32_t got = 0;
    tls_error[0] = 0;       /* so that a non-empty lzs_last_error() after a 0 return means failure */
    /* The stream can never be longer than the worst case, so a huge caller buffer need not
     * be mirrored on the device in full. */
    size_t useful = cap;
    if (launch == lzs_hip_launch_compress) {
        size_t worst = LZS_COMPRESSED_MAX(n);
        if (useful > worst) useful = worst;
    } else {
        /* every 4-bit extension nibble can yield 15 bytes, so the true bound is 30x
         * (the reference's LZS_DECOMPRESSED_MAX, 16x, under-estimates runs) */
        size_t worst = n > (SIZE_MAX / 32u) ? SIZE_MAX : n * 30u + 64u;
        if (useful > worst) useful = worst;
    }
    int rc = host_batch(who, launch, out, 0, useful, &got, in, 0, NULL, n, 1);
    if (rc != LZS_OK) {
        fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
        return 0;
    }
    /* positions are 32-bit on the device: a stream that fills 4 GiB - 1 of a larger buffer may have
     * more to give, and "full" would be a lie */
    if (launch != lzs_hip_launch_compress && got == 0xFFFFFFFFu && useful > 0xFFFFFFFFull) {
        fail(LZS_E_ARG, "%s: the stream expands to 4 GiB or more; decompress it in pieces (lzs_decompress_incremental)", who);
        fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
        return 0;
    }
    return got;
}

/* Buffers beyond what one launch addresses (positions are 32-bit on the device: LZS_BLOCK_MAX of
 * input, 4 GiB - 1 of output) go through the incremental interface, which carries a stream of any
 * length across pieces of <= 1 GiB with the history, the undecided tail, a running long match and
 * the partial output byte kept in a parameter block (lzs_incremental.c): the bytes
 * are those of the one-shot call, and the reference's size_t lengths (lzs.h:218,229) hold. */
static size_t long_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    LzsCompressParameters_t *p = (LzsCompressParameters_t *)malloc(sizeof(*p));
    if (!p) { fail(LZS_E_NOMEM, "lzs_compress: out of host memory"); return 0; }
    lzs_compress_init_full(p);
    p->inPtr = in; p->inLength = n; p->outPtr = out; p->outLength = cap;
    size_t made = 0;
    for (;;) {
        const size_t in_before = p->inLength;
        const size_t got = lzs_compress_incremental(p, true);
        made += got;
        if (p->status & LZS_C_STATUS_ERROR) { made = 0; break; }         /* (reported; lzs_last_error() has the text) */
        if (p->status & LZS_C_STATUS_END_MARKER) break;
        if (p->outLength == 0) break;                                     /* the buffer is full: the stream is cut there, like the one-shot call's */
        if (got == 0 && p->inLength == in_before) {
            fail(LZS_E_HIP, "lzs_compress: no progress on a long buffer (%zu bytes left)", p->inLength);
            made = 0;
            break;
        }
    }
    free(p);
    return made;
}

static size_t long_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    LzsDecompressParameters_t *p = (LzsDecompressParameters_t *)malloc(sizeof(*p));
    if (!p) { fail(LZS_E_NOMEM, "lzs_decompress: out of host memory"); return 0; }
    lzs_decompress_init(p);
    p->inPtr = in; p->inLength = n; p->outPtr = out; p->outLength = cap;
    size_t made = 0;
    for (;;) {
        const size_t in_before = p->inLength;
        const size_t got = lzs_decompress_incremental(p);
        made += got;
        if (p->status & LZS_D_STATUS_ERROR) { made = 0; break; }
        if (p->status & LZS_D_STATUS_END_MARKER) break;                   /* the one-shot call stops at the first one (:255-260) */
        if (p->outLength == 0 || p->inLength == 0) break;
        if (got == 0 && p->inLength == in_before) break;                  /* a token that the input does not finish */
    }
    free(p);
    return made;
}

/* The small calls' host route (lzs_hostcodec.c): taken BY SIZE on a box that has its device -- require_device() comes
 * first, the library still fails loudly without one -- or by name (LZS_ROUTE=host). */
static int small_call_on_host(const char *who, size_t n, size_t crossover)
{
    if (!route_on_host(n, crossover)) return 0;
    if (lzs_env()->route != LZS_ROUTE_HOST && require_device() != LZS_OK) {
        fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
        return -1;
    }
    return 1;
}

size_t lzs_compress(uint8_t *a_pOutData, size_t a_outBufferSize, const uint8_t *a_pInData, size_t a_inLen)
{
    if ((a_pOutData || !a_outBufferSize) && (a_pInData || !a_inLen)) {
        tls_error[0] = 0;
        const int h = small_call_on_host("lzs_compress", a_inLen, HOST_COMPRESS_MAX);
        if (h < 0) return 0;
        if (h && a_inLen <= 0x40000000u) {
            const size_t got = hostcodec_compress(a_pOutData, a_outBufferSize, a_pInData, a_inLen);
            if (got != SIZE_MAX) return got;
            fail(LZS_E_NOMEM, "lzs_compress: out of host memory");
            fprintf(stderr, "liblzs: lzs_compress failed: %s\n", tls_error);
            return 0;
        }
    }
    if (a_inLen > LZS_BLOCK_MAX && a_pOutData && a_pInData) {
        tls_error[0] = 0;
        return long_compress(a_pOutData, a_outBufferSize, a_pInData, a_inLen);
    }
    if ((a_inLen > STREAM_MIN || (a_inLen && lzs_env()->force_stream)) && a_inLen <= LZS_BLOCK_MAX && a_pOutData && a_pInData && !lzs_env()->one_workgroup)
        return stream_compress(a_pOutData, a_outBufferSize, a_pInData, a_inLen, 0, NULL);
    return one_shot("lzs_compress", lzs_hip_launch_compress, a_pOutData, a_outBufferSize, a_pInData, a_inLen);
}

size_t lzs_decompress(uint8_t *a_pOutData, size_t a_outBufferSize, const uint8_t *a_pInData, size_t a_inLen)
{
    if ((a_pOutData || !a_outBufferSize) && (a_pInData || !a_inLen)) {
        tls_error[0] = 0;
        const int h = small_call_on_host("lzs_decompress", a_inLen, HOST_DECOMPRESS_MAX);
        if (h < 0) return 0;
        if (h) return hostcodec_decompress(a_pOutData, a_outBufferSize, a_pInData, a_inLen);
    }
    /* a stream longer than one launch takes, or one that may fill more than the 32-bit positions of
     * the device reach (a length nibble stands for up to 15 bytes: 30 bytes of output per byte) */
    if (a_pOutData && a_pInData &&
        (a_inLen > LZS_BLOCK_MAX || (a_outBufferSize > 0xE0000000ull && a_inLen > (0xE0000000ull - 64u) / 30u))) {
        tls_error[0] = 0;
        return long_decompress(a_pOutData, a_outBufferSize, a_pInData, a_inLen);
    }
    if ((a_inLen > STREAM_DEC_MIN || (a_inLen && lzs_env()->force_stream)) && a_inLen <= LZS_BLOCK_MAX &&
        a_pOutData && a_pInData && a_outBufferSize && !lzs_env()->one_wave) {
        const size_t got = stream_decompress(a_pOutData, a_outBufferSize, a_pInData, a_inLen, 0, NULL, 0, NULL);
        if (got != SIZE_MAX) return got;
    }
    return one_shot("lzs_decompress", lzs_hip_launch_decompress, a_pOutData, a_outBufferSize, a_pInData, a_inLen);
}

size_t lzs_decompress_concat(uint8_t *out, size_t out_cap, const uint8_t *in, size_t in_len)
{
    if ((in_len > STREAM_DEC_MIN || (in_len && lzs_env()->force_stream)) && in_len <= LZS_BLOCK_MAX &&
        out && in && out_cap && !lzs_env()->one_wave) {
        const size_t got = stream_decompress(out, out_cap, in, in_len, 0, NULL, 1, NULL);
        if (got != SIZE_MAX) return got;
    }
    return one_shot("lzs_decompress_concat", lzs_hip_launch_decompress_concat, out, out_cap, in, in_len);
}

