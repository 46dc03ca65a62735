/*
 * lzs_host.c -- the C host side of liblzs (MI355X build): argument checking, error
 * reporting, host<->device staging.  All codec work happens in lzs_kernels.hip, reached
 * through the extern-"C" shim in lzs_hip_shim.h.  There is deliberately NO CPU codec
 * here: without a HIP device every entry point fails loudly.
 *
 * Public surface: include/lzs/lzs.h (the reference's one-shot calls,
 * c/src/liblzs/lzs.h:218,229) and include/lzs/lzs_batch.h (additive batch calls).
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lzs/lzs.h"
#include "lzs/lzs_batch.h"
#include "lzs_hip_shim.h"

static _Thread_local char tls_error[512];

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_error, sizeof(tls_error), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(int hip_error, const char *what)
{
    return fail(LZS_E_HIP, "%s: %s", what, lzs_hip_strerror(hip_error));
}

const char *lzs_last_error(void) { return tls_error; }

static int require_device(void)
{
    int n = 0;
    int e = lzs_hip_device_count(&n);
    if (e != 0 || n <= 0)
        return fail(LZS_E_NO_DEVICE, "no HIP device available (%s); liblzs has no CPU codec",
                    e ? lzs_hip_strerror(e) : "device count is 0");
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
typedef int (*launch_fn)(void *, size_t, uint32_t, uint32_t *, const void *, size_t,
                         const uint32_t *, uint32_t, uint32_t, void *);

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
/* Each host thread keeps one HIP stream and four grow-only device buffers for the host-buffer
 * entry points, so a small one-shot call costs copies + a launch, not allocations.  They are
 * released when the thread exits; buffers above KEEP_MAX are released right after the call. */
enum { BUF_IN, BUF_OUT, BUF_LEN, BUF_INLEN, BUF_AUX, BUF_KEEP, BUF_MARKS, BUF_COUNT };
#define KEEP_MAX ((size_t)256 << 20)
#define STREAM_DEC_MIN 4096u                                /* shorter streams are decompressed by one wavefront */
/* Batches up to this much output are decompressed in segments; larger ones fill the device with a
 * wavefront per block.  Measured (text, 64 KiB blocks, host buffers, ms; segments / wavefront per
 * block): 4 blocks 0.74 / 8.3, 64 blocks 1.5 / 8.5, 256 blocks 4.2 / 9.4, 512 blocks 11.4 / 13.4,
 * 1024 blocks 25.5 / 24.8. */
#define BATCH_SEG_MAX_BLOCKS 4096u
#define BATCH_SEG_MAX_EXTENT ((unsigned long long)32 << 20)

typedef struct {
    void  *stream;
    void  *buf[BUF_COUNT];
    size_t cap[BUF_COUNT];
    uint8_t *host_box;              /* host side of the small incremental calls' single copies */
} staging_t;

static pthread_key_t  staging_key;
static pthread_once_t staging_once = PTHREAD_ONCE_INIT;

static void staging_destroy(void *p)
{
    staging_t *st = (staging_t *)p;
    if (!st) return;
    for (int i = 0; i < BUF_COUNT; i++)
        if (st->buf[i]) lzs_hip_free(st->buf[i]);
    if (st->stream) lzs_hip_stream_destroy(st->stream);
    free(st->host_box);
    free(st);
}

static void staging_make_key(void) { pthread_key_create(&staging_key, staging_destroy); }

static staging_t *staging_get(void)
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
static int staging_reserve(staging_t *st, int which, size_t bytes, void **out)
{
    if (st->cap[which] < bytes) {
        if (st->buf[which]) { lzs_hip_free(st->buf[which]); st->buf[which] = NULL; st->cap[which] = 0; }
        size_t want = (bytes + 65535u) & ~(size_t)65535u;
        int e = lzs_hip_malloc(&st->buf[which], want);
        if (e) return e;
        st->cap[which] = want;
    }
    *out = st->buf[which];
    return 0;
}

/* Buffers above the limit are released right after the call (LZS_KEEP_MAX_MB overrides the
 * default of 256 MiB per buffer for programs that compress large buffers over and over). */
static size_t keep_max(void)
{
    static size_t limit = 0;
    if (!limit) {
        const char *v = getenv("LZS_KEEP_MAX_MB");
        const unsigned long mb = v ? strtoul(v, NULL, 10) : 0;
        limit = mb ? (size_t)mb << 20 : KEEP_MAX;
    }
    return limit;
}

static void staging_trim(staging_t *st)
{
    for (int i = 0; i < BUF_COUNT; i++)
        if (st->cap[i] > keep_max()) { lzs_hip_free(st->buf[i]); st->buf[i] = NULL; st->cap[i] = 0; }
}

/* -------------------------------------------------------------------- host batches */
static uint32_t stream_dec_seg(size_t n);
static double now_ms(void);
static int batch_decompress_segments(staging_t *st, void *stream, const char *who, void *d_out, size_t d_out_stride,
                                     uint32_t cap32, uint32_t *out_len, uint32_t *d_len, const void *d_in, size_t d_in_stride,
                                     const uint32_t *in_len_each, uint32_t in_len, size_t nblocks);
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
        bounce[0] = (uint8_t *)malloc(per * d_in_stride);
        if (!bounce[0]) { rc = fail(LZS_E_NOMEM, "%s: out of host memory", who); goto done; }
        for (size_t b0 = 0; b0 < nblocks; b0 += per) {
            const size_t nb = nblocks - b0 < per ? nblocks - b0 : per;
            for (size_t b = 0; b < nb; b++)
                memcpy(bounce[0] + b * d_in_stride, in + (b0 + b) * in_stride, in_len_each ? in_len_each[b0 + b] : in_len);
            HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + b0 * d_in_stride, bounce[0], nb * d_in_stride, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        }
        free(bounce[0]); bounce[0] = NULL;
    }
    if (in_len_each)
        HIP_TRY(lzs_hip_h2d(d_in_len, in_len_each, sizeof(uint32_t) * nblocks, stream), "hipMemcpy H2D");

    /* A small batch of blocks to decompress does not fill the device with one wavefront per block
     * (a wavefront takes 9 ms for a 64 KiB block, whatever the batch): then the blocks are cut
     * into segments for many wavefronts, like one long stream. */
    int segmented = 0;
    if (launch == lzs_hip_launch_decompress && cap32 && !getenv("LZS_ONE_WAVE")) {
        size_t total_in = 0;
        for (size_t b = 0; b < nblocks; b++) total_in += in_len_each ? in_len_each[b] : in_len;
        if (nblocks <= BATCH_SEG_MAX_BLOCKS && total_in >= STREAM_DEC_MIN && total_in / nblocks >= 1024u &&
            (unsigned long long)nblocks * d_out_stride <= BATCH_SEG_MAX_EXTENT) {
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
        bounce[0] = (uint8_t *)malloc(total < piece ? total + 1 : piece);
        bounce[1] = total > piece ? (uint8_t *)malloc(piece) : NULL;
        if (!bounce[0] || (total > piece && !bounce[1])) { rc = fail(LZS_E_NOMEM, "%s: out of host memory", who); goto done; }
        size_t b = 0, within = 0;                              /* next block to lay out, bytes of it already done */
        for (size_t at = 0, k = 0; at < total || k == 0; k++) {
            const size_t len = total - at < piece ? total - at : piece;
            HIP_TRY(lzs_hip_d2h(bounce[k & 1], (uint8_t *)d_dense + at, len, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            const uint8_t *src = bounce[k & 1];
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
#undef HIP_TRY

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
    return got;
}

/* ------------------------------------------------ one long stream on the whole device */
/* lzs_compress() of a buffer too long for one workgroup to be worth waiting for.  The search is a
 * pure function of (input, position), so the stream is cut into 64 KiB segments, one workgroup
 * each (lzs_compress_segments_kernel), every one writing its bits into a slot of its own.  What a
 * segment cannot know by itself is where its first token starts -- the last token of the segment
 * before usually reaches a few bytes into it -- and at which bit its output begins.  So: (1)
 * every segment is compressed entered at its own start and reports where its last token ends;
 * (2) segments whose predecessor ended elsewhere are compressed again from there, until all
 * entries agree (the greedy parses from two nearby entries merge after a few tokens, so a second
 * round changes almost no exit; a long run simply skips the segments it covers); (3) prefix sum
 * of the bit counts on the host; (4) lzs_stitch_segments_kernel shifts every slot to its bit
 * offset in the zeroed output and appends the end marker.  Same bytes as one workgroup (or the
 * reference) produces. */
static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define STREAM_SEG_MAX 65536u
#define STREAM_SEG_MIN 4096u
#define STREAM_MIN   24576u                     /* shorter inputs stay with one workgroup */

/* Segment size for a stream of n bytes: 64 KiB for long streams, smaller for shorter ones so that
 * they too spread over the device -- a workgroup takes ~1.2 ms per 64 KiB, and every segment pays
 * for a 2.2 KB warm-up of its chains.  Measured (text, host buffers, ms): 64 KiB 1.12 with one
 * workgroup, 0.29 in 4 KiB segments; 1 MiB 1.38 in 64 KiB segments, 0.45 in 4 KiB ones; 4 MiB
 * 2.84 / 1.05 (8 KiB); 16 MiB 5.17 / 4.13 (16 KiB). */
static uint32_t stream_seg(size_t n)
{
    const char *v = getenv("LZS_STREAM_SEG");
    size_t seg = v ? strtoul(v, NULL, 10) : (n / 512u + 4095u) & ~(size_t)4095u;
    if (seg < STREAM_SEG_MIN) seg = STREAM_SEG_MIN;
    if (seg > STREAM_SEG_MAX) seg = STREAM_SEG_MAX;
    return (uint32_t)(seg & ~(size_t)63u);
}

/* in/out on the host (dev == 0: staged through this thread's device buffers) or on the device */
/* A piece of a stream for lzs_compress_incremental(): the data is `prefix` (history and the
 * bytes the call before could not encode yet) followed by `in`; encoding starts at c0, inside a
 * long match if ext_off is set, at bit bit0 of the first output byte (whose earlier bits are in
 * `first`).  Unless `last`, it stops where the tokens are no longer decided by the data so far. */
typedef struct {
    const uint8_t *prefix;
    uint32_t prefix_len;
    uint32_t c0, ext_off, bit0;
    uint8_t  first;
    int      last;
    /* results */
    uint32_t c_exit;        /* everything before it is encoded */
    uint32_t ext_exit;      /* != 0: still inside a long match at this offset */
    uint64_t nbits;         /* bits in the output (bit0 included, end marker not) */
} piece_t;

static size_t stream_compress_piece(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status,
                                    piece_t *pc)
{
    const char *who = pc ? "lzs_compress_incremental" : dev ? "lzs_compress_stream_device" : "lzs_compress";
    const uint32_t STREAM_SEG = stream_seg(n);
    const uint32_t nseg = n ? (uint32_t)((n + STREAM_SEG - 1) / STREAM_SEG) : 1u;
    const size_t worst = LZS_COMPRESSED_MAX(n - (pc ? pc->c0 : 0)) + 8;
    const int end_marker = !pc || pc->last;
    const uint32_t lim = end_marker ? (uint32_t)n : (n > LZS_MAX_LOOK_AHEAD_LEN ? (uint32_t)n - LZS_MAX_LOOK_AHEAD_LEN : 0u);
    size_t result = 0;
    int e = 0, rc = LZS_OK;
    void *d_in = NULL, *d_out = NULL, *d_aux = NULL, *d_slots = NULL;
    const size_t slot_stride = (LZS_COMPRESSED_MAX((size_t)STREAM_SEG) + 15u) & ~(size_t)15u;
    uint32_t *entry = NULL, *exitp = NULL, *openi = NULL;
    uint64_t *nbits = NULL, *bitat = NULL;
    uint8_t *dirty = NULL;
    tls_error[0] = 0;
    if (require_device() != LZS_OK) goto failed;
    staging_t *st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
    entry = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    exitp = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    nbits = (uint64_t *)malloc(sizeof(uint64_t) * nseg);
    bitat = (uint64_t *)malloc(sizeof(uint64_t) * nseg);
    dirty = (uint8_t *)malloc(nseg);
    openi = (uint32_t *)malloc(sizeof(uint32_t) * 2 * nseg);
    if (!entry || !exitp || !nbits || !bitat || !dirty || !openi) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }

#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    /* device arrays in one allocation: bit_at, nbits (8 B each), entry, exit, open info (4 + 4 + 8 B), dirty */
    const size_t aux_bytes = (size_t)nseg * (8 + 8 + 4 + 4 + 8 + 1) + 64;
    if (dev) { d_in = (void *)in; d_out = out; }
    else {
        e = staging_reserve(st, BUF_IN, n + 64, &d_in);
        if (!e) e = staging_reserve(st, BUF_OUT, worst + 1024, &d_out);
    }
    if (!e) e = staging_reserve(st, BUF_AUX, aux_bytes, &d_aux);
    if (!e) e = staging_reserve(st, BUF_KEEP, slot_stride * nseg + 64, &d_slots);
    if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
    uint64_t *d_bitat = (uint64_t *)d_aux;
    uint64_t *d_nbits = d_bitat + nseg;
    uint32_t *d_entry = (uint32_t *)(d_nbits + nseg);
    uint32_t *d_exit = d_entry + nseg;
    uint32_t *d_open = d_exit + nseg;
    uint8_t *d_dirty = (uint8_t *)(d_open + 2 * (size_t)nseg);

    const int debug = getenv("LZS_STREAM_DEBUG") != NULL;      /* stage times on stderr */
    double t0 = debug ? now_ms() : 0, t1;
    if (!dev) {
        const size_t pre = pc ? pc->prefix_len : 0;
        if (pre) HIP_TRY(lzs_hip_h2d(d_in, pc->prefix, pre, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + pre, in, n - pre, stream), "hipMemcpy H2D");
    }
    HIP_TRY(lzs_hip_memset(d_out, 0, worst + 1024, stream), "hipMemset");
    if (debug) { lzs_hip_stream_sync(stream); t1 = now_ms(); fprintf(stderr, "liblzs stream: %zu B, %u segments; H2D + memset %.2f ms\n", n, nseg, t1 - t0); t0 = t1; }
    uint32_t c_first = 0, ext_now = 0;
    uint64_t total = 0;
    if (pc) {
        c_first = pc->c0;
        total = pc->bit0;
        if (pc->bit0) HIP_TRY(lzs_hip_h2d(d_out, &pc->first, 1, stream), "hipMemcpy H2D");
        if (pc->ext_off) {
            /* the piece begins inside a long match: its length nibbles first (d_exit as scratch) */
            uint32_t res[4];
            HIP_TRY(lzs_hip_launch_extend_resume(d_out, pc->bit0, d_in, (uint32_t)n, pc->c0, pc->ext_off, pc->last, d_exit, stream), who);
            HIP_TRY(lzs_hip_d2h(res, d_exit, sizeof(res), stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            c_first = res[0];
            ext_now = res[1] ? pc->ext_off : 0;
            total += ((uint64_t)res[3] << 32) | res[2];
        }
    }
    for (uint32_t k = 0; k < nseg; k++) {
        entry[k] = k * STREAM_SEG > c_first ? k * STREAM_SEG : c_first;
        dirty[k] = 1; exitp[k] = entry[k]; nbits[k] = 0; openi[2 * k] = openi[2 * k + 1] = 0;
    }
    /* (a match still open after the nibbles covers all the data there is: no tokens in this piece) */
    for (uint32_t round = 0, ndirty = ext_now ? 0 : nseg; ndirty; round++) {
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_compress_segments(d_slots, slot_stride, d_in, (uint32_t)n, STREAM_SEG, nseg,
                                                 d_entry, d_dirty, d_exit, d_nbits, NULL, NULL, lim, pc ? d_open : NULL, stream), who);
        HIP_TRY(lzs_hip_d2h(exitp, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        /* a segment is entered where the one before stopped (its own start for segment 0) */
        const uint32_t was = ndirty;
        ndirty = 0;
        dirty[0] = 0;
        for (uint32_t k = 1; k < nseg; k++) {
            dirty[k] = exitp[k - 1] != entry[k];
            if (dirty[k]) { entry[k] = exitp[k - 1]; ndirty++; }
        }
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream: round %u compressed %u segments in %.2f ms; %u to redo\n", round, was, t1 - t0, ndirty); t0 = t1; }
    }
    if (!ext_now) {
        HIP_TRY(lzs_hip_d2h(nbits, d_nbits, sizeof(uint64_t) * nseg, stream), "hipMemcpy D2H");
        if (pc) HIP_TRY(lzs_hip_d2h(openi, d_open, sizeof(uint32_t) * 2 * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
    }
    if (pc) {
        pc->c_exit = ext_now ? c_first : exitp[nseg - 1];
        pc->ext_exit = ext_now;
        if (!pc->last && !ext_now && pc->c_exit >= n && n > c_first) {
            /* The last token is a match that reaches the end of the data so far: it may go on in
             * the next piece.  Its full groups of 15 stand; the closing nibble is taken back and
             * the bytes it covered wait for more data (state COMPRESS_EXTENDED, :750-758). */
            uint32_t k = nseg;
            while (k > 0 && openi[2 * (k - 1)] == 0) k--;
            if (k == 0 || nbits[k - 1] < 4) {
                rc = fail(LZS_E_HIP, "%s: inconsistent state from the device (piece of %zu bytes from %u ends at %u, no open match reported)",
                          who, n, c_first, pc->c_exit);
                goto failed;
            }
            const uint32_t off = openi[2 * (k - 1)], start = openi[2 * (k - 1) + 1];
            const uint32_t rest = ((uint32_t)n - start - 8u) % 15u;
            nbits[k - 1] -= 4;
            HIP_TRY(lzs_hip_h2d(d_nbits, nbits, sizeof(uint64_t) * nseg, stream), "hipMemcpy H2D");
            pc->c_exit = (uint32_t)n - rest;
            pc->ext_exit = off;
        }
    }
    for (uint32_t k = 0; k < nseg; k++) { bitat[k] = total; total += nbits[k]; }
    if (pc) pc->nbits = total;
    HIP_TRY(lzs_hip_h2d(d_bitat, bitat, sizeof(uint64_t) * nseg, stream), "hipMemcpy H2D");
    if (!ext_now)
        HIP_TRY(lzs_hip_launch_stitch_segments(d_out, d_slots, slot_stride, d_bitat, d_nbits, nseg, end_marker, stream), who);
    /* segments whose bits did not fit their slot (a match running on for more than ~120 KB past
     * the segment): once more, ORed straight into place */
    uint32_t nbig = 0;
    for (uint32_t k = 0; k < nseg; k++) { dirty[k] = nbits[k] > 8u * (uint64_t)slot_stride; nbig += dirty[k]; }
    if (nbig) {
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_compress_segments(d_slots, slot_stride, d_in, (uint32_t)n, STREAM_SEG, nseg,
                                                 d_entry, d_dirty, d_exit, d_nbits, d_out, d_bitat, lim, NULL, stream), who);
    }
    if (debug) { lzs_hip_stream_sync(stream); t1 = now_ms(); fprintf(stderr, "liblzs stream: stitch %.2f ms\n", t1 - t0); t0 = t1; }
    result = end_marker ? (size_t)((total + 9 + 7) / 8)        /* end marker, padded to a byte */
                        : (size_t)((total + 7) / 8);            /* a piece: the last byte may be partial */
    if (result > cap) result = cap;                            /* cut at the capacity, prefix unchanged */
    if (!dev) HIP_TRY(lzs_hip_d2h(out, d_out, result, stream), "hipMemcpy D2H");
    HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
#undef HIP_TRY
    goto done;

failed:
    result = 0;
    if (rc == LZS_OK) rc = LZS_E_HIP;
    if (!dev) fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
done:
    free(entry); free(exitp); free(nbits); free(bitat); free(dirty); free(openi);
    { staging_t *s2 = staging_get(); if (s2) staging_trim(s2); }
    if (status) *status = rc;
    return result;
}

static size_t stream_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status)
{
    return stream_compress_piece(out, cap, in, n, dev, status, NULL);
}

int lzs_compress_stream_device(void *d_out, size_t out_cap, size_t *out_len, const void *d_in, size_t in_len)
{
    if (!out_len) return fail(LZS_E_ARG, "lzs_compress_stream_device: out_len is NULL");
    *out_len = 0;
    if (!d_out || (!d_in && in_len)) return fail(LZS_E_ARG, "lzs_compress_stream_device: NULL buffer");
    if (((uintptr_t)d_out & 3u) != 0) return fail(LZS_E_ARG, "lzs_compress_stream_device: d_out is not 4-byte aligned");
    if (in_len == 0 || in_len > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "lzs_compress_stream_device: length must be 1..LZS_BLOCK_MAX");
    int rc = LZS_OK;
    *out_len = stream_compress((uint8_t *)d_out, out_cap, (const uint8_t *)d_in, in_len, 1, &rc);
    return rc;
}

size_t lzs_compress(uint8_t *a_pOutData, size_t a_outBufferSize, const uint8_t *a_pInData, size_t a_inLen)
{
    if ((a_inLen > STREAM_MIN || (a_inLen && getenv("LZS_FORCE_STREAM"))) && a_inLen <= LZS_BLOCK_MAX && a_pOutData && a_pInData && !getenv("LZS_ONE_WORKGROUP"))
        return stream_compress(a_pOutData, a_outBufferSize, a_pInData, a_inLen, 0, NULL);
    return one_shot("lzs_compress", lzs_hip_launch_compress, a_pOutData, a_outBufferSize, a_pInData, a_inLen);
}

/* Segment size for decompressing a stream of n compressed bytes: 8 KiB for long streams, smaller
 * for short ones so that they too are spread over many wavefronts (one wavefront decodes ~7 MB/s).
 * Measured (text, host buffers, ms; output size): 64 KiB 7.7 with one wavefront, 0.67 in 256-byte
 * segments; 256 KiB 30.8 / 1.0; 1 MiB 122.8 / 2.4 (6.3 in 8 KiB segments); 4 MiB 8.2 in 1 KiB
 * segments, 10.1 in 8 KiB ones. */
static uint32_t stream_dec_seg(size_t n)
{
    const char *v = getenv("LZS_DEC_SEG");
    size_t seg = v ? strtoul(v, NULL, 10) : (n / 2048u + 255u) & ~(size_t)255u;
    const size_t most = lzs_hip_dec_segment_bytes();
    if (seg < 256u) seg = 256u;
    if (seg > most) seg = most;
    return (uint32_t)(seg & ~(size_t)63u);
}

/* lzs_decompress() of one long stream by many wavefronts: see lzs_scan_stream_kernel.  Returns
 * SIZE_MAX if this path does not apply (output of 4 GiB or more) and the caller should decode
 * with one wavefront. */
/* A piece of a stream for lzs_decompress_incremental(): the input is `prefix` (the bytes that hold
 * the bits left over from the call before) followed by `in`; the walk starts in state `entry0`
 * (bit offset into the first byte, extension running, offset: the kernels' state word); copies may
 * reach back into `hist`, the last bytes produced before.  Only whole segments are decoded, and
 * only those before the first one in which the stream stops (end marker, unfinished token) or
 * which would overflow the output: the rest is the one wavefront's (lzs_decode_resume_kernel). */
typedef struct {
    const uint8_t *prefix;
    uint32_t prefix_len;
    uint32_t entry0;
    const uint8_t *hist;
    uint32_t hist_len;
    /* results */
    uint32_t seg, segs_done;    /* segment size used; segments decoded */
    uint32_t next_entry;        /* state word at the start of segment segs_done */
} dec_piece_t;

static size_t stream_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status, int concat,
                                dec_piece_t *dp)
{
    const char *who = dp ? "lzs_decompress_incremental" : dev ? "lzs_decompress_stream_device" : concat ? "lzs_decompress_concat" : "lzs_decompress";
    const uint32_t seg = stream_dec_seg(n);
    const uint32_t nseg = (uint32_t)((n + seg - 1) / seg);
    size_t result = 0;
    int e = 0, rc = LZS_OK;
    void *d_in = NULL, *d_out = NULL, *d_aux = NULL, *d_origin = NULL, *d_marks = NULL;
    uint32_t *entry = NULL, *exits = NULL, *count = NULL, *start = NULL;
    uint8_t *dirty = NULL, *ones = NULL;
    uint32_t *seen = NULL;
    tls_error[0] = 0;
    if (require_device() != LZS_OK) goto failed;
    staging_t *st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
    ones = (uint8_t *)malloc(nseg);
    seen = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    entry = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    exits = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    count = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    start = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
    dirty = (uint8_t *)malloc(nseg);
    if (!entry || !exits || !count || !start || !dirty || !ones || !seen) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }

#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    const size_t aux_bytes = (size_t)nseg * (4 + 4 + 4 + 4 + 1 + 1) + 128;
    if (dev) d_in = (void *)in; else e = staging_reserve(st, BUF_IN, n + 64, &d_in);
    if (!e) e = staging_reserve(st, BUF_AUX, aux_bytes, &d_aux);
    if (!e) e = staging_reserve(st, BUF_MARKS, (size_t)nseg * LZS_SCAN_MARK_WORDS * 4u, &d_marks);
    if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
    uint32_t *d_entry = (uint32_t *)d_aux;
    uint32_t *d_exit = d_entry + nseg;
    uint32_t *d_count = d_exit + nseg;
    uint32_t *d_start = d_count + nseg;
    uint32_t *d_counters = d_start + nseg;                     /* [0] bytes with an origin, [1] left open */
    uint8_t *d_dirty = (uint8_t *)(d_counters + 2);
    uint8_t *d_ones = d_dirty + nseg;

    const int debug = getenv("LZS_STREAM_DEBUG") != NULL;
    double t0 = debug ? now_ms() : 0, t1;
    if (!dev) {
        const size_t pre = dp ? dp->prefix_len : 0;
        if (pre) HIP_TRY(lzs_hip_h2d(d_in, dp->prefix, pre, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + pre, in, n - pre, stream), "hipMemcpy H2D");
    }
    /* SCAN rounds: every segment entered at its first bit in the normal state, then corrected */
    for (uint32_t k = 0; k < nseg; k++) { entry[k] = 0; dirty[k] = 1; seen[k] = 0xFFFFFFFFu; }
    if (dp) entry[0] = dp->entry0;
    for (uint32_t round = 0, ndirty = nseg; ndirty; round++) {
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_scan_stream(d_in, (uint32_t)n, nseg, d_entry, d_dirty, d_exit, d_count,
                                           round == 0 ? d_ones : NULL, (uint32_t *)d_marks, round != 0 && !getenv("LZS_NO_MARKS"), seg, concat, NULL, NULL, stream), who);
        HIP_TRY(lzs_hip_d2h(exits, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        if (round == 0) HIP_TRY(lzs_hip_d2h(ones, d_ones, nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_d2h(count, d_count, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        const uint32_t was = ndirty;
        for (uint32_t k = 0; k < nseg; k++) if (dirty[k]) seen[k] = entry[k];   /* exits[k], count[k] belong to this entry */
        ndirty = 0;
        dirty[0] = 0;
        int ended = 0;
        int settled = 1;            /* every segment before k has been walked from its final entry */
        for (uint32_t k = 1; k < nseg; k++) {
            uint32_t want = exits[k - 1];
            if (want & LZS_SEG_STOP) {
                /* end marker or end of input before k -- believed only from a settled walk: one that
                 * was entered at a guessed bit reads end markers into the data now and then */
                if (settled) ended = 1; else want = entry[k];
            }
            if (ended) want = LZS_SEG_STOP;
            if (want != entry[k]) settled = 0;
            entry[k] = want;
            dirty[k] = 0;
            /* a segment behind the (current) end of the stream keeps what it reported for its last
             * entry: the end may turn out to be a misread of a walk that had not fallen in step */
            if (ended || want == seen[k]) continue;
            if (((want >> 8) & 1u) && (ones[k] == 2 || (ones[k] && (want & 3u) == 0)) && !getenv("LZS_NO_ONES")) {
                /* all 0xFF inside a running extension: nothing but nibbles of 15, one every 4 bits
                 * from the cursor on (which is up to 20 bits in if the match token itself straddles
                 * the border) for as long as they start inside the segment -- provided the last of
                 * them is all ones too, which reaches up to 3 bits into the next segment unless the
                 * cursor is a multiple of 4.  No need to walk it then. */
                const uint32_t r = want & 0xFFu;
                const uint32_t nibbles = (seg * 8u - r + 3u) / 4u;
                exits[k] = (want & ~0xFFu) | (r + 4u * nibbles - seg * 8u);
                count[k] = 15u * nibbles;
                seen[k] = want;
                continue;
            }
            dirty[k] = 1;
            ndirty++;
        }
        /* what the host worked out itself must survive the next round's copy back */
        if (ndirty) {
            HIP_TRY(lzs_hip_h2d(d_exit, exits, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_h2d(d_count, count, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        }
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream decode: round %u scanned %u of %u segments in %.2f ms; %u to redo\n", round, was, nseg, t1 - t0, ndirty); t0 = t1; }
    }
    if (getenv("LZS_VERIFY_SCAN")) {
        /* development check: every segment walked in full from its final entry must report what
         * the rounds arrived at (merged walks and the all-0xFF shortcut included) */
        uint32_t *ex2 = (uint32_t *)malloc(sizeof(uint32_t) * nseg), *cn2 = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
        memset(dirty, 1, nseg);
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_scan_stream(d_in, (uint32_t)n, nseg, d_entry, d_dirty, d_exit, d_count, NULL, NULL, 0, seg, concat, NULL, NULL, stream), who);
        HIP_TRY(lzs_hip_d2h(ex2, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_d2h(cn2, d_count, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        for (uint32_t k = 0; k < nseg; k++) {
            if (entry[k] & LZS_SEG_STOP) break;
            if (ex2[k] != exits[k] || cn2[k] != count[k])
                fprintf(stderr, "liblzs verify: segment %u entry %08x: rounds say exit %08x count %u, a full walk says %08x %u (all-ones %u)\n",
                        k, entry[k], exits[k], count[k], ex2[k], cn2[k], ones[k]);
        }
        free(ex2); free(cn2);
    }
    uint64_t total = 0;
    uint32_t ndec = nseg;                                      /* segments to decode */
    const uint32_t before = dp ? dp->hist_len : 0;             /* bytes in front of out[0] that copies may reach */
    for (uint32_t k = 0; k < nseg; k++) {
        if (dp && ((exits[k] & LZS_SEG_STOP) || (entry[k] & LZS_SEG_STOP) || total + count[k] > cap)) { ndec = k; break; }
        start[k] = (uint32_t)total + before;
        if (!(entry[k] & LZS_SEG_STOP)) total += count[k];
        if (total >= 0xFFFFFF00ull - 0x100000ull) break;
    }
    if (total >= 0xFFFFFF00ull - 0x100000ull) { result = SIZE_MAX; goto done; }   /* positions are 32-bit here */
    if (dp) { dp->seg = seg; dp->segs_done = ndec; dp->next_entry = ndec < nseg ? entry[ndec] : exits[nseg - 1]; }
    const uint32_t produce = (uint32_t)(total < cap ? total : cap);
    if (produce) {
        if (dev) d_out = out; else e = staging_reserve(st, BUF_OUT, (size_t)before + produce + 64, &d_out);
        if (!e) e = staging_reserve(st, BUF_KEEP, 4 * ((size_t)before + produce) + 64, &d_origin);
        if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
        if (before) {                                          /* the history: final bytes (origin "clean" = all ones) */
            HIP_TRY(lzs_hip_h2d(d_out, dp->hist, before, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_memset(d_origin, 0xFF, 4 * (size_t)before, stream), "hipMemset");
        }
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_start, start, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_memset(d_counters, 0, 8, stream), "hipMemset");
        HIP_TRY(lzs_hip_launch_decode_stream(d_out, before + produce, (uint32_t *)d_origin, d_counters, d_in, (uint32_t)n,
                                             ndec, d_entry, d_start, seg, concat, NULL, NULL, NULL, NULL, stream), who);
        uint32_t open[2] = {0, 0};
        HIP_TRY(lzs_hip_d2h(open, d_counters, 8, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream decode: %u bytes decoded in %.2f ms, %u with an origin in another segment\n", produce, t1 - t0, open[0]); t0 = t1; }
        uint32_t left = open[0];
        for (uint32_t round = 1; left && round < 250; round++) {
            HIP_TRY(lzs_hip_memset(d_counters + 1, 0, 4, stream), "hipMemset");
            HIP_TRY(lzs_hip_launch_resolve_stream(d_out, (uint32_t *)d_origin, before + produce, round, d_counters + 1, stream), who);
            HIP_TRY(lzs_hip_d2h(&left, d_counters + 1, 4, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream decode: resolve round %u in %.2f ms, %u left\n", round, t1 - t0, left); t0 = t1; }
        }
        if (left) { fail(LZS_E_HIP, "%s: origins did not resolve", who); goto failed; }
        if (!dev) HIP_TRY(lzs_hip_d2h(out, (uint8_t *)d_out + before, produce, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
    }
    result = produce;
#undef HIP_TRY
    goto done;

failed:
    result = 0;
    if (rc == LZS_OK) rc = LZS_E_HIP;
    if (!dev) fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
done:
    free(entry); free(exits); free(count); free(start); free(dirty); free(ones); free(seen);
    { staging_t *s2 = staging_get(); if (s2) staging_trim(s2); }
    if (status) *status = rc;
    return result;
}

/* A batch of blocks decompressed like one long stream: every block is cut into segments of its
 * own (tables tell the kernels where a segment starts, where its stream ends, where its block's
 * output begins and must end), the scan rounds run over all of them at once -- a block's first
 * segment is always entered at bit 0 in the normal state -- and the origins are resolved over the
 * whole strided output.  d_in / d_out are the staged device buffers of host_batch(). */
static int batch_decompress_segments(staging_t *st, void *stream, const char *who, void *d_out, size_t d_out_stride,
                                     uint32_t cap32, uint32_t *out_len, uint32_t *d_len, const void *d_in, size_t d_in_stride,
                                     const uint32_t *in_len_each, uint32_t in_len, size_t nblocks)
{
    int e = 0, rc = LZS_OK;
    size_t total_in = 0;
    for (size_t b = 0; b < nblocks; b++) total_in += in_len_each ? in_len_each[b] : in_len;
    const uint32_t seg = stream_dec_seg(total_in / 4);        /* (smaller than for one stream of that size: measured) */
    uint32_t nseg = 0;
    for (size_t b = 0; b < nblocks; b++) nseg += ((in_len_each ? in_len_each[b] : in_len) + seg - 1) / seg;
    const uint32_t extent = (uint32_t)(nblocks * d_out_stride);
    memset(out_len, 0, sizeof(uint32_t) * nblocks);
    if (nseg == 0) return LZS_OK;
    /* host tables: 12 words and 3 bytes per segment */
    uint32_t *tab = (uint32_t *)malloc((size_t)nseg * (12 * 4 + 4));
    if (!tab) return fail(LZS_E_NOMEM, "%s: out of host memory", who);
    uint32_t *entry = tab, *exits = entry + nseg, *count = exits + nseg, *start = count + nseg;
    uint32_t *base = start + nseg, *end = base + nseg, *floor_ = end + nseg, *limit = floor_ + nseg;
    uint32_t *seen = limit + nseg, *blk = seen + nseg, *spare = blk + nseg;   /* (spare: two unused rows) */
    uint8_t *dirty = (uint8_t *)(spare + 2 * (size_t)nseg), *ones = dirty + nseg, *first = ones + nseg;
    void *d_aux = NULL, *d_marks = NULL, *d_origin = NULL;
#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto done; } } while (0)
    e = staging_reserve(st, BUF_AUX, (size_t)nseg * (8 * 4 + 2) + 128, &d_aux);
    if (!e) e = staging_reserve(st, BUF_MARKS, (size_t)nseg * LZS_SCAN_MARK_WORDS * 4u, &d_marks);
    if (!e) e = staging_reserve(st, BUF_KEEP, 4 * (size_t)extent + 64, &d_origin);
    if (e) { rc = fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto done; }
    uint32_t *d_entry = (uint32_t *)d_aux, *d_exit = d_entry + nseg, *d_count = d_exit + nseg, *d_start = d_count + nseg;
    uint32_t *d_base = d_start + nseg, *d_end = d_base + nseg, *d_floor = d_end + nseg, *d_limit = d_floor + nseg;
    uint32_t *d_counters = d_limit + nseg;
    uint8_t *d_dirty = (uint8_t *)(d_counters + 2), *d_ones = d_dirty + nseg;
    {
        uint32_t k = 0;
        for (size_t b = 0; b < nblocks; b++) {
            const uint32_t len = in_len_each ? in_len_each[b] : in_len;
            for (uint32_t at = 0; at < len; at += seg, k++) {
                base[k] = (uint32_t)(b * d_in_stride) + at;
                end[k] = (uint32_t)(b * d_in_stride) + len;
                floor_[k] = (uint32_t)(b * d_out_stride);
                limit[k] = floor_[k] + cap32;
                blk[k] = (uint32_t)b;
                first[k] = at == 0;
                entry[k] = 0; dirty[k] = 1; seen[k] = 0xFFFFFFFFu;
            }
        }
    }
    const int debug = getenv("LZS_STREAM_DEBUG") != NULL;
    double t0 = debug ? now_ms() : 0, t1;
    HIP_TRY(lzs_hip_h2d(d_base, base, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_end, end, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    for (uint32_t round = 0, ndirty = nseg; ndirty; round++) {
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_scan_stream(d_in, 0, nseg, d_entry, d_dirty, d_exit, d_count, round == 0 ? d_ones : NULL,
                                           (uint32_t *)d_marks, round != 0, seg, 0, d_base, d_end, stream), who);
        HIP_TRY(lzs_hip_d2h(exits, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        if (round == 0) HIP_TRY(lzs_hip_d2h(ones, d_ones, nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_d2h(count, d_count, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        for (uint32_t k = 0; k < nseg; k++) if (dirty[k]) seen[k] = entry[k];
        ndirty = 0;
        int ended = 0, settled = 1;
        for (uint32_t k = 0; k < nseg; k++) {                  /* as in stream_decompress(), block by block */
            dirty[k] = 0;
            if (first[k]) { ended = 0; settled = 1; continue; }
            uint32_t want = exits[k - 1];
            if (want & LZS_SEG_STOP) { if (settled) ended = 1; else want = entry[k]; }
            if (ended) want = LZS_SEG_STOP;
            if (want != entry[k]) settled = 0;
            entry[k] = want;
            if (ended || want == seen[k]) continue;
            if (((want >> 8) & 1u) && (ones[k] == 2 || (ones[k] && (want & 3u) == 0))) {
                const uint32_t r = want & 0xFFu;
                const uint32_t nibbles = (seg * 8u - r + 3u) / 4u;
                exits[k] = (want & ~0xFFu) | (r + 4u * nibbles - seg * 8u);
                count[k] = 15u * nibbles;
                seen[k] = want;
                continue;
            }
            dirty[k] = 1;
            ndirty++;
        }
        if (ndirty) {
            HIP_TRY(lzs_hip_h2d(d_exit, exits, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_h2d(d_count, count, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        }
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs batch decode: %zu blocks, %u segments of %u; round %u in %.2f ms, %u to redo\n", nblocks, nseg, seg, round, t1 - t0, ndirty); t0 = t1; }
    }
    {
        uint64_t total = 0;
        for (uint32_t k = 0; k < nseg; k++) {
            if (first[k]) total = 0;
            start[k] = floor_[k] + (uint32_t)(total < cap32 ? total : cap32);
            if (!(entry[k] & LZS_SEG_STOP)) total += count[k];
            out_len[blk[k]] = (uint32_t)(total < cap32 ? total : cap32);
        }
    }
    HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_start, start, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_floor, floor_, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_limit, limit, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_memset(d_counters, 0, 8, stream), "hipMemset");
    HIP_TRY(lzs_hip_memset(d_origin, 0xFF, 4 * (size_t)extent, stream), "hipMemset");      /* everything "clean" */
    HIP_TRY(lzs_hip_launch_decode_stream(d_out, extent, (uint32_t *)d_origin, d_counters, d_in, 0, nseg, d_entry, d_start,
                                         seg, 0, d_base, d_end, d_floor, d_limit, stream), who);
    {
        uint32_t open[2] = {0, 0};
        HIP_TRY(lzs_hip_d2h(open, d_counters, 8, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs batch decode: tables + memset + decode in %.2f ms, %u bytes with an origin elsewhere\n", t1 - t0, open[0]); t0 = t1; }
        /* origins never leave their block: one workgroup per block resolves them to the end */
        HIP_TRY(lzs_hip_h2d(d_len, out_len, sizeof(uint32_t) * nblocks, stream), "hipMemcpy H2D");
        if (open[0]) HIP_TRY(lzs_hip_launch_resolve_blocks(d_out, (uint32_t *)d_origin, d_out_stride, d_len, (uint32_t)nblocks, stream), who);
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs batch decode: resolve in %.2f ms\n", t1 - t0); t0 = t1; }
    }
#undef HIP_TRY
done:
    free(tab);
    return rc;
}

int lzs_decompress_stream_device(void *d_out, size_t out_cap, size_t *out_len, const void *d_in, size_t in_len)
{
    if (!out_len) return fail(LZS_E_ARG, "lzs_decompress_stream_device: out_len is NULL");
    *out_len = 0;
    if ((!d_out && out_cap) || (!d_in && in_len)) return fail(LZS_E_ARG, "lzs_decompress_stream_device: NULL buffer");
    if (in_len == 0 || out_cap == 0) return LZS_OK;
    if (in_len > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "lzs_decompress_stream_device: stream exceeds LZS_BLOCK_MAX");
    int rc = LZS_OK;
    const size_t got = stream_decompress((uint8_t *)d_out, out_cap, (const uint8_t *)d_in, in_len, 1, &rc, 0, NULL);
    if (got == SIZE_MAX) return fail(LZS_E_ARG, "lzs_decompress_stream_device: output of 4 GiB or more");
    *out_len = got;
    return rc;
}

size_t lzs_decompress(uint8_t *a_pOutData, size_t a_outBufferSize, const uint8_t *a_pInData, size_t a_inLen)
{
    if ((a_inLen > STREAM_DEC_MIN || (a_inLen && getenv("LZS_FORCE_STREAM"))) && a_inLen <= LZS_BLOCK_MAX &&
        a_pOutData && a_pInData && a_outBufferSize && !getenv("LZS_ONE_WAVE")) {
        const size_t got = stream_decompress(a_pOutData, a_outBufferSize, a_pInData, a_inLen, 0, NULL, 0, NULL);
        if (got != SIZE_MAX) return got;
    }
    return one_shot("lzs_decompress", lzs_hip_launch_decompress, a_pOutData, a_outBufferSize, a_pInData, a_inLen);
}

size_t lzs_decompress_concat(uint8_t *out, size_t out_cap, const uint8_t *in, size_t in_len)
{
    if ((in_len > STREAM_DEC_MIN || (in_len && getenv("LZS_FORCE_STREAM"))) && in_len <= LZS_BLOCK_MAX &&
        out && in && out_cap && !getenv("LZS_ONE_WAVE")) {
        const size_t got = stream_decompress(out, out_cap, in, in_len, 0, NULL, 1, NULL);
        if (got != SIZE_MAX) return got;
    }
    return one_shot("lzs_decompress_concat", lzs_hip_launch_decompress_concat, out, out_cap, in, in_len);
}

/* ------------------------------------------------------- incremental interface: decoding */
/* reference lzs-decompression.c:420-743.  What the reference keeps in its private members is
 * kept here, in the same bytes of the caller's block, in this layout; every call ships it to
 * the device with the input and back (lzs_decode_resume_kernel). */
typedef struct __attribute__((packed)) {
    uint32_t bitq;                  /* bits of an unfinished token, left-aligned */
    uint16_t off;                   /* copy in progress: offset */
    uint16_t hist_len;              /* bytes in hist[], oldest first */
    uint8_t  qlen, rem, extended;   /* bits in bitq; copy bytes left; a length nibble follows */
    uint8_t  hist[LZS_MAX_HISTORY_SIZE];
} dec_priv_t;
#define DEC_PRIV_AT 36u
#define DEC_SMALL     16384u        /* calls up to this much input and output take the short way */
#define INC_DEC_STREAM_MIN 16384u   /* pieces from this size on go to many wavefronts first */
#define DEC_STATE_PAD 2112u         /* sizeof(lzs_dec_resume_t) rounded up to 64 */
#define INC_BOX_BYTES (2 * (size_t)DEC_SMALL + DEC_STATE_PAD + 64)
_Static_assert(sizeof(lzs_dec_resume_t) <= DEC_STATE_PAD, "state fits its slot");
_Static_assert(sizeof(LzsDecompressParameters_t) == 2096, "size of the reference's LzsDecompressParameters_t");
_Static_assert(sizeof(LzsCompressParameters_t) == 14432, "size of the reference's LzsCompressParameters_t");
_Static_assert(DEC_PRIV_AT + sizeof(dec_priv_t) <= sizeof(LzsDecompressParameters_t), "private state fits");

void lzs_decompress_init(LzsDecompressParameters_t *p)
{
    if (!p) return;
    p->status = LZS_D_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
}

size_t lzs_decompress_incremental(LzsDecompressParameters_t *p)
{
    const char *who = "lzs_decompress_incremental";
    if (!p) return 0;
    dec_priv_t *pv = (dec_priv_t *)((uint8_t *)p + DEC_PRIV_AT);
    size_t made = 0;
    int e = 0;
    p->status = LZS_D_STATUS_NONE;
    tls_error[0] = 0;
    /* nothing to read and no bit queued: the answer needs no device (:475-478; the reference
     * stops there even with a copy pending) */
    if (p->inLength == 0 && pv->qlen == 0) {
        p->status = LZS_D_STATUS_INPUT_FINISHED | LZS_D_STATUS_INPUT_STARVED;
        return 0;
    }
    if ((p->inLength && !p->inPtr) || (p->outLength && !p->outPtr)) {
        fail(LZS_E_ARG, "%s: NULL buffer", who);
        p->status = LZS_D_STATUS_ERROR;
        return 0;
    }
    if (require_device() != LZS_OK) goto failed;
    staging_t *st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
#define HIP_TRY(call, what) do { e = (call); if (e) { hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    lzs_dec_resume_t h;
    memset(&h, 0, sizeof(h));
    size_t wave_limit = (size_t)16 << 20;
    for (;;) {
        /* A large piece first goes to many wavefronts (stream_decompress) as far as
         * whole segments can be decoded; what is left -- the segment with the end marker, the
         * unfinished token at the end of the input, the last bytes before the output is full, a
         * copy still running -- is the one wavefront's below. */
        if (pv->rem == 0 && p->inLength >= INC_DEC_STREAM_MIN && p->outLength >= 4096u && !getenv("LZS_ONE_WAVE")) {
            const size_t big = p->inLength < ((size_t)256 << 20) ? p->inLength : ((size_t)256 << 20);
            const uint32_t nb = (pv->qlen + 7u) / 8u;
            uint8_t pre[4] = {0, 0, 0, 0};
            const uint32_t v = pv->qlen ? pv->bitq >> (32u - pv->qlen) : 0u;     /* the queued bits, right-aligned */
            for (uint32_t i = 0; i < nb; i++) pre[i] = (uint8_t)(v >> (8u * (nb - 1u - i)));
            dec_piece_t dp;
            memset(&dp, 0, sizeof(dp));
            dp.prefix = pre; dp.prefix_len = nb;
            dp.entry0 = (8u * nb - pv->qlen) | ((uint32_t)(pv->extended != 0) << 8) | ((pv->extended ? (uint32_t)pv->off : 0u) << 9);
            dp.hist = pv->hist; dp.hist_len = pv->hist_len;
            const size_t room = p->outLength < 0xE0000000u ? p->outLength : 0xE0000000u;
            int rc = LZS_OK;
            const size_t got = stream_decompress(p->outPtr, room, p->inPtr, nb + big, 0, &rc, 0, &dp);
            if (rc != LZS_OK) { p->status = LZS_D_STATUS_ERROR; return made; }
            if (got != SIZE_MAX && dp.segs_done > 0) {
                const size_t at = (size_t)dp.segs_done * dp.seg + ((dp.next_entry & 0xFFu) >> 3);   /* in prefix + input */
                const uint32_t b = dp.next_entry & 7u;
                if (at < nb || at - nb + (b ? 1u : 0u) > big || (dp.next_entry & LZS_SEG_STOP) || got > room) {
                    fail(LZS_E_HIP, "%s: inconsistent state from the device", who);
                    goto failed;
                }
                pv->qlen = b ? (uint8_t)(8u - b) : 0;
                pv->bitq = b ? (uint32_t)p->inPtr[at - nb] << (24u + b) : 0u;
                pv->extended = (uint8_t)((dp.next_entry >> 8) & 1u);
                if (pv->extended) pv->off = (uint16_t)((dp.next_entry >> 9) & 0x7FFu);
                /* the history: the last 2047 bytes of what was there and what came now */
                if (got >= LZS_MAX_HISTORY_SIZE) {
                    memcpy(pv->hist, p->outPtr + got - LZS_MAX_HISTORY_SIZE, LZS_MAX_HISTORY_SIZE);
                    pv->hist_len = LZS_MAX_HISTORY_SIZE;
                } else {
                    const size_t keep = (size_t)pv->hist_len + got > LZS_MAX_HISTORY_SIZE ? LZS_MAX_HISTORY_SIZE - got : pv->hist_len;
                    memmove(pv->hist, pv->hist + pv->hist_len - keep, keep);
                    memcpy(pv->hist + keep, p->outPtr, got);
                    pv->hist_len = (uint16_t)(keep + got);
                }
                const size_t used = at - nb + (b ? 1u : 0u);
                p->inPtr += used;  p->inLength -= used;
                p->outPtr += got;  p->outLength -= got;
                made += got;
                if (p->inLength == 0 && pv->qlen == 0) { p->status = LZS_D_STATUS_INPUT_FINISHED | LZS_D_STATUS_INPUT_STARVED; break; }
            }
            /* not even one whole segment this time: the wavefront takes the next stretch */
            wave_limit = 4u * (size_t)(dp.seg ? dp.seg : 8192u);
        }
        /* one launch takes at most 16 MiB of input; its output is bounded by 30x that
         * (a length nibble stands for 15 bytes) */
        const size_t take = p->inLength < wave_limit ? p->inLength : wave_limit;
        const size_t most = 30u * (take + 4u) + 64u;
        const size_t cap = p->outLength < most ? p->outLength : most;
        void *d_in = NULL, *d_out = NULL, *d_state = NULL;
        h.bitq = pv->bitq; h.qlen = pv->qlen; h.off = pv->off; h.rem = pv->rem;
        h.extended = pv->extended; h.hist_len = pv->hist_len;
        memcpy(h.hist, pv->hist, pv->hist_len);
        if (take <= DEC_SMALL && cap <= DEC_SMALL) {
            /* A small call is all latency: one copy in ([input | state], the input right-aligned
             * before the state), one launch, one copy out ([state | output]), one wait. */
            if (!st->host_box) st->host_box = (uint8_t *)malloc(INC_BOX_BYTES);
            uint8_t *box = st->host_box;
            if (!box) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
            uint8_t *d_box = NULL;
            e = staging_reserve(st, BUF_AUX, INC_BOX_BYTES, (void **)&d_box);
            if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
            const size_t in_at = DEC_SMALL - ((take + 3u) & ~(size_t)3u);
            memcpy(box + in_at, p->inPtr, take);
            memcpy(box + DEC_SMALL, &h, sizeof(h));
            HIP_TRY(lzs_hip_h2d(d_box + in_at, box + in_at, DEC_SMALL - in_at + sizeof(h), stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_launch_decode_resume((lzs_dec_resume_t *)(d_box + DEC_SMALL), d_box + in_at, (uint32_t)take,
                                                 d_box + DEC_SMALL + DEC_STATE_PAD, (uint32_t)cap, stream), who);
            HIP_TRY(lzs_hip_d2h(box + DEC_SMALL, d_box + DEC_SMALL, DEC_STATE_PAD + cap, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            memcpy(&h, box + DEC_SMALL, sizeof(h));
            if (h.in_used > take || h.out_made > cap || h.hist_len > LZS_MAX_HISTORY_SIZE) {
                fail(LZS_E_HIP, "%s: inconsistent state from the device", who);
                goto failed;
            }
            memcpy(p->outPtr, box + DEC_SMALL + DEC_STATE_PAD, h.out_made);
        } else {
            e = staging_reserve(st, BUF_IN, take + 64, &d_in);
            if (!e) e = staging_reserve(st, BUF_OUT, cap + 64, &d_out);
            if (!e) e = staging_reserve(st, BUF_AUX, sizeof(h), &d_state);
            if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
            HIP_TRY(lzs_hip_h2d(d_state, &h, sizeof(h), stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_h2d(d_in, p->inPtr, take, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_launch_decode_resume((lzs_dec_resume_t *)d_state, d_in, (uint32_t)take, d_out, (uint32_t)cap, stream), who);
            HIP_TRY(lzs_hip_d2h(&h, d_state, sizeof(h), stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            if (h.in_used > take || h.out_made > cap || h.hist_len > LZS_MAX_HISTORY_SIZE) {
                fail(LZS_E_HIP, "%s: inconsistent state from the device", who);
                goto failed;
            }
            HIP_TRY(lzs_hip_d2h(p->outPtr, d_out, h.out_made, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        }
        pv->bitq = h.bitq; pv->qlen = (uint8_t)h.qlen; pv->off = (uint16_t)h.off; pv->rem = (uint8_t)h.rem;
        pv->extended = (uint8_t)h.extended; pv->hist_len = (uint16_t)h.hist_len;
        memcpy(pv->hist, h.hist, h.hist_len);
        p->inPtr += h.in_used;   p->inLength -= h.in_used;
        p->outPtr += h.out_made; p->outLength -= h.out_made;
        made += h.out_made;
        /* stopped only because of this loop's own limits: go on */
        if ((h.status & LZS_INC_INPUT_STARVED) && p->inLength) continue;
        if ((h.status & LZS_INC_NO_OUTPUT_SPACE) && p->outLength && cap == most) continue;
        p->status = (uint8_t)h.status;
        break;
    }
#undef HIP_TRY
    staging_trim(st);
    return made;

failed:
    p->status = LZS_D_STATUS_ERROR;
    fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
    return made;
}

/* ---------------------------------------------------- incremental interface: compression */
/* reference lzs-compression.c:479-823.  Between calls the caller's block holds, in place of the
 * reference's ring and hash tables: the last INC_HIST bytes already encoded (what the next
 * piece's chains are built from: lzs_compress_segments_kernel warms up over 2176 + 64 bytes
 * before its first token), the <= 15 bytes after them that wait for more look-ahead, the offset
 * of a long match still running, the bits of the last, partial output byte, and output that
 * found no room.  Every call encodes what the data so far decides, as one piece of the stream
 * on the device (stream_compress_piece). */
#define INC_HIST      2304u
#define INC_CARRY_MAX 16u
#define INC_PEND_MAX  8192u
typedef struct __attribute__((packed)) {
    uint32_t data_len;              /* bytes in data[]: history, then carry_len bytes not yet encoded */
    uint32_t carry_len;
    uint32_t pend_pos, pend_len;    /* output waiting in pend[pend_pos .. pend_len) */
    uint16_t ext_off;               /* != 0: inside a long match at this offset */
    uint8_t  bit_len, bit_val;      /* bits of the partial last output byte, left-aligned */
    uint8_t  marker_waiting;        /* the end marker is among the waiting output */
    uint8_t  data[INC_HIST + INC_CARRY_MAX];
    uint8_t  pend[INC_PEND_MAX];
} enc_priv_t;
#define ENC_PRIV_AT 40u
_Static_assert(ENC_PRIV_AT + sizeof(enc_priv_t) <= sizeof(LzsCompressParameters_t), "private state fits");

void lzs_compress_init_full(LzsCompressParameters_t *p)
{
    if (!p) return;
    p->status = LZS_C_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
}

void lzs_compress_init_quick(LzsCompressParameters_t *p) { lzs_compress_init_full(p); }

/* hand `len` bytes to the caller's buffer, what does not fit to pend[] (room was reserved) */
static size_t inc_deliver(LzsCompressParameters_t *p, enc_priv_t *pv, const uint8_t *src, size_t len)
{
    const size_t now = len < p->outLength ? len : p->outLength;
    memcpy(p->outPtr, src, now);
    p->outPtr += now; p->outLength -= now;
    memcpy(pv->pend, src + now, len - now);
    pv->pend_pos = 0; pv->pend_len = (uint32_t)(len - now);
    return now;
}

size_t lzs_compress_incremental(LzsCompressParameters_t *p, bool add_end_marker)
{
    const char *who = "lzs_compress_incremental";
    if (!p) return 0;
    enc_priv_t *pv = (enc_priv_t *)((uint8_t *)p + ENC_PRIV_AT);
    size_t made = 0;
    uint8_t *tmp = NULL;
    p->status = LZS_C_STATUS_NONE;
    tls_error[0] = 0;
    if ((p->inLength && !p->inPtr) || (p->outLength && !p->outPtr) ||
        pv->data_len > sizeof(pv->data) || pv->carry_len > pv->data_len || pv->pend_len > INC_PEND_MAX || pv->pend_pos > pv->pend_len) {
        fail(LZS_E_ARG, "%s: NULL buffer or a parameter block that was not initialised", who);
        p->status = LZS_C_STATUS_ERROR;
        return 0;
    }
    /* output still waiting from the call before goes first (:574-588) */
    if (pv->pend_pos < pv->pend_len) {
        const size_t have = pv->pend_len - pv->pend_pos;
        const size_t now = have < p->outLength ? have : p->outLength;
        memcpy(p->outPtr, pv->pend + pv->pend_pos, now);
        p->outPtr += now; p->outLength -= now; pv->pend_pos += (uint32_t)now; made += now;
        if (pv->pend_pos < pv->pend_len) {
            p->status = LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE;
            return made;
        }
        pv->pend_pos = pv->pend_len = 0;
        if (pv->marker_waiting) {
            pv->marker_waiting = 0;
            p->status = LZS_C_STATUS_END_MARKER | (p->inLength ? 0 : LZS_C_STATUS_INPUT_FINISHED | LZS_C_STATUS_INPUT_STARVED);
            return made;
        }
    }
    for (;;) {
        /* Take as much input as the room for its output allows: 9 bits a byte at worst, into the
         * caller's buffer and then into pend[]; one piece is at most 1 GiB. */
        const size_t room = (p->outLength < ((size_t)1 << 40) ? p->outLength : ((size_t)1 << 40)) + INC_PEND_MAX;
        const size_t fits = (8u * room - 64u) / 9u - pv->carry_len;
        size_t take = p->inLength < fits ? p->inLength : fits;
        if (take > ((size_t)1 << 30)) take = (size_t)1 << 30;
        const int last = add_end_marker && take == p->inLength;
        const size_t n = (size_t)pv->data_len + take;
        const uint32_t c0 = pv->data_len - pv->carry_len;
        if (!last && n - c0 <= LZS_MAX_LOOK_AHEAD_LEN) {
            /* too little to decide the next token (:641-647): it waits in the block */
            memcpy(pv->data + pv->data_len, p->inPtr, take);
            pv->data_len += (uint32_t)take; pv->carry_len += (uint32_t)take;
            p->inPtr += take; p->inLength -= take;
            break;
        }
        piece_t pc;
        memset(&pc, 0, sizeof(pc));
        pc.prefix = pv->data; pc.prefix_len = pv->data_len;
        pc.c0 = c0; pc.ext_off = pv->ext_off; pc.bit0 = pv->bit_len; pc.first = pv->bit_val; pc.last = last;
        const size_t cap = LZS_COMPRESSED_MAX(n - c0) + 16;
        tmp = (uint8_t *)malloc(cap);
        if (!tmp) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
        int rc = LZS_OK;
        const size_t got = stream_compress_piece(tmp, cap, p->inPtr, n, 0, &rc, &pc);
        if (rc != LZS_OK) goto failed_quiet;
        const size_t whole = last ? got : (size_t)(pc.nbits / 8);
        if (whole > got || whole > room || pc.c_exit > n || pc.c_exit < c0 || (last ? pc.c_exit != n : n - pc.c_exit > INC_CARRY_MAX - 1u)) {
            fail(LZS_E_HIP, "%s: inconsistent state from the device (piece of %zu bytes from %u: %zu bytes out, %llu bits, ends at %u, room %zu)",
                 who, n, c0, got, (unsigned long long)pc.nbits, pc.c_exit, room);
            goto failed;
        }
        made += inc_deliver(p, pv, tmp, whole);
        pv->bit_len = last ? 0 : (uint8_t)(pc.nbits & 7u);
        pv->bit_val = pv->bit_len ? (uint8_t)(tmp[whole] & (0xFF00u >> pv->bit_len)) : 0;
        pv->ext_off = (uint16_t)pc.ext_exit;
        free(tmp); tmp = NULL;
        /* the new history and carry: bytes [c_exit - INC_HIST, n) of prefix + input */
        {
            const size_t from = pc.c_exit > INC_HIST ? pc.c_exit - INC_HIST : 0;
            uint8_t keep[INC_HIST + INC_CARRY_MAX];
            size_t k = 0;
            for (size_t i = from; i < n; ) {
                if (i < pv->data_len) { const size_t m = (pv->data_len < n ? pv->data_len : n) - i; memcpy(keep + k, pv->data + i, m); k += m; i += m; }
                else { const size_t m = n - i; memcpy(keep + k, p->inPtr + (i - pv->data_len), m); k += m; i += m; }
            }
            memcpy(pv->data, keep, k);
            pv->data_len = (uint32_t)k;
            pv->carry_len = (uint32_t)(n - pc.c_exit);
        }
        p->inPtr += take; p->inLength -= take;
        if (last) {
            if (pv->pend_len) pv->marker_waiting = 1;
            else p->status |= LZS_C_STATUS_END_MARKER;
            break;
        }
        if (pv->pend_len || p->inLength == 0) break;
    }
    if (pv->pend_len) p->status |= LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE;
    if (p->inLength == 0) p->status |= LZS_C_STATUS_INPUT_FINISHED | LZS_C_STATUS_INPUT_STARVED;
    return made;

failed:
    fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
failed_quiet:
    free(tmp);
    p->status = LZS_C_STATUS_ERROR;
    return made;
}
