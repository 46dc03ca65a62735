/*
 * tests/cpu_shim/san_driver.c -- drives the product's UNCHANGED host sources (linked with lzs_cpu_shim.c instead of the
 * HIP translation unit) through the cases of the GPU suite that exercise the intricate host logic, under the
 * sanitizers (tests/test_sanitizers.py builds this with -fsanitize=address,undefined and with -fsanitize=thread):
 *   ragged batches with cut capacities (one-after-the-other route); a batch of >= 24 MiB through the overlapped pipeline
 *   (pinned pieces, four worker threads, launches by the group) both ways, from two calling threads at once; the pipeline's
 *   degrade path (LZS_STAGING_FAIL_MB); one stream in segments with dirty-segment re-entry (runs and periods spanning
 *   many segments, cut capacities); the incremental interface in random pieces both ways on BOTH routes (device route:
 *   segments / stitch / extend-resume / decode-resume; host route: lzs_hostcodec.c) with 1-12 bytes of room for the
 *   low-memory block; the one-shot calls on both routes.
 * Every result is compared with the oracle.  Exit code 0 = all cases passed (the sanitizers make it non-zero themselves).
 * TEST INFRASTRUCTURE ONLY.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lzs/lzs.h"
#include "lzs/lzs_batch.h"

size_t lzs_oracle_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);
size_t lzs_oracle_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);
int lzs_workload_fill(uint8_t *dst, unsigned cls, uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_len, int nthreads);

static int failures;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; fprintf(stderr, "FAIL %s:%d: ", __func__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 24); }
static uint32_t rnd_in(uint32_t lo, uint32_t hi) { return lo + rnd() % (hi - lo + 1u); }

static uint8_t *sample(unsigned cls, size_t n, uint64_t first)
{
    const size_t nb = (n + 65535u) / 65536u;
    uint8_t *d = (uint8_t *)malloc(nb * 65536u + 1);
    lzs_workload_fill(d, cls, 0x4C5A5331ull, first, nb, 65536, 2);
    return d;
}

/* ---------------------------------------------------------------- ragged batches, cut capacities */
static void ragged_batches(void)
{
    enum { NB = 37, STRIDE = 5000 };
    uint8_t *x = sample(0, (size_t)NB * STRIDE, 3);
    uint32_t lens[NB];
    for (int b = 0; b < NB; b++) lens[b] = b % 7 == 0 ? 0 : rnd_in(1, STRIDE);
    lens[5] = STRIDE;
    for (int pass = 0; pass < 3; pass++) {
        const size_t cap = pass == 0 ? LZS_COMPRESSED_MAX(STRIDE) : pass == 1 ? 700 : 1;
        const size_t ostride = cap + 9;
        uint8_t *out = (uint8_t *)malloc(NB * ostride);
        uint32_t out_len[NB];
        memset(out, 0xA5, NB * ostride);
        const int rc = lzs_compress_batch(out, ostride, cap, out_len, x, STRIDE, lens, STRIDE, NB);
        CHECK(rc == 0, "lzs_compress_batch: %s", lzs_last_error());
        uint8_t *want = (uint8_t *)malloc(LZS_COMPRESSED_MAX(STRIDE));
        for (int b = 0; b < NB; b++) {
            const size_t w = lzs_oracle_compress(want, cap, x + (size_t)b * STRIDE, lens[b]);
            CHECK(out_len[b] == w && memcmp(out + b * ostride, want, w) == 0, "block %d differs (cap %zu)", b, cap);
            for (size_t i = w; i < ostride; i++) if (out[b * ostride + i] != 0xA5) { CHECK(0, "block %d: byte %zu past its length touched", b, i); break; }
        }
        if (pass == 0) {
            const size_t bcap = 3000, bstride = bcap + 5;           /* and back, into slots too small for some */
            uint8_t *back = (uint8_t *)malloc(NB * bstride);
            uint32_t back_len[NB];
            memset(back, 0x5A, NB * bstride);
            CHECK(lzs_decompress_batch(back, bstride, bcap, back_len, out, ostride, out_len, cap, NB) == 0, "lzs_decompress_batch: %s", lzs_last_error());
            for (int b = 0; b < NB; b++) {
                const uint32_t w = lens[b] < bcap ? lens[b] : (uint32_t)bcap;
                CHECK(back_len[b] == w && memcmp(back + b * bstride, x + (size_t)b * STRIDE, w) == 0, "block %d: round trip", b);
                CHECK(back[b * bstride + bcap] == 0x5A, "block %d: byte past the capacity touched", b);
            }
            free(back);
        }
        free(want); free(out);
    }
    free(x);
}

/* ---------------------------------------------------------------- the overlapped pipeline (>= 24 MiB), two threads at once */
typedef struct { unsigned cls; int id; } pipe_job_t;
static void *pipeline_once(void *argp)
{
    const pipe_job_t *job = (const pipe_job_t *)argp;
    enum { NB = 416 };                                              /* 26 MiB of input: four chunks of 128 and a short fifth */
    const size_t cap = LZS_COMPRESSED_MAX(65536), ostride = cap + 13;
    uint8_t *x = sample(job->cls, (size_t)NB * 65536, 100 * (uint64_t)job->id);
    uint8_t *out = (uint8_t *)malloc(NB * ostride), *want = (uint8_t *)malloc(cap);
    uint32_t *out_len = (uint32_t *)calloc(NB, 4), *lens = (uint32_t *)malloc(NB * 4);
    for (int b = 0; b < NB; b++) lens[b] = b % 5 == 0 ? 65536 : 1 + (uint32_t)((b * 2654435761u) % 65536u);
    for (int ragged = 0; ragged < 2; ragged++) {
        memset(out, 0xA5, NB * ostride);
        const int rc = lzs_compress_batch(out, ostride, cap, out_len, x, 65536, ragged ? lens : NULL, 65536, NB);
        CHECK(rc == 0, "thread %d: lzs_compress_batch: %s", job->id, lzs_last_error());
        for (int b = 0; b < NB; b += (b < 8 ? 1 : 23)) {
            const size_t w = lzs_oracle_compress(want, cap, x + (size_t)b * 65536, ragged ? lens[b] : 65536);
            CHECK(out_len[b] == w && memcmp(out + b * ostride, want, w) == 0, "thread %d: block %d differs", job->id, b);
            CHECK(out[b * ostride + w] == 0xA5, "thread %d: block %d: byte past its length touched", job->id, b);
        }
        uint8_t *back = (uint8_t *)malloc((size_t)NB * 65540);
        uint32_t *back_len = (uint32_t *)calloc(NB, 4);
        memset(back, 0x5A, (size_t)NB * 65540);
        CHECK(lzs_decompress_batch(back, 65540, 65536, back_len, out, ostride, out_len, cap, NB) == 0, "thread %d: lzs_decompress_batch: %s", job->id, lzs_last_error());
        for (int b = 0; b < NB; b++) {
            const uint32_t n = ragged ? lens[b] : 65536;
            if (back_len[b] != n || memcmp(back + (size_t)b * 65540, x + (size_t)b * 65536, n) != 0) { CHECK(0, "thread %d: block %d: round trip", job->id, b); break; }
            if (back[(size_t)b * 65540 + 65536] != 0x5A) { CHECK(0, "thread %d: block %d: past the capacity", job->id, b); break; }
        }
        free(back); free(back_len);
    }
    free(x); free(out); free(want); free(out_len); free(lens);
    return NULL;
}
static void pipeline(void)
{
    pthread_t th[2];
    pipe_job_t jobs[2] = { { 1, 0 }, { 1, 1 } };                    /* (low-entropy blocks: the oracle behind the launches is fast on them) */
    for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, pipeline_once, &jobs[i]);
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
    /* the degrade path: the ring does not fit, the batch takes the one-after-the-other route and succeeds */
    setenv("LZS_STAGING_FAIL_MB", "30", 1);
    pipe_job_t one = { 1, 2 };
    pipeline_once(&one);
    unsetenv("LZS_STAGING_FAIL_MB");
}

/* ---------------------------------------------------------------- one stream in segments (dirty re-entry), both routes */
static void one_shot_streams(void)
{
    uint8_t *t = sample(0, 300000, 7), *r = sample(2, 70000, 9);
    const size_t n_mix = 50000 + 70000 + 40000 + 80000 + 30000 + 5;
    uint8_t *mix = (uint8_t *)calloc(n_mix, 1);
    size_t at = 0;
    memcpy(mix, t, 50000); at = 50000 + 70000;                      /* 70 000 zeros: a run over many segments */
    memcpy(mix + at, t + 50000, 40000); at += 40000;
    for (size_t i = 0; i < 80000; i++) mix[at + i] = "ab"[i & 1];   /* period 2 over many segments */
    at += 80000;
    memcpy(mix + at, r, 30000); at += 30000 + 5;
    const struct { const uint8_t *d; size_t n; } cases[] = { { t, 1 }, { t, 4096 }, { t, 6145 }, { t, 20000 }, { t, 300000 }, { mix, n_mix }, { r, 70000 } };
    const char *routes[] = { "device", "host" };
    for (unsigned ri = 0; ri < 2; ri++) {
        setenv("LZS_ROUTE", routes[ri], 1);
        for (unsigned ci = 0; ci < sizeof cases / sizeof cases[0]; ci++) {
            const size_t n = cases[ci].n, cap = LZS_COMPRESSED_MAX(n);
            uint8_t *want = (uint8_t *)malloc(cap), *got = (uint8_t *)malloc(cap + 8), *back = (uint8_t *)malloc(n + 8);
            const size_t w = lzs_oracle_compress(want, cap, cases[ci].d, n);
            memset(got, 0xA5, cap + 8);
            const size_t g = lzs_compress(got, cap, cases[ci].d, n);
            CHECK(g == w && memcmp(got, want, w) == 0 && got[cap] == 0xA5, "%s route: lzs_compress of %zu bytes differs (%zu / %zu): %s", routes[ri], n, g, w, lzs_last_error());
            for (int k = 0; k < 3; k++) {                            /* cut capacities: the stream is cut, never altered */
                const size_t cut = k == 0 ? 0 : rnd_in(1, (uint32_t)w);
                memset(got, 0xA5, cap + 8);
                const size_t gc = lzs_compress(got, cut, cases[ci].d, n);
                CHECK(gc == cut && memcmp(got, want, cut) == 0 && got[cut] == 0xA5, "%s route: lzs_compress of %zu bytes into %zu differs", routes[ri], n, cut);
            }
            memset(back, 0x5A, n + 8);
            const size_t b = lzs_decompress(back, n + 3, want, w);
            CHECK(b == n && memcmp(back, cases[ci].d, n) == 0 && back[n + 3] == 0x5A, "%s route: lzs_decompress to %zu bytes differs (%zu)", routes[ri], n, b);
            const size_t half = lzs_decompress(back, n / 2, want, w);
            CHECK(half == n / 2 && memcmp(back, cases[ci].d, n / 2) == 0, "%s route: lzs_decompress into half the room", routes[ri]);
            free(want); free(got); free(back);
        }
    }
    unsetenv("LZS_ROUTE");
    free(t); free(r); free(mix);
}

/* ---------------------------------------------------------------- the incremental interface in random pieces, both routes */
static size_t encode_pieces(const uint8_t *d, size_t n, uint8_t *out, size_t out_cap, uint32_t in_lo, uint32_t in_hi, uint32_t out_lo, uint32_t out_hi, int simple)
{
    LzsCompressParameters_t *p = (LzsCompressParameters_t *)malloc(sizeof *p);
    LzsSimpleCompressParameters_t *q = (LzsSimpleCompressParameters_t *)malloc(sizeof *q);
    if (simple) lzs_simple_compress_init(q); else lzs_compress_init(p);
    size_t pos = 0, made = 0, pending = 0, calls = 0;
    const uint8_t *pend_ptr = d;
    int finish = 0;
    uint8_t status = 0;
    while (!(status & LZS_C_STATUS_END_MARKER)) {
        if (!pending && !finish) {
            if (pos < n) { pending = rnd_in(in_lo, in_hi); if (pending > n - pos) pending = n - pos; pend_ptr = d + pos; pos += pending; }
            if (!pending && pos >= n) finish = 1;
        }
        size_t room = rnd_in(out_lo, out_hi);
        if (room > out_cap - made) room = out_cap - made;
        size_t got;
        if (simple) {
            q->inPtr = pend_ptr; q->inLength = pending; q->outPtr = out + made; q->outLength = room;
            got = lzs_simple_compress_incremental(q, finish);
            pend_ptr = q->inPtr; pending = q->inLength; status = q->status;
        } else {
            p->inPtr = pend_ptr; p->inLength = pending; p->outPtr = out + made; p->outLength = room;
            got = lzs_compress_incremental(p, finish);
            pend_ptr = p->inPtr; pending = p->inLength; status = p->status;
        }
        made += got;
        if (status & LZS_C_STATUS_ERROR) { CHECK(0, "incremental compression reported ERROR: %s", lzs_last_error()); break; }
        if (++calls > 4000000) { CHECK(0, "incremental compression: no progress"); break; }
    }
    free(p); free(q);
    return made;
}

static size_t decode_pieces(const uint8_t *s, size_t n, uint8_t *out, size_t out_cap, uint32_t in_lo, uint32_t in_hi, uint32_t out_lo, uint32_t out_hi)
{
    LzsDecompressParameters_t *p = (LzsDecompressParameters_t *)malloc(sizeof *p);
    lzs_decompress_init(p);
    size_t pos = 0, made = 0, pending = 0, calls = 0;
    const uint8_t *pend_ptr = s;
    for (;;) {
        if (!pending && pos < n) { pending = rnd_in(in_lo, in_hi); if (pending > n - pos) pending = n - pos; pend_ptr = s + pos; pos += pending; }
        size_t room = rnd_in(out_lo, out_hi);
        if (room > out_cap - made) room = out_cap - made;
        p->inPtr = pend_ptr; p->inLength = pending; p->outPtr = out + made; p->outLength = room;
        made += lzs_decompress_incremental(p);
        pend_ptr = p->inPtr; pending = p->inLength;
        if (p->status & LZS_D_STATUS_ERROR) { CHECK(0, "incremental decompression reported ERROR: %s", lzs_last_error()); break; }
        if (!pending && pos >= n && (p->status & LZS_D_STATUS_INPUT_STARVED)) break;
        if (++calls > 4000000) { CHECK(0, "incremental decompression: no progress"); break; }
    }
    free(p);
    return made;
}

static void incremental(void)
{
    uint8_t *t = sample(0, 200000, 11), *low = sample(1, 200000, 12);
    const size_t n_mix = 30000 + 40000 + 30000 + 20000;
    uint8_t *mix = (uint8_t *)calloc(n_mix, 1);
    memcpy(mix, t, 30000); memcpy(mix + 70000, t + 30000, 30000);
    for (size_t i = 0; i < 20000; i++) mix[100000 + i] = "abc"[i % 3];
    const struct { const uint8_t *d; size_t n; } cases[] = { { t, 507 }, { t, 90000 }, { low, 200000 }, { mix, n_mix } };
    const char *routes[] = { "device", "host" };
    for (unsigned ri = 0; ri < 2; ri++) {
        setenv("LZS_ROUTE", routes[ri], 1);
        for (unsigned ci = 0; ci < sizeof cases / sizeof cases[0]; ci++) {
            const size_t n = cases[ci].n, cap = LZS_COMPRESSED_MAX(n) + 64;
            uint8_t *want = (uint8_t *)malloc(cap), *got = (uint8_t *)malloc(cap), *back = (uint8_t *)malloc(n + 64);
            const size_t w = lzs_oracle_compress(want, cap, cases[ci].d, n);
            const uint32_t ranges[][4] = { { 1, 60, 3, 60 }, { 100, 5000, 100, 5000 }, { 20000, 90000, 20000, 90000 }, { 512, 512, 512, 512 } };
            for (unsigned k = 0; k < 4; k++) {
                if (ranges[k][0] == 1 && n > 100000) continue;      /* (tiny pieces of the long cases: time) */
                const size_t g = encode_pieces(cases[ci].d, n, got, cap, ranges[k][0], ranges[k][1], ranges[k][2], ranges[k][3], 0);
                CHECK(g == w && memcmp(got, want, w) == 0, "%s route: incremental compression of case %u in pieces of %u..%u differs (%zu / %zu)", routes[ri], ci, ranges[k][0], ranges[k][1], g, w);
                const size_t b = decode_pieces(want, w, back, n + 64, ranges[k][0], ranges[k][1], ranges[k][2], 3 * ranges[k][3]);
                CHECK(b == n && memcmp(back, cases[ci].d, n) == 0, "%s route: incremental decompression of case %u in pieces of %u..%u differs (%zu / %zu)", routes[ri], ci, ranges[k][0], ranges[k][1], b, n);
            }
            if (n <= 90000) {                                       /* the low-memory block: 1-12 bytes of room, random pieces */
                const size_t m = n < 20000 ? n : 20000;
                const size_t w2 = lzs_oracle_compress(want, cap, cases[ci].d, m);
                const size_t g = encode_pieces(cases[ci].d, m, got, cap, 1, 300, 1, 12, 1);
                CHECK(g == w2 && memcmp(got, want, w2) == 0, "%s route: lzs_simple_compress_incremental of case %u with 1..12 bytes of room differs (%zu / %zu)", routes[ri], ci, g, w2);
            }
            free(want); free(got); free(back);
        }
    }
    unsetenv("LZS_ROUTE");
    free(t); free(low); free(mix);
}

/* ---------------------------------------------------------------- what a thread keeps, and giving it back
 * The reference keeps nothing after return (lzs.h:218,229); this build keeps a thread's staging for its next call -- at most
 * LZS_KEEP_MAX_MB in sum -- until the thread exits or calls lzs_release_thread_cache(). */
void lzs_shim_outstanding(size_t *dev_bytes, size_t *host_bytes, size_t *live);
typedef struct { int releases; size_t dev_after_calls, dev_after_release, live_after_release; } keeper_t;
static void *keeper(void *arg)
{
    keeper_t *k = (keeper_t *)arg;
    const size_t n = 300000;
    uint8_t *t = sample(0, n, 40), *got = (uint8_t *)malloc(LZS_COMPRESSED_MAX(n)), *back = (uint8_t *)malloc(n + 4);
    setenv("LZS_ROUTE", "device", 1);
    const size_t g = lzs_compress(got, LZS_COMPRESSED_MAX(n), t, n);                 /* one stream in segments: staging, tables, a stream */
    const size_t b = lzs_decompress(back, n, got, g);
    CHECK(b == n && memcmp(back, t, n) == 0, "keeper: round trip of %zu bytes (%zu, %zu)", n, g, b);
    enum { NB = 24, STRIDE = 4096 };
    uint32_t lens[NB]; for (int i = 0; i < NB; i++) lens[i] = STRIDE - (uint32_t)i;
    uint8_t *out = (uint8_t *)malloc((size_t)NB * LZS_COMPRESSED_MAX(STRIDE)); uint32_t out_len[NB];
    CHECK(lzs_compress_batch(out, LZS_COMPRESSED_MAX(STRIDE), LZS_COMPRESSED_MAX(STRIDE), out_len, t, STRIDE, lens, STRIDE, NB) == 0, "keeper: batch: %s", lzs_last_error());
    lzs_shim_outstanding(&k->dev_after_calls, NULL, NULL);
    if (k->releases) {
        lzs_release_thread_cache();
        lzs_release_thread_cache();                                                  /* (twice is once) */
        lzs_shim_outstanding(&k->dev_after_release, NULL, &k->live_after_release);
        /* ... and the thread's next call starts from nothing and works */
        const size_t g2 = lzs_compress(got, LZS_COMPRESSED_MAX(n), t, 20000);
        const size_t b2 = lzs_decompress(back, n, got, g2);
        CHECK(b2 == 20000 && memcmp(back, t, 20000) == 0, "keeper: a call after the release");
        lzs_release_thread_cache();
    }
    free(t); free(got); free(back); free(out);
    return NULL;
}
static void release_cache(void)
{
    lzs_release_thread_cache();                          /* (the main thread's own, from the cases before) */
    size_t dev0, host0, live0;
    lzs_shim_outstanding(&dev0, &host0, &live0);
    for (int pass = 0; pass < 3; pass++) {
        /* pass 0: the thread gives its cache back itself; 1: it just exits; 2: a limit of 1 MiB in sum while it lives */
        if (pass == 2) setenv("LZS_KEEP_MAX_MB", "1", 1);
        keeper_t k = { pass == 0, 0, 0, 0 };
        pthread_t th;
        pthread_create(&th, NULL, keeper, &k);
        pthread_join(th, NULL);
        size_t dev1, host1, live1;
        lzs_shim_outstanding(&dev1, &host1, &live1);
        CHECK(dev1 == dev0 && host1 == host0 && live1 == live0, "pass %d: after the thread is gone %zu device bytes, %zu pinned, %zu allocations are outstanding (before: %zu, %zu, %zu)",
              pass, dev1, host1, live1, dev0, host0, live0);
        if (pass == 0) CHECK(k.dev_after_calls > dev0 && k.dev_after_release == dev0 && k.live_after_release == live0,
                             "lzs_release_thread_cache(): %zu device bytes kept after the calls, %zu after the release (before: %zu)", k.dev_after_calls, k.dev_after_release, dev0);
        if (pass == 2) CHECK(k.dev_after_calls <= dev0 + ((size_t)1 << 20) + 4096, "LZS_KEEP_MAX_MB=1: %zu device bytes kept between calls", k.dev_after_calls - dev0);
        if (pass == 1) CHECK(k.dev_after_calls > dev0 + ((size_t)1 << 20), "the default limit keeps the staging of small calls (%zu bytes)", k.dev_after_calls - dev0);
    }
    unsetenv("LZS_KEEP_MAX_MB"); unsetenv("LZS_ROUTE");
}

/* proof that the harness is alive: "device" memory is instrumented heap memory (the test expects the sanitizer to stop this) */
int lzs_hip_malloc(void **p, size_t bytes);
static void canary(void)
{
    void *d = NULL;
    lzs_hip_malloc(&d, 100);
    volatile uint8_t *q = (volatile uint8_t *)d;
    q[100] = 1;                                  /* one byte past a device buffer */
    printf("canary: the write past the buffer went unnoticed\n");
}

int main(int argc, char **argv)
{
    setenv("LZS_DEV_ENV", "1", 1);          /* the switches are read afresh on every call: this program flips them */
    setenv("LZS_ONE_WAVE", "1", 1);         /* decompression stays off the many-wavefront launches (not modelled by the shim) */
    const char *only = argc > 1 ? argv[1] : "";
    char info[256];
    if (lzs_backend_info(info, sizeof info) != 0 || !strstr(info, "cpu shim")) { fprintf(stderr, "not linked with the cpu shim: %s\n", info); return 2; }
    if (!*only || !strcmp(only, "ragged")) ragged_batches();
    if (!*only || !strcmp(only, "streams")) one_shot_streams();
    if (!*only || !strcmp(only, "incremental")) incremental();
    if (!*only || !strcmp(only, "pipeline")) pipeline();
    if (!*only || !strcmp(only, "release")) release_cache();
    if (!strcmp(only, "canary")) canary();
    printf("san_driver: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
