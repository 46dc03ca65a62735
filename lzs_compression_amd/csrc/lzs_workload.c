/*
 * lzs_workload.c -- deterministic synthetic block streams for the benchmark
 * configurations of BASELINE.json (text / low-entropy / high-entropy).
 *
 * Integer-only and counter-based: block b of a class depends on
 * (seed, class, b) alone, so any rank / thread / machine regenerates the same
 * bytes, and digests of the reference compressor's output minted in the build
 * container (tests/golden/) stay valid on the GPU box.
 *
 * This is bench/test tooling, not part of the reference's interface
 * (the reference ships no workload generator; SURVEY.md §8d defines these
 * classes).  Built as liblzs_workload.so; the codec library does not need it.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { LZS_WL_TEXT = 0, LZS_WL_LOWENT = 1, LZS_WL_RANDOM = 2 };

/* ---------------------------------------------------------------- PRNG */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct { uint64_t key, ctr; } rng_t;

static rng_t rng_for(uint64_t seed, unsigned cls, uint64_t block)
{
    rng_t r;
    r.key = mix64(seed ^ mix64(((uint64_t)cls << 56) ^ block));
    r.ctr = 0;
    return r;
}
static uint64_t rng_next(rng_t *r) { return mix64(r->key ^ (r->ctr++ * 0xD1342543DE82EF95ull)); }
static uint32_t rng_below(rng_t *r, uint32_t n)          /* uniform in [0,n) */
{
    return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32);
}

/* ------------------------------------------------- class 2: high entropy */
static void gen_random(uint8_t *dst, size_t len, uint64_t seed, uint64_t block)
{
    rng_t r = rng_for(seed, LZS_WL_RANDOM, block);
    size_t i = 0;
    while (i < len) {
        uint64_t v = rng_next(&r);
        for (int k = 0; k < 8 && i < len; k++, i++)
            dst[i] = (uint8_t)(v >> (8 * k));
    }
}

/* -------------------------------------------------- class 1: low entropy */
/* Alternating segments: a run of 0x00 of length U[1,4096], then a 16-byte
 * pattern repeated U[1,256] times. */
static void gen_lowent(uint8_t *dst, size_t len, uint64_t seed, uint64_t block)
{
    rng_t r = rng_for(seed, LZS_WL_LOWENT, block);
    size_t i = 0;
    while (i < len) {
        size_t run = 1 + rng_below(&r, 4096);
        for (; run && i < len; run--)
            dst[i++] = 0;
        uint8_t pat[16];
        uint64_t a = rng_next(&r), b = rng_next(&r);
        for (int k = 0; k < 8; k++) { pat[k] = (uint8_t)(a >> (8 * k)); pat[8 + k] = (uint8_t)(b >> (8 * k)); }
        size_t reps = 1 + rng_below(&r, 256);
        for (size_t t = 0; t < reps * 16 && i < len; t++)
            dst[i++] = pat[t & 15];
    }
}

/* ---------------------------------------------------------- class 0: text */
/* "enwik-style": Zipf-distributed words from a procedurally built vocabulary,
 * sentence punctuation, newlines, and wiki/XML markup.  No corpus is embedded. */
#define VOCAB      5000
#define WORD_MAX   14
typedef struct {
    uint8_t  len[VOCAB];
    char     txt[VOCAB][WORD_MAX];
    uint32_t cdf[VOCAB];          /* cumulative Zipf weights, scaled to 2^32-1 at the end */
} vocab_t;

/* cumulative letter frequencies (per 1000) for e t a o i n s h r d l c u m w f g y p b v k j x q z */
static const char     LETTERS[26] = "etaoinshrdlcumwfgypbvkjxqz";
static const uint16_t LETCUM[26]  = { 127, 218, 300, 375, 445, 512, 575, 636, 696, 739, 779, 807,
                                      835, 859, 883, 905, 925, 945, 964, 979, 989, 997, 998, 999, 1000, 1000 };
static const char     VOWELS[5]   = "eaoiu";

static void vocab_build(vocab_t *v)
{
    uint64_t total = 0;
    static const char *common[] = { "the", "of", "and", "in", "to", "a", "is", "was", "for", "as",
                                    "by", "with", "that", "on", "it", "from", "at", "his", "an", "are" };
    for (unsigned r = 0; r < VOCAB; r++) {
        rng_t g = rng_for(0x766F636162ull, 7, r);
        unsigned len;
        if (r < 20) {
            len = (unsigned)strlen(common[r]);
            memcpy(v->txt[r], common[r], len);
        } else {
            /* rarer words are longer on average */
            unsigned base = 3 + (r > 200) + (r > 1000) + (r > 3000);
            len = base + rng_below(&g, 6);
            if (len > WORD_MAX) len = WORD_MAX;
            for (unsigned k = 0; k < len; k++) {
                if ((k & 1) && rng_below(&g, 10) < 7) {
                    v->txt[r][k] = VOWELS[rng_below(&g, 5)];
                } else {
                    unsigned u = rng_below(&g, 1000), j = 0;
                    while (LETCUM[j] <= u) j++;
                    v->txt[r][k] = LETTERS[j];
                }
            }
        }
        v->len[r] = (uint8_t)len;
        /* Zipf weight ~ 1/(r+1.7): integer arithmetic only */
        total += (uint64_t)1000000000ull / (10ull * r + 17ull);
        v->cdf[r] = 0;
    }
    uint64_t run = 0;
    for (unsigned r = 0; r < VOCAB; r++) {
        run += (uint64_t)1000000000ull / (10ull * r + 17ull);
        /* scale so the last entry is exactly 2^32-1 */
        v->cdf[r] = (uint32_t)((run * 0xFFFFFFFFull) / total);
    }
}

/* Topic locality: about a quarter of the picks repeat one of the last 64 words,
 * as running prose does; the rest are fresh Zipf draws. */
typedef struct { uint16_t recent[64]; unsigned n; } topic_t;

static unsigned zipf_pick(const vocab_t *v, rng_t *r)
{
    uint32_t u = (uint32_t)(rng_next(r) >> 32);
    unsigned lo = 0, hi = VOCAB - 1;
    while (lo < hi) {
        unsigned mid = (lo + hi) >> 1;
        if (v->cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static unsigned vocab_pick(const vocab_t *v, rng_t *r, topic_t *t)
{
    unsigned w;
    if (t->n >= 8 && rng_below(r, 100) < 27)
        w = t->recent[rng_below(r, t->n < 64 ? t->n : 64)];
    else
        w = zipf_pick(v, r);
    t->recent[t->n & 63] = (uint16_t)w;
    t->n++;
    return w;
}

typedef struct { uint8_t *dst; size_t len, at; } wr_t;
static void wr_c(wr_t *w, char c) { if (w->at < w->len) w->dst[w->at] = (uint8_t)c; w->at++; }
static void wr_s(wr_t *w, const char *s) { while (*s) wr_c(w, *s++); }
static void wr_word(wr_t *w, const vocab_t *v, unsigned r, int cap)
{
    for (unsigned k = 0; k < v->len[r]; k++) {
        char c = v->txt[r][k];
        wr_c(w, (cap && k == 0) ? (char)(c - 32) : c);
    }
}
static void wr_num(wr_t *w, uint32_t x)
{
    char buf[12]; int n = 0;
    do { buf[n++] = (char)('0' + x % 10); x /= 10; } while (x);
    while (n) wr_c(w, buf[--n]);
}

static void gen_text(const vocab_t *v, uint8_t *dst, size_t len, uint64_t seed, uint64_t block)
{
    rng_t r = rng_for(seed, LZS_WL_TEXT, block);
    wr_t  w = { dst, len, 0 };
    int   sentence_start = 1;
    topic_t topic;
    topic.n = 0;
    while (w.at < len) {
        uint32_t kind = rng_below(&r, 1000);
        if (kind < 8) {                         /* section heading */
            unsigned depth = 2 + rng_below(&r, 2);
            wr_c(&w, '\n');
            for (unsigned k = 0; k < depth; k++) wr_c(&w, '=');
            wr_c(&w, ' ');
            wr_word(&w, v, vocab_pick(v, &r, &topic), 1);
            if (rng_below(&r, 2)) { wr_c(&w, ' '); wr_word(&w, v, vocab_pick(v, &r, &topic), 0); }
            wr_c(&w, ' ');
            for (unsigned k = 0; k < depth; k++) wr_c(&w, '=');
            wr_c(&w, '\n');
            sentence_start = 1;
        } else if (kind < 11) {                 /* XML page scaffolding */
            wr_s(&w, "\n  </revision>\n</page>\n<page>\n  <title>");
            wr_word(&w, v, vocab_pick(v, &r, &topic), 1);
            wr_s(&w, "</title>\n  <id>");
            wr_num(&w, 1000 + rng_below(&r, 9000000));
            wr_s(&w, "</id>\n  <revision>\n    <timestamp>20");
            wr_num(&w, 10 + rng_below(&r, 16)); wr_c(&w, '-');
            wr_num(&w, 10 + rng_below(&r, 3));  wr_c(&w, '-');
            wr_num(&w, 10 + rng_below(&r, 19));
            wr_s(&w, "T00:00:00Z</timestamp>\n    <text xml:space=\"preserve\">");
            sentence_start = 1;
        } else if (kind < 60) {                 /* wiki link */
            wr_s(&w, "[[");
            /* two draws for one call: the order is spelled out (gcc evaluated the last argument
             * first; the committed digests and lzs_workload_gen.hip follow that order) */
            { int cap = (int)rng_below(&r, 2); unsigned word = vocab_pick(v, &r, &topic); wr_word(&w, v, word, cap); }
            if (rng_below(&r, 3) == 0) { wr_c(&w, ' '); wr_word(&w, v, vocab_pick(v, &r, &topic), 0); }
            if (rng_below(&r, 4) == 0) { wr_c(&w, '|'); wr_word(&w, v, vocab_pick(v, &r, &topic), 0); }
            wr_s(&w, "]] ");
            sentence_start = 0;
        } else if (kind < 75) {                 /* emphasis */
            unsigned q = 2 + rng_below(&r, 2);
            for (unsigned k = 0; k < q; k++) wr_c(&w, '\'');
            wr_word(&w, v, vocab_pick(v, &r, &topic), 0);
            for (unsigned k = 0; k < q; k++) wr_c(&w, '\'');
            wr_c(&w, ' ');
            sentence_start = 0;
        } else if (kind < 95) {                 /* number */
            { uint32_t range = rng_below(&r, 2) ? 2100 : 100000; wr_num(&w, rng_below(&r, range)); }
            wr_c(&w, ' ');
            sentence_start = 0;
        } else {                                /* plain word + separator */
            { int cap = sentence_start || rng_below(&r, 40) == 0; unsigned word = vocab_pick(v, &r, &topic); wr_word(&w, v, word, cap); }
            sentence_start = 0;
            uint32_t sep = rng_below(&r, 100);
            if (sep < 8)       { wr_s(&w, ". "); sentence_start = 1; if (rng_below(&r, 5) == 0) wr_c(&w, '\n'); }
            else if (sep < 15) wr_s(&w, ", ");
            else if (sep < 16) wr_s(&w, "; ");
            else if (sep < 17) wr_s(&w, " (");
            else if (sep < 18) wr_s(&w, ") ");
            else               wr_c(&w, ' ');
        }
    }
}

/* ------------------------------------------------------------ public API */
/* The vocabulary as flat arrays (len[5000], txt[5000*14], cdf[5000]) for the device generator
 * (lzs_workload_gen.hip), which must draw from the very same table. */
int lzs_workload_vocab(uint8_t *len, char *txt, uint32_t *cdf)
{
    vocab_t *v = (vocab_t *)malloc(sizeof(vocab_t));
    if (!v) return -1;
    vocab_build(v);
    memcpy(len, v->len, sizeof(v->len));
    memcpy(txt, v->txt, sizeof(v->txt));
    memcpy(cdf, v->cdf, sizeof(v->cdf));
    free(v);
    return 0;
}

typedef struct {
    const vocab_t *v;
    unsigned cls;
    uint64_t seed, first_block;
    size_t   nblocks, block_len, next;
    uint8_t *dst;
    pthread_mutex_t lock;
} genjob_t;

static void *gen_worker(void *arg)
{
    genjob_t *j = (genjob_t *)arg;
    for (;;) {
        pthread_mutex_lock(&j->lock);
        size_t b = j->next, e = b + 16 < j->nblocks ? b + 16 : j->nblocks;
        j->next = e;
        pthread_mutex_unlock(&j->lock);
        if (b >= j->nblocks) return NULL;
        for (; b < e; b++) {
            uint8_t *d = j->dst + b * j->block_len;
            uint64_t blk = j->first_block + b;
            if (j->cls == LZS_WL_TEXT)        gen_text(j->v, d, j->block_len, j->seed, blk);
            else if (j->cls == LZS_WL_LOWENT) gen_lowent(d, j->block_len, j->seed, blk);
            else                               gen_random(d, j->block_len, j->seed, blk);
        }
    }
}

/* Fill dst[nblocks*block_len] with blocks first_block .. first_block+nblocks-1
 * of class cls (0 text, 1 low-entropy, 2 high-entropy).  Returns 0, or -1 on a
 * bad class / allocation failure. */
int lzs_workload_fill(uint8_t *dst, unsigned cls, uint64_t seed, uint64_t first_block,
                      size_t nblocks, size_t block_len, int nthreads)
{
    if (cls > LZS_WL_RANDOM) return -1;
    vocab_t *v = NULL;
    if (cls == LZS_WL_TEXT) {
        v = (vocab_t *)malloc(sizeof(vocab_t));
        if (!v) return -1;
        vocab_build(v);
    }
    genjob_t j = { v, cls, seed, first_block, nblocks, block_len, 0, dst, PTHREAD_MUTEX_INITIALIZER };
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t tid[256];
    int started = 0;
    for (; started < nthreads; started++)
        if (pthread_create(&tid[started], NULL, gen_worker, &j) != 0) break;
    if (started == 0) gen_worker(&j);
    for (int i = 0; i < started; i++) pthread_join(tid[i], NULL);
    free(v);
    return 0;
}
