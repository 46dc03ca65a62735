/*
 * oracle/lzs_oracle.c -- CPU restatement of the LZS one-shot codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library (liblzs.so, the
 * lzs_compression_amd package) links, loads or calls this file.  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py may use it, and
 * only as the checker, never as the thing measured as the product.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here
 * against (1) the reference's own golden vector and size laws
 * (c/src/test/test-lzs-decompression.c:34-96, c/src/test/test-lzs.c:44-167),
 * (2) fixtures minted from the compiled reference (tests/golden/), and, when
 * oracle/_ref/liblzs_ref.so is present, (3) the real reference live.
 *
 * This is a restatement of *behaviour*, written from the decision rule, not a
 * transcription: flat input indexing (no history ring), an exact 2-gram
 * previous-occurrence chain (no 12-bit hash, no uninitialised tables), a
 * 64-bit bit sink, and a bit-cursor decoder.  Reference lines each routine
 * answers to are cited at the routine.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Wire-format constants: c/src/liblzs/lzs-common.h:38-53, lzs.h:57-60,
 * search cap c/src/liblzs/lzs-compression.c:62. */
enum {
    WINDOW      = 2047,   /* farthest offset a token can name (11 bits)        */
    SEARCH_CAP  = 12,     /* match length at which the search stops improving  */
    SHORT_MAX   = 127,    /* offsets <= this use the 7-bit form                */
    TOKEN_MAX   = 8,      /* longest length the first length code can carry    */
    NIBBLE_MAX  = 15      /* an extension nibble of 15 means "and continue"    */
};

/* ------------------------------------------------------------------ */
/* MSB-first bit sink with the reference's truncation rule:            */
/* bytes past the capacity are dropped, the count stops at capacity    */
/* (c/src/liblzs/lzs-compression.c:304-313, 456-465).                  */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t *dst;
    size_t   cap;
    size_t   total;     /* bytes the untruncated stream would have had so far */
    uint64_t acc;       /* pending bits, right-aligned */
    unsigned pending;   /* how many */
} sink_t;

static void sink_put(sink_t *s, uint32_t value, unsigned width)
{
    s->acc = (s->acc << width) | value;
    s->pending += width;
    while (s->pending >= 8) {
        s->pending -= 8;
        if (s->total < s->cap)
            s->dst[s->total] = (uint8_t)(s->acc >> s->pending);
        s->total++;
    }
}

/* Number of equal leading bytes of in[a..] and in[b..], at most lim.
 * b < a; the two ranges may overlap (c/src/liblzs/lzs-compression.c:178-191). */
static unsigned common_prefix(const uint8_t *in, size_t a, size_t b, unsigned lim)
{
    unsigned k = 0;
    while (k < lim && in[a + k] == in[b + k])
        k++;
    return k;
}

/* Emit the (offset, first length code) of a match
 * (c/src/liblzs/lzs-compression.c:376-409, tables :100-124). */
static void put_match_head(sink_t *s, unsigned off, unsigned len_first)
{
    if (off <= SHORT_MAX)
        sink_put(s, (3u << 7) | off, 9);        /* 1 1 ooooooo        */
    else
        sink_put(s, (2u << 11) | off, 13);      /* 1 0 ooooooooooo    */
    if (len_first <= 4)
        sink_put(s, len_first - 2, 2);          /* 00 01 10           */
    else
        sink_put(s, 0xC + (len_first - 5), 4);  /* 1100 1101 1110 1111*/
}

/* ------------------------------------------------------------------ */
/* The search rule, brute force: the specification itself.             */
/* (c/src/liblzs/lzs-compression.c:322-363, equivalently               */
/*  c/src/liblzs/lzs-compression-simple.c:264-278) */
/* Returns the capped best length (0 or 1 = "no usable match") and the */
/* nearest offset that attains it.                                     */
/* ------------------------------------------------------------------ */
unsigned lzs_oracle_search(const uint8_t *in, size_t n, size_t c, unsigned *best_off)
{
    unsigned lim = (n - c < SEARCH_CAP) ? (unsigned)(n - c) : SEARCH_CAP;
    unsigned best = 0;
    size_t   reach = (c < WINDOW) ? c : WINDOW;
    *best_off = 0;
    if (lim < 2)
        return 0;
    for (size_t off = 1; off <= reach; off++) {
        unsigned l = common_prefix(in, c, c - off, lim);
        if (l > best) {
            best = l;
            *best_off = (unsigned)off;
            if (l == lim)
                break;
        }
    }
    return best;
}

/* Shared token loop.  `finder` selects brute force or the chained finder. */
typedef struct {
    int32_t *head;      /* 65536 entries: latest position of each exact 2-gram */
    int32_t *prev;      /* per position: previous position with the same 2-gram */
} chains_t;

static unsigned chained_search(const chains_t *ch, const uint8_t *in, size_t n, size_t c,
                               unsigned *best_off)
{
    unsigned lim = (n - c < SEARCH_CAP) ? (unsigned)(n - c) : SEARCH_CAP;
    unsigned best = 0;
    *best_off = 0;
    if (lim < 2)
        return 0;
    /* Every position whose first two bytes equal ours, nearest first.  Any
     * offset not on this list has common prefix <= 1 and can never win. */
    int32_t q = ch->head[((unsigned)in[c] << 8) | in[c + 1]];
    while (q >= 0 && c - (size_t)q <= WINDOW) {
        unsigned l = common_prefix(in, c, (size_t)q, lim);
        if (l > best) {
            best = l;
            *best_off = (unsigned)(c - (size_t)q);
            if (l == lim)
                break;
        }
        q = ch->prev[q];
    }
    return best;
}

static size_t compress_core(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int brute)
{
    sink_t   s = { out, cap, 0, 0, 0 };
    chains_t ch = { NULL, NULL };
    size_t   c = 0, inserted = 0;

    if (!brute) {
        ch.head = (int32_t *)malloc(65536 * sizeof(int32_t));
        ch.prev = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
        if (!ch.head || !ch.prev) { free(ch.head); free(ch.prev); return (size_t)-1; }
        memset(ch.head, 0xFF, 65536 * sizeof(int32_t));
    }

    while (c < n) {
        unsigned off, len;
        if (!brute) {
            /* make every position before c searchable */
            for (; inserted < c; inserted++) {
                if (inserted + 1 < n) {
                    unsigned key = ((unsigned)in[inserted] << 8) | in[inserted + 1];
                    ch.prev[inserted] = ch.head[key];
                    ch.head[key] = (int32_t)inserted;
                }
            }
            len = chained_search(&ch, in, n, c, &off);
        } else {
            len = lzs_oracle_search(in, n, c, &off);
        }

        if (len < 2) {                                   /* :365-375 literal */
            sink_put(&s, in[c], 9);
            c += 1;
            continue;
        }
        unsigned first = len < TOKEN_MAX ? len : TOKEN_MAX;   /* :399 */
        put_match_head(&s, off, first);
        c += first;
        if (first == TOKEN_MAX) {                        /* :411-431 extension */
            unsigned e;
            do {
                unsigned lim = (n - c < NIBBLE_MAX) ? (unsigned)(n - c) : NIBBLE_MAX;
                e = common_prefix(in, c, c - off, lim);
                sink_put(&s, e, 4);
                c += e;
            } while (e == NIBBLE_MAX);
        }
    }
    /* End marker 1 1 0000000 then zero bits to the byte boundary (:449-466). */
    sink_put(&s, 0x180, 9);
    if (s.pending)
        sink_put(&s, 0, 8 - s.pending);

    free(ch.head);
    free(ch.prev);
    return s.total < cap ? s.total : cap;
}

/* lzs_compress() semantics (c/src/liblzs/lzs-compression.c:249-467). */
size_t lzs_oracle_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    return compress_core(out, cap, in, n, 0);
}

/* Same contract, brute-force finder: slow, but it *is* the written rule. */
size_t lzs_oracle_compress_brute(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    return compress_core(out, cap, in, n, 1);
}

/* ------------------------------------------------------------------ */
/* lzs_decompress() semantics (c/src/liblzs/lzs-decompression.c:156-412,*/
/* SURVEY.md App. A.4): stop at the first end marker, when the output  */
/* is full (also mid-copy), or when a field needs more bits than the   */
/* input still holds; sources before out[0] read as zero.              */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *src;
    uint64_t       nbits;   /* total bits in the input */
    uint64_t       at;      /* cursor */
} cursor_t;

static uint64_t bits_left(const cursor_t *r) { return r->nbits - r->at; }

/* Next `w` (<= 16) bits, MSB first; bits past the end read as 0. */
static unsigned peek_bits(const cursor_t *r, unsigned w)
{
    unsigned v = 0;
    for (unsigned i = 0; i < w; i++) {
        uint64_t p = r->at + i;
        unsigned bit = (p < r->nbits) ? (r->src[p >> 3] >> (7 - (p & 7))) & 1u : 0u;
        v = (v << 1) | bit;
    }
    return v;
}

static unsigned take_bits(cursor_t *r, unsigned w)
{
    unsigned v = peek_bits(r, w);
    r->at += w;
    return v;
}

/* Copy `len` bytes from `off` back; 1 = output became full. (:346-365, :381-400) */
static int copy_back(uint8_t *out, size_t cap, size_t *count, unsigned off, unsigned len)
{
    for (unsigned i = 0; i < len; i++) {
        out[*count] = (*count >= off) ? out[*count - off] : 0;
        (*count)++;
        if (*count >= cap)
            return 1;
    }
    return 0;
}

size_t lzs_oracle_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    cursor_t r = { in, (uint64_t)n * 8u, 0 };
    size_t   count = 0;
    unsigned off = 0;
    int      extended = 0;

    for (;;) {
        if (bits_left(&r) == 0 || count >= cap)          /* :189, :200 */
            break;
        if (extended) {                                  /* :370-406 */
            if (bits_left(&r) < 4)
                break;
            unsigned e = take_bits(&r, 4);
            if (copy_back(out, cap, &count, off, e))
                break;
            if (e != NIBBLE_MAX)
                extended = 0;
            continue;
        }
        if (take_bits(&r, 1) == 0) {                     /* literal :217-233 */
            if (bits_left(&r) < 8)
                break;
            out[count++] = (uint8_t)take_bits(&r, 8);
            continue;
        }
        if (bits_left(&r) < 1)                           /* :238-241 */
            break;
        if (take_bits(&r, 1)) {                          /* short offset :248-260 */
            if (bits_left(&r) < 7)
                break;
            off = take_bits(&r, 7);
            if (off == 0)
                break;                                   /* end marker */
        } else {                                         /* long offset :272-279 */
            if (bits_left(&r) < 11)
                break;
            off = take_bits(&r, 11);
            if (off == 0)
                continue;                                /* :280 no copy, not an end marker */
        }
        /* length: 00 01 10 -> 2 3 4 ; 11xy -> 5 6 7 8 (:103-120, :325-342) */
        unsigned code = peek_bits(&r, 4);
        unsigned len, width;
        if (code < 0xC) { len = 2 + (code >> 2); width = 2; }
        else            { len = 5 + (code - 0xC); width = 4; }
        if (bits_left(&r) < width)
            break;
        r.at += width;
        if (len == TOKEN_MAX)
            extended = 1;
        if (copy_back(out, cap, &count, off, len))
            break;
    }
    return count;
}

/* ------------------------------------------------------------------ */
/* Token trace, for debugging kernels against the rule: writes up to   */
/* max_tok records {pos, off, len_total} (off 0 = literal) and returns */
/* the token count.  Not part of any reference interface.              */
/* ------------------------------------------------------------------ */
size_t lzs_oracle_trace(const uint8_t *in, size_t n, uint32_t *rec, size_t max_tok)
{
    size_t c = 0, t = 0;
    while (c < n) {
        unsigned off, len = lzs_oracle_search(in, n, c, &off);
        size_t start = c;
        if (len < 2) { off = 0; c += 1; }
        else {
            unsigned first = len < TOKEN_MAX ? len : TOKEN_MAX;
            c += first;
            if (first == TOKEN_MAX) {
                unsigned e;
                do {
                    unsigned lim = (n - c < NIBBLE_MAX) ? (unsigned)(n - c) : NIBBLE_MAX;
                    e = common_prefix(in, c, c - off, lim);
                    c += e;
                } while (e == NIBBLE_MAX);
            }
        }
        if (t < max_tok) {
            rec[3 * t + 0] = (uint32_t)start;
            rec[3 * t + 1] = off;
            rec[3 * t + 2] = (uint32_t)(c - start);
        }
        t++;
    }
    return t;
}
