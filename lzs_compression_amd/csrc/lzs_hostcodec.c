/*
 * lzs_hostcodec.c -- the HOST ROUTE of the small calls (SURVEY.md 8(b) "small inputs -> CPU path",
 * 8(f) N3).
 *
 * A launch and its wait are ~10 us on the GPU box and one wavefront walks a token in ~0.6 us, so below a crossover
 * the device loses to one host core by factors: lzs_decompress_incremental() at the reference tools' 512-byte reads
 * ran at 4 MB/s against the reference's 235-816 (c/src/liblzs/lzs-decompression.c:459-743, c/src/utils/
 * lzs-decompress.c:82-121), a 4 KiB lzs_compress() took 0.14-0.4 ms against 0.09 (lzs-compression.c:249-467).  Those
 * calls -- and only those: the one-shot calls below a measured size, the incremental calls on small pieces -- are
 * served here.  bench.py, the batch calls and every device-pointer entry point never come here, and the route is a
 * matter of SIZE, not of failure: without a HIP device the library still fails loudly (require_device() comes first),
 * unless LZS_ROUTE=host asks for this route by name.  LZS_ROUTE=device keeps every call on the device; the parity
 * tests run both routes against the same fixtures.
 *
 * Written from the rule of DESIGN.md section 2, and from the two device kernels whose contracts it shares
 * (lzs_compress_segments_kernel's piece of a stream, lzs_decode_resume_kernel's state block) -- not from the checkers'
 * restatement, which the product never touches (tests/test_abi.py).  What differs from both the reference and that
 * restatement is the finder:
 * positions are chained by a hash of THREE bytes in a 2048-entry ring of distances (a match of 3 and more is on that
 * chain, nearest first), the nearest match of exactly two bytes is one look into a last-occurrence table of exact
 * 2-grams, and neither table is ever cleared: entries carry an epoch base, so what an earlier call left behind lies
 * out of every window of this one.
 */
#include "lzs_internal.h"

#define HC_WINDOW   2047u
#define HC_CAP      12u
#define HC_TOKEN    8u
#define HC_NIBBLE   15u
#define HC_H3_BITS  15u
#define HC_NOLINK   0xFFFFu

typedef struct {
    uint32_t head3[1u << HC_H3_BITS];   /* base + position of the latest position with this 3-byte hash */
    uint32_t head2[65536];              /* base + position of the latest position with exactly these 2 bytes */
    uint16_t link3[2048];               /* per position & 2047: distance to the one before it on its 3-byte chain */
    uint32_t base;                      /* what this call adds to its positions: above everything stored so far */
    uint8_t *scratch; size_t scratch_cap;   /* a piece's [history | input] as one array (grown as needed, freed with the tables) */
} hc_tables_t;

static uint32_t hc_hash3(const uint8_t *p)
{
    const uint32_t t = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    return (t * 0x9E3779B1u) >> (32u - HC_H3_BITS);
}

/* the calling thread's tables, made on first use (they live with the thread's staging: staging_destroy frees them) */
static hc_tables_t *hc_tables(size_t span)
{
    staging_t *st = staging_get();
    if (!st) return NULL;
    hc_tables_t *t = (hc_tables_t *)st->hostcodec;
    if (!t) {
        t = (hc_tables_t *)calloc(1, sizeof *t);
        if (!t) return NULL;
        t->base = HC_WINDOW + 1u;                               /* (zeroed entries lie before every window) */
        st->hostcodec = t;
    }
    if (span > 0x7FFFFFFFu || t->base > 0xFFFFFFFFu - (uint32_t)span - 2u * (HC_WINDOW + 1u)) {
        memset(t->head3, 0, sizeof t->head3);                   /* the epochs are used up (every 4 GiB of input): start over */
        memset(t->head2, 0, sizeof t->head2);
        t->base = HC_WINDOW + 1u;
    }
    return t;
}

/* The scratch copy of a piece stays for the next call while it is small (the 512-byte calls this route exists for);
 * a large one -- LZS_ROUTE=host takes pieces up to 1 GiB -- is given back when its call is done. */
#define HC_SCRATCH_KEEP ((size_t)1 << 20)
static void hc_scratch_trim(hc_tables_t *t)
{
    if (t->scratch_cap > HC_SCRATCH_KEEP) { free(t->scratch); t->scratch = NULL; t->scratch_cap = 0; }
}

LZS_HIDDEN void hostcodec_free(void *tables)
{
    hc_tables_t *t = (hc_tables_t *)tables;
    if (!t) return;
    free(t->scratch);
    free(t);
}

/* ---- MSB-first bit sink; bytes past `cap` are counted, not stored (lzs-compression.c:304-313, 456-465) */
typedef struct { uint8_t *dst; size_t cap, total; uint64_t acc; unsigned pending; } hc_sink_t;

static inline void hc_put(hc_sink_t *s, uint32_t value, unsigned width)
{
    s->acc = (s->acc << width) | value;
    s->pending += width;
    while (s->pending >= 8u) {
        s->pending -= 8u;
        if (s->total < s->cap) s->dst[s->total] = (uint8_t)(s->acc >> s->pending);
        s->total++;
    }
}

static inline unsigned hc_lcp(const uint8_t *a, const uint8_t *b, unsigned lim)
{
    unsigned k = 0;
    while (k < lim && a[k] == b[k]) k++;
    return k;
}

/* insert position p of d[0..n) (both tables as far as its bytes are there) */
static inline void hc_insert(hc_tables_t *t, const uint8_t *d, size_t n, size_t p)
{
    const uint32_t at = t->base + (uint32_t)p;
    if (p + 3u <= n) {
        const uint32_t h = hc_hash3(d + p);
        const uint32_t dist = at - t->head3[h];
        t->link3[p & 2047u] = (uint16_t)(dist <= HC_WINDOW ? dist : HC_NOLINK);
        t->head3[h] = at;
    }
    if (p + 2u <= n) t->head2[(uint32_t)d[p] | ((uint32_t)d[p + 1] << 8)] = at;
}

/* The rule (lzs-compression.c:322-363): the nearest offset in 1..min(c, 2047) that maximises
 * min(common prefix, min(n - c, 12)); returns that length (< 2: a literal) and the offset. */
static inline unsigned hc_search(const hc_tables_t *t, const uint8_t *d, size_t n, size_t c, unsigned *off_out)
{
    const unsigned lim = n - c < HC_CAP ? (unsigned)(n - c) : HC_CAP;
    const uint32_t reach = c < HC_WINDOW ? (uint32_t)c : HC_WINDOW;
    const uint32_t at = t->base + (uint32_t)c;
    unsigned best = 2, off = 0;                                 /* only a match of 3 and more counts on the 3-byte chain */
    if (lim < 2u) return 0;
    if (lim >= 3u) {
        uint32_t dist = at - t->head3[hc_hash3(d + c)];
        while (dist <= reach) {
            const uint8_t *q = d + c - dist;
            if (q[best] == d[c + best]) {                        /* (longer than the best so far: its byte there must agree; best < lim here) */
                const unsigned l = hc_lcp(d + c, q, lim);
                if (l > best) { best = l; off = dist; if (l == lim) break; }
            }
            const uint32_t step = t->link3[(c - dist) & 2047u];
            if (step == HC_NOLINK) break;
            dist += step;
        }
        if (off) { *off_out = off; return best; }
    }
    /* nothing of 3 and more: the nearest position with the same two bytes, if it is inside the window */
    const uint32_t d2 = at - t->head2[(uint32_t)d[c] | ((uint32_t)d[c + 1] << 8)];
    if (d2 <= reach) { *off_out = d2; return 2; }
    return 0;
}

static inline void hc_put_head(hc_sink_t *s, unsigned off, unsigned first)
{
    if (off <= 127u) hc_put(s, (3u << 7) | off, 9);             /* 1 1 ooooooo */
    else             hc_put(s, (2u << 11) | off, 13);           /* 1 0 ooooooooooo */
    if (first <= 4u) hc_put(s, first - 2u, 2);                  /* 00 01 10 */
    else             hc_put(s, 7u + first, 4);                  /* 1100 1101 1110 1111 */
}

/* The token loop over d[0..n) from token start c0: no token starts at or after `lim`; a match may run to n.
 * `open_ok`: a match whose nibbles reach exactly n stays open (its closing nibble is not written; *open_off / *open_start
 * say which).  Returns where the next token would start. */
static size_t hc_tokens(hc_tables_t *t, hc_sink_t *s, const uint8_t *d, size_t n, size_t c0, size_t w0, size_t lim, int open_ok,
                        unsigned *open_off, size_t *open_rest)
{
    size_t c = c0, inserted = w0;
    *open_off = 0; *open_rest = 0;
    while (c < lim) {
        for (; inserted < c; inserted++) hc_insert(t, d, n, inserted);
        unsigned off = 0;
        const unsigned len = hc_search(t, d, n, c, &off);
        if (len < 2u) { hc_put(s, d[c], 9); c++; continue; }   /* 0 bbbbbbbb */
        const unsigned first = len < HC_TOKEN ? len : HC_TOKEN;
        hc_put_head(s, off, first);
        c += first;
        if (first == HC_TOKEN) {                                /* :411-431: nibbles of up to 15 at the same offset */
            for (;;) {
                const unsigned room = n - c < HC_NIBBLE ? (unsigned)(n - c) : HC_NIBBLE;
                const unsigned e = hc_lcp(d + c, d + c - off, room);
                if (open_ok && c + e == n) {                    /* reaches the end of the data so far: may still grow */
                    *open_off = off; *open_rest = e == HC_NIBBLE ? 0 : e;
                    if (e == HC_NIBBLE) { hc_put(s, HC_NIBBLE, 4); c += e; }
                    return c;                                   /* (the bytes of an unfinished nibble wait: c stays before them) */
                }
                hc_put(s, e, 4);
                c += e;
                if (e != HC_NIBBLE) break;
            }
        }
    }
    return c;
}

/* ---------------------------------------------------------------- one-shot: lzs_compress() on the host route */
LZS_HIDDEN size_t hostcodec_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    hc_tables_t *t = hc_tables(n);
    if (!t) return SIZE_MAX;
    hc_sink_t s = { out, cap, 0, 0, 0 };
    unsigned oo; size_t orest;
    (void)hc_tokens(t, &s, in, n, 0, 0, n, 0, &oo, &orest);
    hc_put(&s, 0x180u, 9);                                       /* end marker 1 1 0000000, zero bits to the byte (:449-466) */
    if (s.pending) hc_put(&s, 0, 8u - s.pending);
    t->base += (uint32_t)n + HC_WINDOW + 1u;
    return s.total < cap ? s.total : cap;
}

/* ------------------------------------------- a piece of a stream: the contract of stream_compress_piece(.., pc)
 * data = pc->prefix[0..prefix_len) then in[0..n - prefix_len); encoding starts at pc->c0, inside a long match if
 * pc->ext_off, at bit pc->bit0 of out[0] (whose earlier bits are pc->first).  Results as the device route's:
 * c_exit, ext_exit, nbits (bit0 included, end marker not); returns the bytes of out[] that hold bits. */
LZS_HIDDEN size_t hostcodec_compress_piece(uint8_t *out, size_t cap, const uint8_t *in, size_t n, piece_t *pc)
{
    const size_t pre = pc->prefix_len;
    hc_tables_t *t = hc_tables(n);
    if (!t) return SIZE_MAX;
    /* (not on the caller's stack: the reference's one-shot compressor takes ~12 KiB of it, lzs-compression.c:253-254, and a
     * program sized for that should not meet 24 KiB here) */
    if (t->scratch_cap < n + 16u) {
        uint8_t *grown = (uint8_t *)realloc(t->scratch, n + 4096u);
        if (!grown) return SIZE_MAX;
        t->scratch = grown; t->scratch_cap = n + 4096u;
    }
    uint8_t *d = t->scratch;
    if (pre) memcpy(d, pc->prefix, pre);
    memcpy(d + pre, in, n - pre);
    hc_sink_t s = { out, cap, 0, 0, 0 };
    if (pc->bit0) { s.acc = (uint64_t)pc->first >> (8u - pc->bit0); s.pending = pc->bit0; }
    const int closing = pc->last || pc->stop;                   /* the data is known to end at n */
    size_t c = pc->c0;
    uint64_t nbits = pc->bit0;
    unsigned ext_exit = 0;
    if (pc->ext_off) {
        /* the piece begins inside a long match (state COMPRESS_EXTENDED, :750-776): how far does the data go on
         * repeating; 1111 for every 15 bytes, the closing nibble only if the run ends inside the data (or the data
         * is known to end here) */
        size_t run = 0;
        while (c + run < n && d[c + run] == d[c + run - pc->ext_off]) run++;
        const int open = c + run == n && !closing;
        const size_t full = run / HC_NIBBLE;
        for (size_t i = 0; i < full; i++) hc_put(&s, HC_NIBBLE, 4);
        nbits += 4u * (uint64_t)full;
        if (open) { c += HC_NIBBLE * full; ext_exit = pc->ext_off; }
        else { hc_put(&s, (uint32_t)(run - HC_NIBBLE * full), 4); nbits += 4; c += run; }
    }
    if (!ext_exit) {
        size_t lim = closing ? n : (n > LZS_MAX_LOOK_AHEAD_LEN ? n - LZS_MAX_LOOK_AHEAD_LEN : 0);
        if (!pc->last && pc->stop) lim = pc->stop < n ? pc->stop : n;
        const size_t w0 = c > HC_WINDOW ? c - HC_WINDOW : 0;
        const uint64_t before = 8u * (uint64_t)s.total + s.pending;
        unsigned ooff = 0; size_t orest = 0;
        const size_t c1 = hc_tokens(t, &s, d, n, c, w0, lim, !closing, &ooff, &orest);
        nbits += 8u * (uint64_t)s.total + s.pending - before;
        c = c1;
        if (ooff) ext_exit = ooff;                               /* (c already stands before the bytes of the unfinished nibble) */
        (void)orest;
    }
    pc->c_exit = (uint32_t)c;
    pc->ext_exit = ext_exit;
    pc->nbits = nbits;
    size_t result;
    if (pc->last) {
        hc_put(&s, 0x180u, 9);
        if (s.pending) hc_put(&s, 0, 8u - s.pending);
        result = s.total;
    } else {
        result = s.total;
        if (s.pending) {                                         /* the last byte may be partial: its bits at the top */
            if (result < cap) out[result] = (uint8_t)(s.acc << (8u - s.pending));
            result++;
        }
    }
    t->base += (uint32_t)n + HC_WINDOW + 1u;
    hc_scratch_trim(t);
    return result < cap ? result : cap;
}

/* ---------------------------------------------------------------- decoding
 * One engine for the one-shot call and for the resumable one: bits come MSB first from a 64-bit buffer topped up eight
 * bytes at a time, every field is believed only once all its bits are there (lzs-decompression.c:220-223, 238-251,
 * 272-275, 325-335), a copy replicates byte by byte (:346-365) and reads zero before the first byte there is
 * (:350-357) -- or, in a resumed stream, what the history holds there. */
typedef struct {
    const uint8_t *in; size_t n, ipos;
    uint64_t bits; unsigned have;                               /* left-aligned */
} hc_reader_t;

static inline void hc_feed(hc_reader_t *r)
{
    if (r->n - r->ipos >= 8u) {                                 /* eight bytes at once, as many of them as fit taken */
        uint64_t w;
        memcpy(&w, r->in + r->ipos, 8);
        w = __builtin_bswap64(w);
        const unsigned take = (64u - r->have) >> 3;             /* whole bytes of room: 1 .. 8 (have <= 56) */
        r->bits |= r->have ? (w >> r->have) & (~0ull << (64u - r->have - 8u * take)) : w;
        r->ipos += take;
        r->have += 8u * take;
        return;
    }
    while (r->have <= 56u && r->ipos < r->n) {
        r->bits |= (uint64_t)r->in[r->ipos++] << (56u - r->have);
        r->have += 8u;
    }
}

/* what stops a decoder (the LZS_INC_* bits of lzs_hip_shim.h, which are the reference's status bits) */
typedef struct {
    uint32_t off, rem; int extended;        /* the copy in progress: offset, bytes left, a length nibble follows */
    const uint8_t *hist; uint32_t hist_len; /* bytes in front of out[0] a copy may reach into (resumed streams) */
} hc_copy_t;

/* Decodes until the input gives no whole field any more, the output is full, or (one_shot) the first end marker / (else)
 * any end marker.  Returns the status bits; *count bytes are in out. */
static inline uint32_t hc_decode(hc_reader_t *r, hc_copy_t *c, uint8_t *out, size_t cap, size_t *count_out, int one_shot)
{
    size_t count = 0;
    uint32_t off = c->off, rem = c->rem, status = 0;
    int extended = c->extended;
    for (;;) {
        if (r->have < 32u) hc_feed(r);
        /* no bit left: the incremental decoder stops here whatever it was doing, also with a copy pending (:475-478,
         * :492-496); the one-shot decoder has done its copy by the time it looks (:189 comes after :346-365) */
        if (r->have == 0 && !(one_shot && rem)) { status |= LZS_INC_INPUT_FINISHED | LZS_INC_INPUT_STARVED; break; }
        if (rem) {                                              /* the copy (:346-365, :381-400; :640-704) */
            const size_t room = cap - count;
            if (room == 0) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }
            uint32_t m = rem < room ? rem : (uint32_t)room;
            rem -= m;
            if (count >= off && off >= m && m >= 4u) {          /* no overlap: two moves that meet (nothing past the m bytes is touched) */
                uint8_t *d = out + count;
                const uint8_t *q = d - off;
                if (m >= 8u) { uint64_t a, z; memcpy(&a, q, 8); memcpy(&z, q + m - 8u, 8); memcpy(d, &a, 8); memcpy(d + m - 8u, &z, 8); }
                else         { uint32_t a, z; memcpy(&a, q, 4); memcpy(&z, q + m - 4u, 4); memcpy(d, &a, 4); memcpy(d + m - 4u, &z, 4); }
                count += m;
            } else if (count >= off) {
                for (; m; m--, count++) out[count] = out[count - off];
            } else {
                for (; m; m--, count++) {
                    const size_t back = off - count;            /* before out[0]: the history, or zero (:350-357, :676-683) */
                    out[count] = count >= off ? out[count - off] : (back <= c->hist_len ? c->hist[c->hist_len - back] : 0);
                }
            }
            continue;
        }
        if (one_shot && count >= cap) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }      /* :200 */
        if (extended) {                                         /* :370-406, :706-723 */
            if (r->have < 4u) { status |= LZS_INC_INPUT_STARVED; break; }
            const unsigned e = (unsigned)(r->bits >> 60);
            r->bits <<= 4; r->have -= 4u;
            rem = e;
            if (e != HC_NIBBLE) extended = 0;
        } else if ((r->bits >> 63) == 0) {                      /* literal :217-233, :516-541 */
            if (r->have < 9u) { status |= LZS_INC_INPUT_STARVED; break; }
            if (count >= cap) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }
            out[count++] = (uint8_t)(r->bits >> 55);
            r->bits <<= 9; r->have -= 9u;
        } else {
            const int is_short = (int)((r->bits >> 62) & 1u);
            const unsigned used = is_short ? 9u : 13u;
            if (r->have < used) { status |= LZS_INC_INPUT_STARVED; break; }             /* :238-251, :272-275 */
            const unsigned o = is_short ? (unsigned)(r->bits >> 55) & 0x7Fu : (unsigned)(r->bits >> 51) & 0x7FFu;
            if (o == 0) {
                if (is_short && one_shot) { status |= LZS_INC_END_MARKER; break; }      /* :255-260: nothing after it is looked at */
                r->bits <<= used; r->have -= used;
                if (is_short) {                                 /* end marker: the pad bits go, history stays (:564-576) */
                    const unsigned pad = r->have & 7u;
                    r->bits <<= pad; r->have -= pad;
                    status |= LZS_INC_END_MARKER;
                    break;
                }
                off = 0;                                        /* long offset 0: no copy, not an end marker (:280) */
            } else {
                const unsigned code = (unsigned)((r->bits << used) >> 60);
                const unsigned width = code < 0xCu ? 2u : 4u;   /* :103-120, :325-342 */
                if (r->have < used + width) { status |= LZS_INC_INPUT_STARVED; break; }   /* the token's bits stay queued */
                const unsigned len = code < 0xCu ? 2u + (code >> 2) : code - 7u;
                r->bits <<= used + width; r->have -= used + width;
                off = o; rem = len;
                extended = len == HC_TOKEN;
            }
        }
    }
    c->off = off; c->rem = rem; c->extended = extended;
    *count_out = count;
    return status;
}

/* lzs_decompress() (lzs-decompression.c:156-412): stops at the first end marker, when the output is full (also in the
 * middle of a copy), or when a field has fewer bits left than it needs. */
LZS_HIDDEN size_t hostcodec_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n)
{
    hc_reader_t r = { in, n, 0, 0, 0 };
    hc_copy_t c = { 0, 0, 0, NULL, 0 };
    size_t count = 0;
    (void)hc_decode(&r, &c, out, cap, &count, 1);
    return count;
}

/* lzs_decompress_incremental()'s engine on the host: the contract of lzs_decode_resume_kernel (kernels/compact_resume.inc;
 * reference lzs-decompression.c:459-743), same state, same stop rules at token granularity.  The state is handed over by
 * its fields (the incremental call keeps it in the caller's block and passes that: no copies of the history). */
LZS_HIDDEN void hostcodec_decode_resume_fields(uint32_t *bitq, uint32_t *qlen, uint32_t *off, uint32_t *rem, uint32_t *extended,
                                               uint8_t *hist, uint32_t *hist_len, const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap,
                                               uint32_t *in_used, uint32_t *out_made, uint32_t *status_out)
{
    hc_reader_t r = { in, n, 0, (uint64_t)*bitq << 32, *qlen };
    const unsigned carried = *qlen;
    hc_copy_t c = { *off, *rem, *extended != 0, hist, *hist_len };
    size_t count = 0;
    const uint32_t status = hc_decode(&r, &c, out, cap, &count, 0);
    /* whole bytes of this call's input that were not needed go back to the caller; a starved call keeps the unfinished
     * token's bits (< 17) and takes all the input, as the reference does */
    const uint32_t fed = (uint32_t)r.ipos;
    const uint32_t consumed = carried + 8u * fed - r.have;
    const uint32_t fed_left = consumed >= carried ? r.have : 8u * fed;
    const uint32_t back = (status & LZS_INC_INPUT_STARVED) ? 0u : fed_left >> 3;
    r.have -= 8u * back;
    /* the history: the last 2047 bytes of what was there and what came now */
    if (count >= HC_WINDOW) {
        memcpy(hist, out + count - HC_WINDOW, HC_WINDOW);
        *hist_len = HC_WINDOW;
    } else if (count) {
        const uint32_t keep = *hist_len + (uint32_t)count > HC_WINDOW ? HC_WINDOW - (uint32_t)count : *hist_len;
        memmove(hist, hist + *hist_len - keep, keep);
        memcpy(hist + keep, out, count);
        *hist_len = keep + (uint32_t)count;
    }
    *bitq = r.have ? (uint32_t)(r.bits >> 32) & (~0u << (32u - r.have)) : 0u;      /* (at most 16 bits stay: see above) */
    *qlen = r.have;
    *off = c.off; *rem = c.rem; *extended = c.extended ? 1u : 0u;
    *in_used = fed - back;
    *out_made = (uint32_t)count;
    *status_out = status;
}

LZS_HIDDEN void hostcodec_decode_resume(lzs_dec_resume_t *st, const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap)
{
    hostcodec_decode_resume_fields(&st->bitq, &st->qlen, &st->off, &st->rem, &st->extended, st->hist, &st->hist_len, in, n, out, cap,
                                   &st->in_used, &st->out_made, &st->status);
}
