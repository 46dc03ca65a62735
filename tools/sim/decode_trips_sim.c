/* decode_trips_sim.c -- development aid: how many trips of the block decoder's loop (kernels/decompress_blocks.inc) a stream
 * takes under the loop's own rules -- one word of input per trip while 32 bits or fewer are left in a 64-bit buffer, a first
 * token (a run of up to 4 literals, a match, or a length nibble) and a second one (a match that reads nothing the first
 * writes, both within 16 bytes) -- and under variations of them: a third token, a wider feed.  The streams are the seeded
 * classes' blocks compressed by the library's host route (LZS_ROUTE=host: no device needed).  Not part of the library.
 *   usage: decode_trips_sim [class 0..2] [blocks]
 *   build: gcc -O2 tools/sim/decode_trips_sim.c -Iinclude/lzs -Llzs_compression_amd -llzs -llzs_workload -Wl,-rpath,$PWD/lzs_compression_amd -o tools/sim/decode_trips_sim */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lzs.h"
int lzs_workload_fill(uint8_t *, unsigned, uint64_t, uint64_t, size_t, size_t, int);

typedef struct { uint8_t kind; uint16_t off; uint16_t len; uint8_t bits; } tok_t;   /* kind 0 literal, 1 match (first code), 2 nibble, 3 end */

static uint32_t getbits(const uint8_t *s, size_t nbits_total, size_t at, unsigned n)
{
    uint32_t v = 0;
    for (unsigned i = 0; i < n; i++) { const size_t b = at + i; const unsigned bit = b < nbits_total ? (s[b >> 3] >> (7 - (b & 7))) & 1u : 0u; v = (v << 1) | bit; }
    return v;
}
/* the stream as the decoder's steps see it: literals one by one, a match's first code (length 2..8), its nibbles */
static size_t parse(const uint8_t *s, size_t n, tok_t *t, size_t cap)
{
    size_t at = 0, k = 0; const size_t nb = 8 * n;
    while (at < nb && k < cap) {
        if (getbits(s, nb, at, 1) == 0) { t[k++] = (tok_t){0, 0, 1, 9}; at += 9; continue; }
        const unsigned sh = getbits(s, nb, at + 1, 1);
        const unsigned o = sh ? getbits(s, nb, at + 2, 7) : getbits(s, nb, at + 2, 11);
        unsigned used = sh ? 9 : 13;
        if (o == 0) { if (sh) { t[k++] = (tok_t){3, 0, 0, (uint8_t)used}; break; } t[k++] = (tok_t){1, 0, 0, (uint8_t)used}; at += used; continue; }
        const unsigned code = getbits(s, nb, at + used, 4);
        const unsigned len = code < 12 ? 2 + (code >> 2) : code - 7;
        used += code < 12 ? 2 : 4;
        t[k++] = (tok_t){1, (uint16_t)o, (uint16_t)len, (uint8_t)used}; at += used;
        if (len == 8) for (;;) { const unsigned e = getbits(s, nb, at, 4); t[k++] = (tok_t){2, (uint16_t)o, (uint16_t)e, 4}; at += 4; if (e != 15) break; }
    }
    return k;
}
/* trips for one stream.  max_tok: tokens per trip (2 = the product's rule, 3 = a third match); feed_words: words per trip (1 or 2:
 * the 96-bit form's rule, second word only while 64 bits or fewer are left -- here simply a buffer of 96); lit_max: literals per run */
static size_t trips(const tok_t *t, size_t k, size_t total_bits, int max_tok, int feed_words, int lit_max)
{
    size_t i = 0, n = 0, fed = 0; long have = 0;
    const long cap_bits = feed_words == 1 ? 64 : 96;
    while (i < k) {
        for (int w = 0; w < feed_words; w++) if (have <= cap_bits - 32 && fed < total_bits) { const long add = total_bits - fed < 32 ? (long)(total_bits - fed) : 32; have += add; fed += (size_t)add; }
        n++;
        unsigned bytes = 0; int ntok = 0; int complete = 1;
        if (t[i].kind == 2) { have -= 4; bytes = t[i].len; i++; continue; }                       /* a nibble: alone */
        if (t[i].kind == 3) { i++; break; }
        if (t[i].kind == 0) {                                                                       /* a run of literals */
            int run = 0;
            while (i < k && t[i].kind == 0 && run < lit_max && have >= 9) { have -= 9; run++; i++; }
            if (run == 0) { if (fed >= total_bits) break; continue; }                               /* (waits for bits) */
            bytes = (unsigned)run; ntok = 1;
        } else {                                                                                    /* a match */
            if (have < t[i].bits) { if (fed >= total_bits) break; continue; }
            have -= t[i].bits; bytes = t[i].len; complete = t[i].len != 8 && t[i].off != 0; i++; ntok = 1;
        }
        while (ntok < max_tok && complete && i < k && t[i].kind == 1 && t[i].off != 0 && have >= t[i].bits &&
               t[i].off >= bytes + t[i].len && bytes + t[i].len <= 16) {
            have -= t[i].bits; bytes += t[i].len; complete = t[i].len != 8; i++; ntok++;
        }
    }
    return n;
}
int main(int argc, char **argv)
{
    setenv("LZS_ROUTE", "host", 1);
    const unsigned cls = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
    const size_t nb = argc > 2 ? (size_t)atoi(argv[2]) : 32, bl = 65536;
    uint8_t *in = malloc(nb * bl), *out = malloc(LZS_COMPRESSED_MAX(bl));
    tok_t *t = malloc(sizeof(tok_t) * 80000);
    lzs_workload_fill(in, cls, 0x4C5A5331ull, 0, nb, bl, 8);
    double sum[6] = {0}; double toks = 0, bits = 0;
    for (size_t b = 0; b < nb; b++) {
        const size_t n = lzs_compress(out, LZS_COMPRESSED_MAX(bl), in + b * bl, bl);
        if (n == 0) { fprintf(stderr, "lzs_compress failed (the host route needs no device: is liblzs.so built?)\n"); return 1; }
        const size_t k = parse(out, n, t, 80000);
        toks += (double)k; bits += 8.0 * (double)n;
        sum[0] += (double)trips(t, k, 8 * n, 1, 1, 7);
        sum[1] += (double)trips(t, k, 8 * n, 2, 1, 4);
        sum[2] += (double)trips(t, k, 8 * n, 3, 1, 4);
        sum[3] += (double)trips(t, k, 8 * n, 3, 2, 4);
        sum[4] += (double)trips(t, k, 8 * n, 4, 2, 4);
        sum[5] += (double)trips(t, k, 8 * n, 2, 2, 7);
    }
    printf("class %u, %zu blocks of 64 KiB: %.0f steps' tokens and %.0f bits a block (%.1f bits a token)\n", cls, nb, toks / nb, bits / nb, bits / toks);
    const char *name[6] = {"one token a trip, runs of 7 literals, one word a trip", "the product's text form: two tokens, runs of 4, one word a trip",
                           "a third token (a match), one word a trip", "a third token, two words a trip (a 96-bit buffer)",
                           "a fourth token, two words a trip", "two tokens, runs of 7, two words a trip"};
    for (int v = 0; v < 6; v++) printf("  %-66s %8.0f trips a block, %5.2f bytes a trip, %5.1f bits a trip\n", name[v], sum[v] / nb, 65536.0 * nb / sum[v], bits / sum[v]);
    return 0;
}
