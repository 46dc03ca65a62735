/*
 * lzs-decompress -- file decompressor on the MI355X path.
 *
 *   lzs-decompress [-x INDEX] IN OUT
 *
 * Without an index the file is decoded the way the reference's tool does it
 * (c/src/utils/lzs-decompress.c:44-124 drives the incremental decoder, which carries on
 * after each end marker): lzs_decompress_concat(), one stream or many back to back.  With the
 * index written by `lzs-compress -x` every block is decoded by its own wavefront in one batch.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lzs.h"
#include "lzs_batch.h"

static uint32_t get32(const unsigned char *b) { return b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24); }

static uint8_t *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    *n = sz > 0 ? (size_t)sz : 0;
    uint8_t *p = (uint8_t *)malloc(*n ? *n : 1);
    if (!p || fread(p, 1, *n, f) != *n) { fprintf(stderr, "lzs-decompress: cannot read %s\n", path); exit(1); }
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    const char *index = NULL;
    int a = 1;
    if (a + 1 < argc && !strcmp(argv[a], "-x")) { index = argv[a + 1]; a += 2; }
    if (argc - a < 2) { fprintf(stderr, "usage: lzs-decompress [-x INDEX] IN OUT\n"); return 2; }
    size_t n, nx = 0;
    uint8_t *in = slurp(argv[a], &n);
    FILE *fo = fopen(argv[a + 1], "wb");
    if (!fo) { perror(argv[a + 1]); return 1; }

    if (index) {
        uint8_t *ix = slurp(index, &nx);
        size_t nblocks = nx / 8, in_stride = 0, out_stride = 0, at = 0;
        uint32_t *len_in = (uint32_t *)malloc((nblocks ? nblocks : 1) * sizeof(uint32_t));
        uint32_t *len_out = (uint32_t *)malloc((nblocks ? nblocks : 1) * sizeof(uint32_t));
        for (size_t b = 0; b < nblocks; b++) {
            len_in[b] = get32(ix + 8 * b);
            uint32_t orig = get32(ix + 8 * b + 4);
            if (len_in[b] > in_stride) in_stride = len_in[b];
            if (orig > out_stride) out_stride = orig;
            at += len_in[b];
        }
        if (at != n) { fprintf(stderr, "lzs-decompress: index does not match the file (%zu vs %zu bytes)\n", at, n); return 1; }
        /* blocks are packed back to back in the file: spread them onto a fixed stride */
        uint8_t *packed = (uint8_t *)calloc(nblocks ? nblocks : 1, in_stride ? in_stride : 1);
        uint8_t *out = (uint8_t *)malloc((nblocks ? nblocks : 1) * (out_stride ? out_stride : 1));
        if (!packed || !out) { fprintf(stderr, "lzs-decompress: out of memory\n"); return 1; }
        at = 0;
        for (size_t b = 0; b < nblocks; b++) { memcpy(packed + b * in_stride, in + at, len_in[b]); at += len_in[b]; }
        if (nblocks && lzs_decompress_batch(out, out_stride, out_stride, len_out, packed, in_stride, len_in, in_stride, nblocks) != LZS_OK) {
            fprintf(stderr, "lzs-decompress: %s\n", lzs_last_error());
            return 1;
        }
        for (size_t b = 0; b < nblocks; b++) {
            if (len_out[b] != get32(ix + 8 * b + 4))
                fprintf(stderr, "lzs-decompress: block %zu decoded to %u bytes, index says %u\n", b, len_out[b], get32(ix + 8 * b + 4));
            fwrite(out + b * out_stride, 1, len_out[b], fo);
        }
    } else {
        /* output size unknown: grow the buffer until the decoder stops short of it */
        size_t cap = n * 4 + 4096;
        for (;;) {
            uint8_t *out = (uint8_t *)malloc(cap);
            if (!out) { fprintf(stderr, "lzs-decompress: out of memory\n"); return 1; }
            size_t got = lzs_decompress_concat(out, cap, in, n);
            if (got == 0 && lzs_last_error()[0]) { fprintf(stderr, "lzs-decompress: %s\n", lzs_last_error()); return 1; }
            if (got < cap) { fwrite(out, 1, got, fo); free(out); break; }
            free(out);
            cap *= 4;
        }
    }
    if (fclose(fo) != 0) { perror(argv[a + 1]); return 1; }
    return 0;
}
