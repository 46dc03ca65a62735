/*
 * lzs-compress -- file compressor on the MI355X batch path.
 *
 *   lzs-compress [-b BLOCK] [-x INDEX] IN OUT
 *
 * The reference's tool (c/src/utils/lzs-compress.c:44-137) writes the whole file as ONE
 * stream: bare bitstream, one end marker, no header.  `-b 0` does exactly that through the
 * 4-argument lzs_compress() and produces the same bytes.  The default, `-b 65536`, cuts the
 * file into independent blocks, compresses them in one GPU batch and writes the streams back
 * to back -- a file the reference's lzs-decompress reads unchanged, because its decoder
 * realigns to the byte boundary after every end marker (lzs-decompression.c:564-576).
 * `-x INDEX` also writes, per block, two little-endian uint32: compressed and original
 * length, which lets lzs-decompress decode the blocks in parallel.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lzs.h"
#include "lzs_batch.h"

static int die(const char *what)
{
    fprintf(stderr, "lzs-compress: %s: %s\n", what, lzs_last_error());
    return 1;
}

static void put32(FILE *f, uint32_t v)
{
    unsigned char b[4] = { (unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24) };
    fwrite(b, 1, 4, f);
}

int main(int argc, char **argv)
{
    size_t block = 65536;
    const char *index = NULL;
    int a = 1;
    while (a < argc && argv[a][0] == '-' && argv[a][1]) {
        if (!strcmp(argv[a], "-b") && a + 1 < argc) { block = strtoull(argv[a + 1], NULL, 0); a += 2; }
        else if (!strcmp(argv[a], "-x") && a + 1 < argc) { index = argv[a + 1]; a += 2; }
        else { fprintf(stderr, "usage: lzs-compress [-b BLOCK] [-x INDEX] IN OUT\n"); return 2; }
    }
    if (argc - a < 2) { fprintf(stderr, "usage: lzs-compress [-b BLOCK] [-x INDEX] IN OUT\n"); return 2; }

    FILE *fi = fopen(argv[a], "rb");
    if (!fi) { perror(argv[a]); return 1; }
    fseek(fi, 0, SEEK_END);
    long fsz = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    size_t n = fsz > 0 ? (size_t)fsz : 0;
    uint8_t *in = (uint8_t *)malloc(n ? n : 1);
    if (!in || fread(in, 1, n, fi) != n) { fprintf(stderr, "lzs-compress: cannot read %s\n", argv[a]); return 1; }
    fclose(fi);

    FILE *fo = fopen(argv[a + 1], "wb");
    if (!fo) { perror(argv[a + 1]); return 1; }
    FILE *fx = index ? fopen(index, "wb") : NULL;
    if (index && !fx) { perror(index); return 1; }

    if (block == 0 || n <= block) {                      /* one stream, like the reference's tool */
        size_t cap = LZS_COMPRESSED_MAX(n);
        uint8_t *out = (uint8_t *)malloc(cap);
        size_t got = out ? lzs_compress(out, cap, in, n) : 0;
        if (got == 0) return die("lzs_compress");
        fwrite(out, 1, got, fo);
        if (fx) { put32(fx, (uint32_t)got); put32(fx, (uint32_t)n); }
        free(out);
    } else {
        size_t nblocks = (n + block - 1) / block;
        size_t cap = LZS_COMPRESSED_MAX(block);
        uint8_t *out = (uint8_t *)malloc(nblocks * cap);
        uint32_t *len_in = (uint32_t *)malloc(nblocks * sizeof(uint32_t));
        uint32_t *len_out = (uint32_t *)malloc(nblocks * sizeof(uint32_t));
        if (!out || !len_in || !len_out) { fprintf(stderr, "lzs-compress: out of memory\n"); return 1; }
        for (size_t b = 0; b < nblocks; b++)
            len_in[b] = (uint32_t)(b + 1 < nblocks ? block : n - b * block);
        if (lzs_compress_batch(out, cap, cap, len_out, in, block, len_in, block, nblocks) != LZS_OK)
            return die("lzs_compress_batch");
        for (size_t b = 0; b < nblocks; b++) {
            fwrite(out + b * cap, 1, len_out[b], fo);
            if (fx) { put32(fx, len_out[b]); put32(fx, len_in[b]); }
        }
        free(out); free(len_in); free(len_out);
    }
    if (fx) fclose(fx);
    if (fclose(fo) != 0) { perror(argv[a + 1]); return 1; }
    free(in);
    return 0;
}
