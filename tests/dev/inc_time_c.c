/* tests/dev/inc_time_c.c -- the incremental interface timed from C the way the reference's file tools call it
 * (utils/lzs-compress.c:28-32,91-134, utils/lzs-decompress.c:30-31,78-118: 512-byte reads, 512-byte output buffer),
 * in memory, on one core.  The same source builds against this library (host sources + tests/cpu_shim, LZS_ROUTE=host) and
 * against the compiled reference (oracle/_ref/liblzs_ref.so): tests/dev/inc_time_c.sh.  Dev aid; Python's ctypes costs more
 * per call than these calls take (tests/dev/inc_time.py measures the large pieces).
 * usage: inc_time_c [class=0] [MiB=16] [in_piece=512] [out_piece=512] */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stdbool.h>
#include <time.h>
#include "lzs/lzs.h"

int lzs_workload_fill(uint8_t *dst, unsigned cls, uint64_t seed, uint64_t first_block, size_t nblocks, size_t block_len, int nthreads);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    const unsigned cls = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
    const size_t n = (size_t)(argc > 2 ? atoi(argv[2]) : 16) << 20;
    const size_t in_piece = argc > 3 ? (size_t)atoi(argv[3]) : 512, out_piece = argc > 4 ? (size_t)atoi(argv[4]) : 512;
    uint8_t *data = malloc(n), *comp = malloc(n + n / 8 + 64), *back = malloc(n + 64), *obuf = malloc(out_piece);
    if (lzs_workload_fill(data, cls, 20240229u, 0, n / 65536, 65536, 1)) return 1;
    for (int rep = 0; rep < 3; rep++) {
        LzsCompressParameters_t cp;
        lzs_compress_init(&cp);
        size_t pos = 0, clen = 0, calls = 0;
        bool finish = false;
        double t0 = now();
        cp.inPtr = data; cp.inLength = 0; cp.outPtr = obuf; cp.outLength = out_piece;
        while (!(cp.status & LZS_C_STATUS_END_MARKER)) {
            if (cp.inLength == 0 && !finish) { const size_t k = n - pos < in_piece ? n - pos : in_piece; cp.inPtr = data + pos; cp.inLength = k; pos += k; }
            if (cp.inLength == 0 && (cp.status & LZS_C_STATUS_INPUT_STARVED)) finish = true;
            const size_t got = lzs_compress_incremental(&cp, finish);
            calls++;
            if (got) { memcpy(comp + clen, cp.outPtr - got, got); clen += got; cp.outPtr = obuf; cp.outLength = out_piece; }
        }
        const double tc = now() - t0;
        LzsDecompressParameters_t dp;
        lzs_decompress_init(&dp);
        size_t cpos = 0, dlen = 0, dcalls = 0;
        t0 = now();
        dp.inPtr = comp; dp.inLength = 0; dp.outPtr = obuf; dp.outLength = out_piece;
        for (;;) {
            if (dp.inLength == 0) { const size_t k = clen - cpos < in_piece ? clen - cpos : in_piece; dp.inPtr = comp + cpos; dp.inLength = k; cpos += k; }
            if (dp.inLength == 0 && (dp.status & LZS_D_STATUS_INPUT_STARVED)) break;
            const size_t got = lzs_decompress_incremental(&dp);
            dcalls++;
            if (got) { memcpy(back + dlen, dp.outPtr - got, got); dlen += got; dp.outPtr = obuf; dp.outLength = out_piece; }
        }
        const double td = now() - t0;
        const int ok = dlen == n && !memcmp(back, data, n);
        printf("class %u, %zu MiB, %zu-byte reads, %zu-byte output buffer: compress %.1f MB/s (%zu calls, %.2f us each), ratio %.4f; decompress %.1f MB/s of output (%zu calls, %.2f us each) %s\n",
               cls, n >> 20, in_piece, out_piece, n / tc / 1e6, calls, tc / calls * 1e6, (double)clen / n, n / td / 1e6, dcalls, td / dcalls * 1e6, ok ? "round trip ok" : "ROUND TRIP DIFFERS");
        if (!ok) return 2;
    }
    return 0;
}
