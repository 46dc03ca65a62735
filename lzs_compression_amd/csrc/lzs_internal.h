/*
 * lzs_internal.h -- what the host-side translation units of liblzs share (not installed).
 *   lzs_host.c         errors, device batches, per-thread staging, host batches, the one-shot calls
 *   lzs_stream.c       one stream (or a small batch) spread over the device: segments + stitch,
 *                      scan / decode / resolve
 *   lzs_incremental.c  the reference's incremental interface on top of both
 */
#ifndef LZS_INTERNAL_H
#define LZS_INTERNAL_H

#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lzs/lzs.h"
#include "lzs/lzs_batch.h"
#include "lzs_hip_shim.h"

#define LZS_HIDDEN __attribute__((visibility("hidden")))

/* errors (lzs_host.c) */
LZS_HIDDEN extern _Thread_local char tls_error[512];
LZS_HIDDEN int fail(int code, const char *fmt, ...);
LZS_HIDDEN int hip_fail(int hip_error, const char *what);
LZS_HIDDEN int require_device(void);

/* The development switches of the environment (tools/README.md), read ONCE per process by lzs_env() -- the
 * entry points ask a struct, not getenv().  LZS_DEV_ENV=1 (set before the first call: the test suites do)
 * makes every lzs_env() read the environment afresh, so a test can flip a switch between two calls. */
typedef struct {
    size_t   keep_max;              /* LZS_KEEP_MAX_MB: staging buffers above it are released after the call; 0: the default (lzs_host.c) */
    int      one_wave, one_workgroup, force_stream;
    uint32_t stream_seg, dec_seg;   /* LZS_STREAM_SEG, LZS_DEC_SEG (0: by size) */
    int      stream_debug, no_marks, no_ones, verify_scan, no_tails, no_chunks;
    int      overlap_off;           /* LZS_HOST_SERIAL: host-buffer batches copy, run and copy back one after the other */
    int      pipe_trace;            /* LZS_PIPE_TRACE (with LZS_STREAM_DEBUG): a line per step of those batches' pipeline */
    int      batch_seg_mb;          /* LZS_BATCH_SEG_MB: host-buffer batches to decompress go by segments up to this extent (0: the default) */
    int      pipe_group, pipe_chunk_mb;   /* LZS_PIPE_GROUP, LZS_PIPE_CHUNK_MB: chunks per launch / chunk size of those batches (0: the defaults) */
    int      pipe_min_mb;           /* LZS_PIPE_MIN_MB: the smallest batch (MiB of the wider side) that takes the overlapped route (0: the default) */
    int      copy_threads;          /* LZS_COPY_THREADS: host threads that fill / empty the pinned pieces of a large batch (default 4) */
    int      staging_fail_mb;       /* LZS_STAGING_FAIL_MB (tests): device reservations above this many MiB fail like a full device */
    int      route;                 /* LZS_ROUTE: 0 by size (the default), 1 "device": every call on the device, 2 "host": the small
                                     * calls' host route for every size it can take, and no device needed (lzs_hostcodec.c) */
} lzs_env_t;
LZS_HIDDEN const lzs_env_t *lzs_env(void);

#define PIPE_STREAMS 12
#define PIPE_EVENTS  24

/* per-thread staging (lzs_host.c): one HIP stream and grow-only device buffers per host thread */
enum { BUF_IN, BUF_OUT, BUF_LEN, BUF_INLEN, BUF_AUX, BUF_KEEP, BUF_MARKS, BUF_COUNT };
typedef struct {
    void  *stream;
    void  *buf[BUF_COUNT];
    size_t cap[BUF_COUNT];
    uint8_t *host_box;              /* host side of the small incremental calls' single copies */
    void  *host_tab;                /* PINNED host memory for the per-segment tables of the stream paths: their */
    size_t host_tab_cap;            /* copies are small and many (five a round), and pageable ones cost ~150 us each */
    /* the overlapped host-buffer batches (lzs_pipeline.c): copy-in, two compute and copy-out streams, events, pinned pieces */
    void  *pipe_stream[PIPE_STREAMS];
    void  *pipe_event[PIPE_EVENTS];
    void  *pin[6];
    size_t pin_cap[6];
    void  *hostcodec;               /* the host route's match-finder tables (lzs_hostcodec.c), made on first use */
} staging_t;
LZS_HIDDEN staging_t *staging_get(void);
LZS_HIDDEN int staging_reserve(staging_t *st, int which, size_t bytes, void **out);   /* 0 or a hipError_t */
LZS_HIDDEN void staging_trim(staging_t *st);
LZS_HIDDEN void *staging_host_tables(staging_t *st, size_t bytes);   /* grow-only, pinned; NULL: out of memory */
/* one of the thread's six pinned pieces (hipHostMallocNonCoherent: only the copy engines and the host touch them -- except
 * PIN_LENGTHS, which a kernel writes and which is coherent), grown to `bytes` if it is smaller; 0 or a HIP error.
 * lzs_pipeline.c: 0, 1 in, 2..4 out, 5 lengths; the one-after-the-other route of host_batch() lays ragged blocks out in
 * 0 (in) and 2, 3 (out). */
#define PIN_LENGTHS 5
LZS_HIDDEN int staging_pin_reserve(staging_t *st, int which, size_t bytes, uint8_t **out);
LZS_HIDDEN double now_ms(void);

/* host-buffer batches with the three stages overlapped (lzs_pipeline.c); LZS_E_* or LZS_OK, *taken = 0: not this batch's route */
typedef int (*launch_fn)(void *, size_t, uint32_t, uint32_t *, const void *, size_t, const uint32_t *, uint32_t, uint32_t, void *);
LZS_HIDDEN int host_batch_pipelined(const char *who, launch_fn launch, uint8_t *out, size_t out_stride, uint32_t cap32, uint32_t *out_len,
                                    const uint8_t *in, size_t in_stride, const uint32_t *in_len_each, size_t in_len, size_t nblocks,
                                    int *taken);

/* ---- the host route of the small calls (lzs_hostcodec.c).  Which call takes it: */
#define LZS_ROUTE_AUTO 0
#define LZS_ROUTE_DEVICE 1
#define LZS_ROUTE_HOST 2
/* the crossovers, measured on the GPU box (profiles/r05/route_crossover.txt, text; one core of an EPYC 9575F against the
 * MI355X): below them one host core is faster than a launch and its wait.  us a call, device / host:
 *   lzs_compress()    4 KiB 99 / 12.5, 16 KiB 118 / 48, 32 KiB 125 / 242          -> 16 KiB of input
 *   lzs_decompress()  16 KiB of output 346 / 14, 64 KiB 494 / 163, 128 KiB 502 / 361, 256 KiB 517 / 736
 *                                                                                   -> 64 KiB of compressed bytes (~115 KiB of text)
 *   lzs_decompress_incremental(), MB/s: 512-byte calls 4.5 / 204, 64 KiB 223 / 339, 256 KiB 773 / 334   -> 64 KiB a call
 *   lzs_compress_incremental(), MB/s: 512-byte calls 65 / 94, 4 KiB 87 / 109, 16 KiB 119 / 112            -> 16 KiB to decide */
#define HOST_COMPRESS_MAX    16384u     /* lzs_compress(): input bytes */
#define HOST_DECOMPRESS_MAX  65536u     /* lzs_decompress(): compressed bytes */
#define INC_DEC_HOST_MAX     65536u     /* lzs_decompress_incremental(): a call's input bytes */
#define INC_ENC_HOST_MAX     16384u     /* lzs_*compress_incremental(): bytes a piece has to decide */
static inline int route_on_host(size_t n, size_t crossover)
{
    const int r = lzs_env()->route;
    return r == LZS_ROUTE_HOST || (r == LZS_ROUTE_AUTO && n <= crossover);
}
LZS_HIDDEN size_t hostcodec_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);      /* SIZE_MAX: out of memory */
LZS_HIDDEN size_t hostcodec_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);
LZS_HIDDEN void hostcodec_free(void *tables);            /* a thread's tables and scratch (staging_destroy) */

/* thresholds of the one-shot calls */
#define STREAM_MIN     6144u        /* shorter inputs are compressed by one workgroup (4 KiB: 0.109 ms alone, 0.127 in segments; 8 KiB: 0.167 / 0.127) */
#define STREAM_DEC_MIN 4096u        /* shorter streams are decompressed by one wavefront */

/* Batches up to this much output are decompressed in segments; larger ones fill the device with a
 * wavefront per block.  Measured (text, 64 KiB blocks, host buffers, ms; segments / wavefront per
 * block), round 2: 4 blocks 0.74 / 8.3, 64 blocks 1.5 / 8.5, 256 blocks 4.2 / 9.4, 512 blocks 11.4 / 13.4,
 * 1024 blocks 25.5 / 24.8; round 4 (the segment decoder at six wavefronts per CU, the copies through pinned
 * pieces; against the overlapped route with the block decoder): 768 blocks 6.2 / 8.6, 1024 blocks 7.3 / 8.9,
 * 1536 blocks 9.9 / 10.2, 2048 blocks 12.4 / 12.4, 4096 blocks 22.2 / 17.8 (profiles/r04/hostbatch_small_routes.txt). */
#define BATCH_SEG_MAX_BLOCKS 4096u
#define BATCH_SEG_MAX_EXTENT ((unsigned long long)64 << 20)

/* one stream on the whole device (lzs_stream.c) */
typedef struct {                    /* a piece of a stream for lzs_compress_incremental(): see stream_compress_piece() */
    const uint8_t *prefix;
    uint32_t prefix_len;
    uint32_t c0, ext_off, bit0;
    uint8_t  first;
    int      last;
    uint32_t stop;          /* != 0: no token starts at or after this position (the data still ends at n: a piece that knows
                             * where the stream ends but must not produce more than its caller has room for) */
    /* results */
    uint32_t c_exit;        /* everything before it is encoded */
    uint32_t ext_exit;      /* != 0: still inside a long match at this offset */
    uint64_t nbits;         /* bits in the output (bit0 included, end marker not) */
} piece_t;
typedef struct {                    /* a piece of a stream for lzs_decompress_incremental(): see stream_decompress() */
    const uint8_t *prefix;
    uint32_t prefix_len;
    uint32_t entry0;
    const uint8_t *hist;
    uint32_t hist_len;
    /* results */
    uint32_t seg, segs_done;    /* segment size used; segments decoded */
    uint32_t next_entry;        /* state word at the start of segment segs_done */
} dec_piece_t;
LZS_HIDDEN size_t stream_compress_piece(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status, piece_t *pc);
LZS_HIDDEN size_t hostcodec_compress_piece(uint8_t *out, size_t cap, const uint8_t *in, size_t n, piece_t *pc);   /* the same on the host; SIZE_MAX: out of memory */
LZS_HIDDEN void hostcodec_decode_resume(lzs_dec_resume_t *st, const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap);   /* lzs_decode_resume_kernel's contract */
LZS_HIDDEN void hostcodec_decode_resume_fields(uint32_t *bitq, uint32_t *qlen, uint32_t *off, uint32_t *rem, uint32_t *extended,
                                               uint8_t *hist, uint32_t *hist_len, const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap,
                                               uint32_t *in_used, uint32_t *out_made, uint32_t *status_out);
LZS_HIDDEN size_t stream_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status);
LZS_HIDDEN size_t stream_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status, int concat,
                                    dec_piece_t *dp);
LZS_HIDDEN int batch_decompress_segments(staging_t *st, void *stream, const char *who, void *d_out, size_t d_out_stride,
                                         uint32_t cap32, uint32_t *out_len, uint32_t *d_len, const void *d_in, size_t d_in_stride,
                                         const uint32_t *in_len_each, uint32_t in_len, size_t nblocks);

#endif
