/*
 * lzs/lzs.h -- drop-in surface of liblzs for the one-shot LZS path
 * (Lempel-Ziv-Stac, ANSI X3.241-1994 / RFC 1974 / RFC 2395), MI355X-native.
 *
 * This header keeps, verbatim in meaning and argument order, the part of the
 * reference's public interface that the hot path lives behind:
 *
 *   reference c/src/liblzs/lzs.h:77    LZS_COMPRESSED_MAX
 *   reference c/src/liblzs/lzs.h:81    LZS_DECOMPRESSED_MAX
 *   reference c/src/liblzs/lzs.h:57-60 LZS_MAX_LOOK_AHEAD_LEN, LZS_MAX_HISTORY_SIZE
 *   reference c/src/liblzs/lzs.h:218   lzs_compress()
 *   reference c/src/liblzs/lzs.h:229   lzs_decompress()
 *
 * A program written against the reference's one-shot calls recompiles against
 * this header and links with -llzs unchanged; the work happens in hand-written
 * HIP kernels on the GPU, and without a GPU every call fails loudly (there is no
 * fallback).  Small calls -- buffers of a few KiB, the incremental calls at the
 * reference tools' 512-byte reads -- are served by the calling thread on a box
 * that has its device, where one host core is faster than a launch and its wait
 * (the device is asked for once per process, at the first call, not before each
 * small call); LZS_ROUTE=device in the environment keeps every call on the GPU.
 * LZS_ROUTE=host is a development and test switch: it sends every size the host
 * route can take (up to 1 GiB) to the calling thread and with that REMOVES the
 * device requirement for those calls -- a CPU codec, not this library's product
 * (DESIGN.md 3.9).  The
 * reference's incremental entry points (lzs.h:220-232) are declared at the end of
 * this header.  Batch and device-pointer entry points, which the reference does
 * not have, are in <lzs/lzs_batch.h>; the partition and the RCCL moves of a job
 * sharded over the GPUs of a node in <lzs/lzs_shard.h>.
 */
#ifndef LZS_MI355X_LZS_H
#define LZS_MI355X_LZS_H

#include <stdint.h>
#include <stdbool.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Wire-format limits (reference lzs.h:57-60). */
#define LZS_MAX_LOOK_AHEAD_LEN      15u
#define LZS_MAX_HISTORY_SIZE        ((1u << 11u) - 1u)

/* Worst-case size of LZS compressed data for X input bytes: 9/8 of the input
 * plus the end marker (reference lzs.h:75-77). */
#define LZS_COMPRESSED_MAX(X)       ((X) + ((X) + 7u) / 8u + 3u)

/* Worst-case size of decompressed data for X compressed bytes, exactly as the
 * reference states it (lzs.h:79-81).  Note that long runs beat it: each 4-bit
 * extension nibble expands to 15 bytes (30x); size output buffers from the known
 * original length where there is one. */
#define LZS_DECOMPRESSED_MAX(X)     ((X) * 16u)

/*
 * Compress a_inLen bytes at a_pInData as ONE LZS stream (history is never reset
 * inside the buffer) into a_pOutData, ending with an end marker.
 *
 * Return value and error convention are the reference's (lzs-compression.c:249-467):
 * the number of bytes written.  If a_outBufferSize is too small the stream is cut
 * at a_outBufferSize (the bytes written are a prefix of the full stream, there is
 * no end marker, nothing past the buffer is touched) and a_outBufferSize is
 * returned.  Empty input yields the two bytes C0 00.  The caller owns both
 * buffers; they must not overlap.  Unlike the reference, a_pInData[a_inLen] is
 * never read.
 *
 * Added failure mode: if no usable HIP device exists (or a HIP call fails) the
 * function writes nothing, returns 0 and lzs_last_error() (lzs_batch.h) describes
 * why; a diagnostic is also printed to stderr.  It never falls back to a CPU codec.
 * Any size_t length, like the reference (lzs.h:218): one launch addresses 3 GiB
 * (LZS_BLOCK_MAX); longer buffers are carried across pieces of <= 1 GiB with the
 * history, the undecided tail, a long match still running and the partial output
 * byte kept from piece to piece -- the bytes are those of the one call.
 *
 * Thread-safe: callable concurrently from any number of host threads.
 */
size_t lzs_compress(uint8_t * a_pOutData, size_t a_outBufferSize, const uint8_t * a_pInData, size_t a_inLen);

/*
 * Decompress one LZS stream.  Semantics are the reference's
 * (lzs-decompression.c:156-412): stops at the first end marker (trailing bytes are
 * ignored), when the output buffer is full (also in the middle of a copy), or when
 * the input runs out in the middle of a token; a match that reaches before the
 * start of the output yields zero bytes.  Returns the number of bytes written.
 * Failure mode and threading as for lzs_compress(); any size_t lengths (streams
 * beyond LZS_BLOCK_MAX and outputs of 4 GiB and more are decoded in pieces).
 */
size_t lzs_decompress(uint8_t * a_pOutData, size_t a_outBufferSize, const uint8_t * a_pInData, size_t a_inLen);

/* ---------------------------------------------------------------------------
 * Incremental (resumable) interface: reference lzs.h:90-134, 168-211, 220-232.
 *
 * Same parameter blocks (the five public members at the same offsets, the structs
 * the same sizes: 14432 and 2096 bytes, so objects compiled against either header
 * are interchangeable), same calling pattern, same status flags.  The private part
 * holds different things: every call is one or a few kernel launches on the device
 * plus the copies, and what must survive between calls (history, look-ahead not yet
 * encoded, bits of an unfinished token, output that did not fit) is kept in the
 * block itself -- nothing is allocated per stream, a block may be copied or dropped
 * at any time.  The streams produced and accepted are the reference's, bit for bit.
 * A call costs ~0.1 ms whatever its size: feed large pieces (MiB) for throughput
 * (pieces of 16 KiB and more are spread over many wavefronts in both directions);
 * 512-byte pieces, the reference tools' habit, work: the compressor collects them in the block
 * up to 10 KiB before it asks the device (67 MB/s), the decompressor runs at ~4.5 MB/s on them
 * (one wavefront walks the tokens of a call: ~0.6 us a token).
 * Any failure (no HIP device, a HIP error, a NULL buffer): status has LZS_x_STATUS_ERROR, the input
 * is dropped and END_MARKER (compressor) / INPUT_STARVED (decompressor) is set next to it, so that a
 * loop that never looks at ERROR ends; lzs_last_error() has the text.
 * ------------------------------------------------------------------------- */
#define LZS_COMPRESS_HISTORY_SIZE   (LZS_MAX_HISTORY_SIZE + LZS_MAX_LOOK_AHEAD_LEN)
#define LZS_DECOMPRESS_HISTORY_SIZE LZS_MAX_HISTORY_SIZE

typedef enum
{
    LZS_C_STATUS_NONE                   = 0x00,
    LZS_C_STATUS_INPUT_STARVED          = 0x01, /* all available input has been read */
    LZS_C_STATUS_INPUT_FINISHED         = 0x02, /* all available input has been read */
    LZS_C_STATUS_END_MARKER             = 0x04, /* the output contains an end marker */
    LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE = 0x08, /* output is waiting for space in the output buffer */
    LZS_C_STATUS_ERROR                  = 0x10  /* no device / a HIP call failed: see lzs_last_error() */
} LzsCompressStatus_t;

typedef struct
{
    /* Set before each call, updated by it (reference lzs.h:107-110). */
    const uint8_t     * inPtr;      /* in: input data; out: first unread byte */
    uint8_t           * outPtr;     /* in: output space; out: one past the last byte written */
    size_t              inLength;   /* in: bytes at inPtr; out: bytes left unread */
    size_t              outLength;  /* in: space at outPtr; out: space left */
    uint8_t             status;     /* LzsCompressStatus_t flags of the last call */
    /* Private.  Sized like the reference's members (lzs.h:123-133). */
    uint8_t             reserved_[14399];
} LzsCompressParameters_t;

typedef enum
{
    LZS_D_STATUS_NONE                   = 0x00,
    LZS_D_STATUS_INPUT_STARVED          = 0x01, /* all input read; an unfinished token may be waiting for more */
    LZS_D_STATUS_INPUT_FINISHED         = 0x02, /* all input read, no bits left over */
    LZS_D_STATUS_END_MARKER             = 0x04, /* stopped after an end marker */
    LZS_D_STATUS_NO_OUTPUT_BUFFER_SPACE = 0x08, /* the next byte has no room in the output buffer */
    LZS_D_STATUS_ERROR                  = 0x10  /* no device / a HIP call failed: see lzs_last_error() */
} LzsDecompressStatus_t;

typedef struct
{
    const uint8_t     * inPtr;
    uint8_t           * outPtr;
    size_t              inLength;
    size_t              outLength;
    uint8_t             status;     /* LzsDecompressStatus_t flags of the last call */
    /* Private.  Sized like the reference's members (lzs.h:197-210). */
    uint8_t             reserved_[2063];
} LzsDecompressParameters_t;

/* Start a stream (reference lzs-compression.c:479-516; the two differ there in how much
 * of the tables they clear, here they are the same). */
void lzs_compress_init_quick(LzsCompressParameters_t * pParams);
void lzs_compress_init_full(LzsCompressParameters_t * pParams);
static inline void lzs_compress_init(LzsCompressParameters_t * pParams) { lzs_compress_init_full(pParams); }

/*
 * Compress the bytes at inPtr into outPtr (reference lzs-compression.c:553-823).  All input is
 * always taken (up to what the waiting output allows, below).  A token is decided once 12 bytes
 * of look-ahead are there (15 inside a long match), so the last 15 bytes always wait for more
 * input; and because a call that reaches the device costs the same whatever its size, input is
 * collected in the block until 3 KiB are waiting.  add_end_marker flushes it all.  With add_end_marker
 * true and inLength 0 everything is flushed and the end marker written (status END_MARKER once
 * it is out); the history stays, so that a following stream may refer back (RFC 1974).
 * The concatenated output of any sequence of calls equals lzs_compress() of the concatenated
 * input.  Output that does not fit outLength (up to 8 KiB of it) waits inside the block and is
 * delivered first by the next calls (status NO_OUTPUT_BUFFER_SPACE while some is waiting).
 * Returns the number of bytes written to outPtr.
 */
size_t lzs_compress_incremental(LzsCompressParameters_t * pParams, bool add_end_marker);

/*
 * The reference's low-memory compressor (lzs-compression-simple.c; lzs.h:136-166, 224-227): the same
 * stream from a parameter block of 2112 bytes.  Here it is the same device code as above behind
 * the smaller block: there is no room in it to collect small pieces or to park output, so every
 * call that can decide a token reaches the device (~0.1 ms), and a call only takes the input whose
 * worst-case output fits outLength and the nine bytes the block can park.  Like the reference's
 * (lzs-compression-simple.c:435-647) it makes progress with any outLength >= 1: with little room a
 * call decides only as many tokens as surely fit (four with one byte of room), status
 * NO_OUTPUT_BUFFER_SPACE when room was the limit.  lzs_simple_compress() is lzs_compress().
 */
typedef struct
{
    const uint8_t     * inPtr;
    uint8_t           * outPtr;
    size_t              inLength;
    size_t              outLength;
    uint8_t             status;     /* LzsCompressStatus_t flags of the last call */
    /* Private.  Sized like the reference's members (lzs.h:155-165). */
    uint8_t             reserved_[2079];
} LzsSimpleCompressParameters_t;

size_t lzs_simple_compress(uint8_t * a_pOutData, size_t a_outBufferSize, const uint8_t * a_pInData, size_t a_inLen);
void lzs_simple_compress_init(LzsSimpleCompressParameters_t * pParams);
size_t lzs_simple_compress_incremental(LzsSimpleCompressParameters_t * pParams, bool add_end_marker);

/* reference lzs-decompression.c:420-428 */
void lzs_decompress_init(LzsDecompressParameters_t * pParams);

/*
 * Decompress from inPtr to outPtr until the input is used up (INPUT_STARVED, with
 * INPUT_FINISHED when no bit of an unfinished token is left), the output is full
 * (NO_OUTPUT_BUFFER_SPACE; also in the middle of a copy, which resumes at the next call) or an
 * end marker has been read (END_MARKER: the next call goes on after it, history kept).
 * reference lzs-decompression.c:459-743.  Returns the number of bytes written to outPtr.
 */
size_t lzs_decompress_incremental(LzsDecompressParameters_t * pParams);

#ifdef __cplusplus
}
#endif

#endif /* LZS_MI355X_LZS_H */
