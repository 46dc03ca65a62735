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
 * HIP kernels on the GPU (there is no CPU codec in this library).  The
 * reference's incremental/streaming entry points (lzs.h:220-232) are outside the
 * scope of this build and are not declared.  Batch and device-pointer entry
 * points, which the reference does not have, are in <lzs/lzs_batch.h>.
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
 * Inputs are limited to 3 GiB per call.
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
 * Failure mode and threading as for lzs_compress().
 */
size_t lzs_decompress(uint8_t * a_pOutData, size_t a_outBufferSize, const uint8_t * a_pInData, size_t a_inLen);

#ifdef __cplusplus
}
#endif

#endif /* LZS_MI355X_LZS_H */
