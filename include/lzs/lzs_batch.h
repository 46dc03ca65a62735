/*
 * lzs/lzs_batch.h -- additive entry points of the MI355X build of liblzs: many
 * independent blocks per call, host or device buffers.
 *
 * The reference has no batch interface: its unit of work is one call of
 * lzs_compress()/lzs_decompress() (c/src/liblzs/lzs.h:218,229) per buffer, and the
 * way to process many buffers is a loop of such calls (c/src/test/test-lzs.c:111-114,
 * c/src/utils/lzs-compress.c:207).  Every function below is defined as exactly that
 * loop -- block b of a batch produces the bytes and the length that one reference call
 * on block b alone would -- executed by one GPU workgroup (compression) or wavefront
 * (decompression) per block.
 *
 * All functions return LZS_OK (0) or a negative LZS_E_* code; lzs_last_error() gives
 * the message of the calling thread's most recent failure.  Plain C types only.
 */
#ifndef LZS_MI355X_LZS_BATCH_H
#define LZS_MI355X_LZS_BATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZS_OK            0
#define LZS_E_NO_DEVICE  (-1)   /* no usable HIP device / runtime                      */
#define LZS_E_HIP        (-2)   /* a HIP call failed (message in lzs_last_error())     */
#define LZS_E_ARG        (-3)   /* invalid argument                                    */
#define LZS_E_NOMEM      (-4)   /* device or host allocation failed                    */

/* Largest single block (bytes) one launch addresses: the limit of the batch and device-pointer calls
 * below.  The plain lzs_compress() / lzs_decompress() take any size_t (longer buffers go in pieces). */
#define LZS_BLOCK_MAX    (3u << 30)

/* Message for the calling thread's most recent failure ("" if none). Never NULL. */
const char *lzs_last_error(void);

/*
 * What the library keeps between calls, and how to get it back.  The reference keeps nothing after return (lzs.h:218,229:
 * the caller's two buffers and some stack).  This build's host-buffer and one-shot calls stage through device memory, a
 * stream and pinned pieces that belong to the CALLING THREAD and stay for its next call (a hipMalloc of gigabytes can
 * take a second): at most LZS_KEEP_MAX_MB MiB of device memory in sum per thread (environment, read once per process;
 * default: 1/32 of the device's memory, at least 640 MiB -- what is above it is freed before the call returns), released
 * when the thread exits -- or now, by this call, from the thread that owns it.  The device-pointer calls
 * (lzs_*_batch_device, lzs_compact_device) allocate nothing and keep nothing.
 */
void lzs_release_thread_cache(void);

/* Human-readable description of the backend ("hip gfx950 ... 256 CUs ...") into buf.
 * Returns LZS_OK, or LZS_E_NO_DEVICE when there is no device (buf then holds the reason). */
int lzs_backend_info(char *buf, size_t cap);

/*
 * Device-pointer batch compression, asynchronous on `hip_stream`.
 *
 *   block b input  : d_in  + b * in_stride ,  length d_in_len ? d_in_len[b] : in_len
 *   block b output : d_out + b * out_stride,  capacity out_cap (same rule as
 *                    a_outBufferSize of lzs_compress(): the stream is cut there)
 *   d_out_len[b]   : bytes written for block b
 *
 * Each block is an independent LZS stream with its own end marker, i.e. the result of
 * lzs_compress(d_out + b*out_stride, out_cap, d_in + b*in_stride, len_b)
 * (reference lzs-compression.c:249-467).  All pointers are device pointers;
 * d_in_len may be NULL; it must not be the array d_out_len (LZS_E_ARG: d_out_len[] is
 * scratch of the launch until a block's length lands there).  `hip_stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Fastest when d_in, in_stride are multiples of 16 and d_out,
 * out_stride multiples of 4; any alignment is accepted.  No allocation, no
 * synchronisation: safe to capture into a hipGraph.  (Once per device and process the
 * library asks the device how it orders same-address LDS exchanges -- a 0.1 ms kernel on a
 * stream of its own, at the first entry into the library on that device; `hip_stream` is not
 * touched by it.  Call lzs_backend_info() first if the very first call is to be captured.)
 */
int lzs_compress_batch_device(void *d_out, size_t out_stride, size_t out_cap, uint32_t *d_out_len,
                              const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                              size_t in_len, size_t nblocks, void *hip_stream);

/*
 * Device-pointer batch decompression, asynchronous on `hip_stream`; block b is
 * lzs_decompress(d_out + b*out_stride, out_cap, d_in + b*in_stride, len_b)
 * (reference lzs-decompression.c:156-412).  Arguments as above.
 */
int lzs_decompress_batch_device(void *d_out, size_t out_stride, size_t out_cap, uint32_t *d_out_len,
                                const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                                size_t in_len, size_t nblocks, void *hip_stream);

/*
 * The same for a batch too small to fill the device with one wavefront per block (a wavefront
 * needs 8-9 ms for a 64 KiB block whatever the batch): every block is cut into segments for many
 * wavefronts (4 blocks 0.7 ms, 64 blocks 1.5 ms).  Buffers in device memory, but
 * the lengths in HOST memory (in_len_each may be NULL: every block in_len bytes; out_len receives
 * the results), and the call is synchronous: it runs on the calling thread's own stream, takes
 * scratch memory from the thread's staging and returns when d_out is complete.  Batches beyond
 * 64 MiB of output (or with blocks under 1 KiB on average) are handed to the one-wavefront-per-
 * block kernel.
 */
int lzs_decompress_batch_device_sync(void *d_out, size_t out_stride, size_t out_cap, uint32_t *out_len,
                                     const void *d_in, size_t in_stride, const uint32_t *in_len_each,
                                     size_t in_len, size_t nblocks);

/*
 * ONE stream from device memory, on the whole device: the result of
 * lzs_compress(d_out, out_cap, d_in, in_len) (reference lzs-compression.c:249-467) for buffers
 * already in HBM.  The stream is cut into segments (0.5 KiB .. 64 KiB), one workgroup each; where each
 * segment's first token starts is agreed in a few rounds and the segments' bits are shifted to
 * their global offsets, so the bytes are those of the one-shot call.
 * The 4-argument lzs_compress() takes the same route for inputs of 6 KiB and more.
 *
 * d_out must be 4-byte aligned and hold LZS_COMPRESSED_MAX(in_len) + 1024 bytes; ALL of that is
 * overwritten (cleared first).  The result is cut at out_cap as lzs_compress() does.  The call
 * synchronises with the device several times (not capturable into a graph) and runs on this thread's
 * own staging stream: d_in must be complete when the call is made (work queued on other streams is
 * not waited for).  Returns the byte count in *out_len.
 */
int lzs_compress_stream_device(void *d_out, size_t out_cap, size_t *out_len,
                               const void *d_in, size_t in_len);

/*
 * The reverse: ONE stream in device memory decompressed by many wavefronts -- the result of
 * lzs_decompress(d_out, out_cap, d_in, in_len) (reference lzs-decompression.c:156-412: stops at
 * the first end marker, at out_cap, or when the bits run out).  The stream is cut into segments
 * (256 bytes .. 8 KiB) that agree on the decoder state at their borders in a few rounds, decode with per-byte
 * origins for copies reaching into another segment's output, and resolve those by pointer jumping
 * (DESIGN.md 3.6).  The 4-argument lzs_decompress() takes the same route from 4 KiB of input on.  Output
 * below 4 GiB; allocates 4 * produced bytes of scratch on this thread's staging; synchronous, on
 * this thread's own stream (d_in must be complete when the call is made).
 */
int lzs_decompress_stream_device(void *d_out, size_t out_cap, size_t *out_len,
                                 const void *d_in, size_t in_len);

/*
 * Gather the variable-length results of a batch into one dense byte string:
 * d_offsets[b] = sum of d_len[0..b) for b = 0..nblocks (nblocks+1 entries, uint64),
 * d_dense[d_offsets[b] .. d_offsets[b+1]) = slot b's first d_len[b] bytes.
 * d_dense must hold sum(d_len) bytes (at most nblocks*max len).  Concatenated
 * independent streams are what the reference's file decompressor consumes
 * (c/src/liblzs/lzs-decompression.c:564-576 realigns after each end marker).
 * Asynchronous on `hip_stream`, no allocation.
 */
int lzs_compact_device(void *d_dense, uint64_t *d_offsets, const void *d_slots, size_t slot_stride,
                       const uint32_t *d_len, size_t nblocks, void *hip_stream);

/*
 * Host-buffer batches: same per-block contract, buffers in host memory.  The call
 * stages through device memory it allocates and frees itself and returns when the
 * results are in `out` / `out_len`.  in_len_each may be NULL (every block in_len bytes).
 * A batch to decompress that is too small to fill the device with a wavefront per block (up to
 * 64 MiB of output) is cut into segments for many wavefronts, block by block:
 * 4 blocks of 64 KiB take 0.7 ms instead of 8.
 */
int lzs_compress_batch(uint8_t *out, size_t out_stride, size_t out_cap, uint32_t *out_len,
                       const uint8_t *in, size_t in_stride, const uint32_t *in_len_each,
                       size_t in_len, size_t nblocks);
int lzs_decompress_batch(uint8_t *out, size_t out_stride, size_t out_cap, uint32_t *out_len,
                         const uint8_t *in, size_t in_stride, const uint32_t *in_len_each,
                         size_t in_len, size_t nblocks);

/*
 * Decode a FILE of concatenated streams as the reference's file decompressor does
 * (c/src/utils/lzs-decompress.c drives the incremental decoder, which after an end marker
 * drops the pad bits to the byte boundary and carries on: lzs-decompression.c:564-576).
 * Same arguments, return value and failure mode as lzs_decompress(); decoding stops at the
 * end of the input or when the output is full.  From 4 KiB on the file is spread over many
 * wavefronts like one long stream (256 MiB of text in 0.45 s with the file I/O);
 * lzs_decompress_batch() with per-block lengths, when they are known, is faster still.
 */
size_t lzs_decompress_concat(uint8_t *out, size_t out_cap, const uint8_t *in, size_t in_len);

#ifdef __cplusplus
}
#endif

#endif /* LZS_MI355X_LZS_BATCH_H */
