/*
 * lzs_hip_shim.h -- the thin extern-"C" seam between the C host library
 * (lzs_host.c, lzs_stream.c, lzs_incremental.c) and the HIP translation unit (lzs_kernels.hip).  Internal: not
 * installed, not part of the public ABI (that is include/lzs/).
 *
 * Every function returns a hipError_t value as int (0 = hipSuccess) unless noted.
 */
#ifndef LZS_HIP_SHIM_H
#define LZS_HIP_SHIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int         lzs_hip_device_count(int *count);
int         lzs_hip_describe(char *buf, size_t cap);          /* "hip gfx950 ..., N CUs, ..." */
const char *lzs_hip_strerror(int hip_error);

int lzs_hip_total_memory(size_t *bytes);             /* the current device's memory */
int lzs_hip_malloc(void **p, size_t bytes);
int lzs_hip_free(void *p);
int lzs_hip_stream_create(void **stream);
int lzs_hip_stream_destroy(void *stream);
int lzs_hip_stream_sync(void *stream);
int lzs_hip_event_create(void **event);              /* (no timing) */
int lzs_hip_event_destroy(void *event);
int lzs_hip_event_record(void *event, void *stream);
int lzs_hip_event_sync(void *event);
int lzs_hip_event_done(void *event);                 /* 1: complete, 0: not yet, < 0: minus a hipError_t */
int lzs_hip_stream_wait_event(void *stream, void *event);
int lzs_hip_host_malloc(void **p, size_t bytes);      /* pinned host memory */
int lzs_hip_host_malloc_staging(void **p, size_t bytes);   /* pinned, hipHostMallocNonCoherent: pieces only the copy engines and the
                                                             * host touch, handed over at events (56 instead of 35-39 GB/s on a quiet
                                                             * device: tools/probes/pipe_copy_probe.hip) */
int lzs_hip_host_free(void *p);
/* Fetch and drop the runtime's sticky "last error" (a failed hipMalloc stays the answer of hipGetLastError() until it is
 * fetched, and the launchers below return hipGetLastError() after their launch). */
void lzs_hip_clear_error(void);
/* `nwords` 32-bit words from device memory to PINNED, COHERENT host memory (lzs_hip_host_malloc) by a kernel on `stream`
 * (not by a copy engine: an engine runs its queue in order, and a small copy waiting for a kernel would hold up the large
 * copies queued behind it); the kernel ends with a system-scope fence */
int lzs_hip_words_to_host(uint32_t *h_dst, const uint32_t *d_src, size_t nwords, void *stream);
int lzs_hip_h2d(void *dst, const void *src, size_t bytes, void *stream);
int lzs_hip_d2h(void *dst, const void *src, size_t bytes, void *stream);
int lzs_hip_d2d(void *dst, const void *src, size_t bytes, void *stream);
int lzs_hip_memset(void *dst, int value, size_t bytes, void *stream);

/* How CHAIN links the positions of a batch on the current device: 0 = by one ordered LDS exchange
 * (the device was asked and applies same-address lanes in lane order), 1 = the order-independent
 * form (the device failed the check, or LZS_CHAIN_FALLBACK=1).  The first call per device runs
 * lzs_lds_order_check_kernel on `stream` and waits for it. */
int lzs_hip_chain_mode(void *stream, int *mode);

/* Kernel launches (asynchronous on `stream`). */
/* The LDS ordering check once more beside the first compress launch (under load): 0 not started yet, 1 in flight,
 * 2 read (a failure switches the device to the order-independent chain build, loudly).  LZS_VERIFY=N in the
 * environment: every N-th compress launch is re-run in the order-independent form and compared on the device. */
int lzs_hip_load_check_state(int dev);
int lzs_hip_launch_compress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                            const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                            uint32_t in_len, uint32_t nblocks, void *stream);
/* lzs_hip_launch_compress() on a device that applies LDS exchanges in lane order launches one variant of the kernel per
 * class of block over the same grid, and the blocks say which is theirs (lzs_classify_blocks_kernel leaves a code in
 * d_out_len[b] until the block's length replaces it).  This runs the classifier alone: d_codes[b] = 0xFFFFFFF0 | 1 the default
 * variant, | 2 few distinct grams (small tables, six workgroups per CU), | 3 nearly all literals (one full step per pass). */
int lzs_hip_classify_blocks(uint32_t *d_codes, const void *d_in, size_t in_stride, const uint32_t *d_in_len, uint32_t in_len,
                            uint32_t nblocks, void *stream);
int lzs_hip_launch_decompress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                              const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                              uint32_t in_len, uint32_t nblocks, void *stream);
int lzs_hip_launch_decompress_concat(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                                     const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                                     uint32_t in_len, uint32_t nblocks, void *stream);
/* One long stream in segments of `seg` bytes (lzs_compress_segments_kernel): segment k, entered
 * at d_entry[k], writes its bits to slot k and reports where its last token ends and its bit
 * count; only segments with d_dirty[k] != 0 run (NULL: all).  lzs_stitch_segments_kernel then ORs
 * the slots into the zeroed, 4-aligned d_out at d_bit_at[k] and appends the end marker.  A segment
 * whose bits did not fit its slot (d_nbits[k] > 8 * slot_stride: a very long match) is run once more
 * with d_out / d_bit_at set and ORs its bits in directly.
 * A piece of a stream that will go on (the incremental interface): no token starts at or after
 * `lim` (= n for a whole stream), d_open[2k..] receives {offset, start} of segment k's last token
 * if that is a match reaching n (it may still grow), and no end marker is written.  A piece that
 * begins inside such a match has lzs_extend_resume_kernel write its length nibbles first
 * (d_result: next position, still open, bits written low/high). */
int lzs_hip_launch_compress_segments(void *d_slots, size_t slot_stride, const void *d_in, uint32_t n,
                                     uint32_t seg, uint32_t nseg, const uint32_t *d_entry,
                                     const uint8_t *d_dirty, uint32_t *d_exit, uint64_t *d_nbits,
                                     void *d_out, const uint64_t *d_bit_at, uint32_t lim, uint32_t *d_open,
                                     void *stream);
int lzs_hip_launch_stitch_segments(void *d_out, const void *d_slots, size_t slot_stride,
                                   const uint64_t *d_bit_at, const uint64_t *d_nbits, uint32_t nseg,
                                   int end_marker, void *stream);
int lzs_hip_launch_extend_resume(void *d_out, uint32_t bit0, const void *d_in, uint32_t n, uint32_t c0,
                                 uint32_t off, int last, uint32_t *d_result, void *stream);
/* One long stream decompressed by many wavefronts (lzs_scan_stream_kernel, lzs_decode_stream_kernel,
 * lzs_resolve_stream_kernel; state words and the scheme are described at the kernels).  With the
 * tables (all NULL for one stream) the segments belong to many streams in one buffer -- a batch of
 * blocks: segment k starts at d_in[d_seg_base[k]], its stream ends at d_in[d_seg_end[k]], copies
 * reaching before d_out[d_out_floor[k]] yield zeros and nothing is written at or past
 * d_out[d_out_limit[k]]. */
#define LZS_SEG_STOP (1u << 30)
unsigned lzs_hip_dec_segment_bytes(void);             /* the largest segment (long streams) */
#define LZS_SCAN_MARK_WORDS 140u     /* per segment in d_marks: what a full walk leaves for repeated ones */
int lzs_hip_launch_scan_stream(const void *d_in, uint32_t n, uint32_t nseg, const uint32_t *d_entry,
                               const uint8_t *d_dirty, uint32_t *d_exit, uint32_t *d_count,
                               uint8_t *d_all_ones /* or NULL */, uint32_t *d_marks, int compare,
                               uint32_t seg, int concat /* go on after end markers */,
                               const uint32_t *d_seg_base, const uint32_t *d_seg_end /* or NULL, NULL */, uint32_t in_extent, void *stream);
int lzs_hip_launch_decode_stream(void *d_out, uint32_t cap, uint32_t *d_origin, uint32_t *d_tainted,
                                 const void *d_in, uint32_t n, uint32_t in_extent /* readable bytes at d_in */,
                                 uint32_t nseg, const uint32_t *d_entry,
                                 const uint32_t *d_out_start, uint32_t seg, int concat,
                                 const uint32_t *d_seg_base, const uint32_t *d_seg_end,
                                 const uint32_t *d_out_floor, const uint32_t *d_out_limit, void *stream);
int lzs_hip_launch_resolve_blocks(void *d_out, uint32_t *d_origin, size_t out_stride, const uint32_t *d_len,
                                  uint32_t nblocks, void *stream);   /* a batch: one workgroup per block, to the end */
int lzs_hip_launch_resolve_stream(void *d_out, uint32_t *d_origin, uint32_t total, uint32_t round,
                                  uint32_t *d_left, int last /* expected to finish: finished groups keep their origins */,
                                  void *stream);
/* The rounds of the same on tails alone: the groups of four bytes in front of the starts
 * d_seg_start[0], [stride], [2 stride] ... (positions in d_out) of `nseg` segments.  d_out must
 * be 4-byte and d_origin 16-byte aligned. */
int lzs_hip_launch_resolve_tails(void *d_out, uint32_t *d_origin, uint32_t total, const uint32_t *d_seg_start,
                                 uint32_t nseg, uint32_t stride, uint32_t round, uint32_t *d_left, void *stream);
/* Without rounds: a workgroup per `per_chunk` consecutive segments takes their tails in order;
 * what is left open afterwards is a copy of a byte in the tail in front of its chunk. */
int lzs_hip_launch_resolve_chunks(void *d_out, uint32_t *d_origin, uint32_t total, const uint32_t *d_seg_start,
                                  uint32_t nseg, uint32_t per_chunk, uint32_t round, void *stream);
/* The incremental entry points (lzs_incremental.c).  Status bits as in the reference's
 * LzsCompressStatus_t / LzsDecompressStatus_t (lzs.h:90-98, 168-176). */
#define LZS_INC_INPUT_STARVED   0x01u
#define LZS_INC_INPUT_FINISHED  0x02u
#define LZS_INC_END_MARKER      0x04u
#define LZS_INC_NO_OUTPUT_SPACE 0x08u
#define LZS_INC_ERROR           0x10u
/* What the incremental decoder carries from call to call (lzs_decode_resume_kernel reads and
 * rewrites it in device memory; the host keeps it in the caller's LzsDecompressParameters_t). */
typedef struct {
    uint32_t bitq, qlen;            /* bits of an unfinished token, left-aligned; how many */
    uint32_t off, rem, extended;    /* copy in progress: offset, bytes left, a nibble follows */
    uint32_t hist_len;              /* bytes of history in hist[] (oldest first), <= 2047 */
    uint32_t in_used, out_made, status, reserved;   /* results of the call */
    uint8_t  hist[2048];
} lzs_dec_resume_t;
int lzs_hip_launch_decode_resume(lzs_dec_resume_t *d_state, const void *d_in, uint32_t n,
                                 void *d_out, uint32_t cap, void *stream);
int lzs_hip_launch_compact(void *d_dense, uint64_t *d_offsets, const void *d_slots,
                           size_t slot_stride, const uint32_t *d_len, uint32_t nblocks,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
