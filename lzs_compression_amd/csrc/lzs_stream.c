/*
 * lzs_stream.c -- one stream (or a small batch of blocks) spread over the whole device: the host
 * side of the segment compressor (lzs_compress_segments_kernel + lzs_stitch_segments_kernel,
 * DESIGN.md 3.5) and of the many-wavefront decompressor (scan / decode / resolve),
 * including their "piece" forms that the incremental interface drives.
 */
#include "lzs_internal.h"

/* ------------------------------------------------ one long stream on the whole device */
/* lzs_compress() of a buffer too long for one workgroup to be worth waiting for.  The search is a
 * pure function of (input, position), so the stream is cut into segments (stream_seg()), one workgroup
 * each (lzs_compress_segments_kernel), every one writing its bits into a slot of its own.  What a
 * segment cannot know by itself is where its first token starts -- the last token of the segment
 * before usually reaches a few bytes into it -- and at which bit its output begins.  So: (1)
 * every segment is compressed entered at its own start and reports where its last token ends;
 * (2) segments whose predecessor ended elsewhere are compressed again from there, until all
 * entries agree (the greedy parses from two nearby entries merge after a few tokens, so a second
 * round changes almost no exit; a long run simply skips the segments it covers); (3) prefix sum
 * of the bit counts on the host; (4) lzs_stitch_segments_kernel shifts every slot to its bit
 * offset in the zeroed output and appends the end marker.  Same bytes as one workgroup (or the
 * reference) produces. */
LZS_HIDDEN double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define STREAM_SEG_MAX 65536u


/* Segment size for a stream of n bytes: 64 KiB for long streams, much smaller for shorter ones so
 * that they too spread over the device -- a workgroup alone on a CU takes 15 us per KiB, and every
 * segment pays for a 2.2 KB warm-up of its chains (HASH and CHAIN only).  Measured in round 3 (text,
 * host buffers, ms; tests/dev/seg_small_sweep.py, seg_mid_sweep.py):
 *            one workgroup   512 B   1 KiB   2 KiB   4 KiB   16 KiB   64 KiB
 *     8 KiB      0.167       0.127   0.141   0.169   0.227
 *    64 KiB      0.971       0.138   0.151   0.184   0.248
 *   256 KiB      3.749       0.169   0.178   0.212   0.275
 *     1 MiB     15.08        0.357   0.340   0.358   0.417   0.777    1.233
 *     4 MiB                          0.985   0.916   0.925   1.230    2.684
 *    16 MiB                          4.032   3.854   3.785   3.961    5.132
 * (round 2 used 4 KiB up to 2 MiB: 64 KiB in 0.29 ms). */
static uint32_t stream_seg(size_t n)
{
    const uint32_t v = lzs_env()->stream_seg;
    size_t seg = v ? v
               : n <= ((size_t)256 << 10) ? 512u : n <= ((size_t)1 << 20) ? 1024u : n <= ((size_t)4 << 20) ? 2048u
               : n <= ((size_t)32 << 20) ? 4096u : (n / 512u + 4095u) & ~(size_t)4095u;
    if (seg < 256u) seg = 256u;
    if (seg > STREAM_SEG_MAX) seg = STREAM_SEG_MAX;
    return (uint32_t)(seg & ~(size_t)63u);
}

/* in/out on the host (dev == 0: staged through this thread's device buffers) or on the device */
/* A piece of a stream for lzs_compress_incremental(): the data is `prefix` (history and the
 * bytes the call before could not encode yet) followed by `in`; encoding starts at c0, inside a
 * long match if ext_off is set, at bit bit0 of the first output byte (whose earlier bits are in
 * `first`).  Unless `last`, it stops where the tokens are no longer decided by the data so far. */

LZS_HIDDEN size_t stream_compress_piece(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status,
                                    piece_t *pc)
{
    const char *who = pc ? "lzs_compress_incremental" : dev ? "lzs_compress_stream_device" : "lzs_compress";
    /* (a short piece of the incremental interface, whose block collects ~10 KiB, is all latency too:
     * at 512-byte calls 41 MB/s in 4 KiB segments, 62 in 1 KiB, 67 in 512 B) */
    const uint32_t STREAM_SEG = stream_seg(n);
    const uint32_t nseg = n ? (uint32_t)((n + STREAM_SEG - 1) / STREAM_SEG) : 1u;
    const size_t worst = LZS_COMPRESSED_MAX(n - (pc ? pc->c0 : 0)) + 8;
    const int end_marker = !pc || pc->last;
    uint32_t lim = end_marker ? (uint32_t)n : (n > LZS_MAX_LOOK_AHEAD_LEN ? (uint32_t)n - LZS_MAX_LOOK_AHEAD_LEN : 0u);
    if (pc && !pc->last && pc->stop) lim = pc->stop < n ? pc->stop : (uint32_t)n;       /* (the caller vouches that the data ends at n) */
    size_t result = 0;
    int e = 0, rc = LZS_OK;
    void *d_in = NULL, *d_out = NULL, *d_aux = NULL, *d_slots = NULL;
    const size_t slot_stride = (LZS_COMPRESSED_MAX((size_t)STREAM_SEG) + 15u) & ~(size_t)15u;
    uint32_t *entry = NULL, *exitp = NULL, *openi = NULL;
    uint64_t *nbits = NULL, *bitat = NULL;
    uint8_t *dirty = NULL;
    tls_error[0] = 0;
    if (require_device() != LZS_OK) goto failed;
    staging_t *st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
    /* the per-segment tables, in pinned memory (they travel every round) */
    nbits = (uint64_t *)staging_host_tables(st, ((size_t)nseg + 16u) * (2u * 8u + 4u * 4u + 1u));
    if (!nbits) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
    bitat = nbits + nseg; entry = (uint32_t *)(bitat + nseg); exitp = entry + nseg; openi = exitp + nseg;
    dirty = (uint8_t *)(openi + 2 * (size_t)nseg);

#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    /* device arrays in one allocation: bit_at, nbits (8 B each), entry, exit, open info (4 + 4 + 8 B), dirty */
    const size_t aux_bytes = (size_t)nseg * (8 + 8 + 4 + 4 + 8 + 1) + 64;
    if (dev) { d_in = (void *)in; d_out = out; }
    else {
        e = staging_reserve(st, BUF_IN, n + 64, &d_in);
        if (!e) e = staging_reserve(st, BUF_OUT, worst + 1024, &d_out);
    }
    if (!e) e = staging_reserve(st, BUF_AUX, aux_bytes, &d_aux);
    if (!e) e = staging_reserve(st, BUF_KEEP, slot_stride * nseg + 64, &d_slots);
    if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
    uint64_t *d_bitat = (uint64_t *)d_aux;
    uint64_t *d_nbits = d_bitat + nseg;
    uint32_t *d_entry = (uint32_t *)(d_nbits + nseg);
    uint32_t *d_exit = d_entry + nseg;
    uint32_t *d_open = d_exit + nseg;
    uint8_t *d_dirty = (uint8_t *)(d_open + 2 * (size_t)nseg);

    const int debug = lzs_env()->stream_debug;      /* LZS_STREAM_DEBUG: stage times on stderr */
    double t0 = debug ? now_ms() : 0, t1;
    if (!dev) {
        const size_t pre = pc ? pc->prefix_len : 0;
        if (pre) HIP_TRY(lzs_hip_h2d(d_in, pc->prefix, pre, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + pre, in, n - pre, stream), "hipMemcpy H2D");
    }
    /* (a caller's device buffer is cleared only as far as it was promised: LZS_COMPRESSED_MAX(n) + 1024) */
    HIP_TRY(lzs_hip_memset(d_out, 0, dev && worst + 1024 > cap ? cap : worst + 1024, stream), "hipMemset");
    if (debug) { lzs_hip_stream_sync(stream); t1 = now_ms(); fprintf(stderr, "liblzs stream: %zu B, %u segments; H2D + memset %.2f ms\n", n, nseg, t1 - t0); t0 = t1; }
    uint32_t c_first = 0, ext_now = 0;
    uint64_t total = 0;
    if (pc) {
        c_first = pc->c0;
        total = pc->bit0;
        if (pc->bit0) HIP_TRY(lzs_hip_h2d(d_out, &pc->first, 1, stream), "hipMemcpy H2D");
        if (pc->ext_off) {
            /* the piece begins inside a long match: its length nibbles first (d_exit as scratch) */
            uint32_t res[4];
            /* (pc->stop: the data is known to end at n -- the run is closed like in a last piece, only the marker waits) */
            HIP_TRY(lzs_hip_launch_extend_resume(d_out, pc->bit0, d_in, (uint32_t)n, pc->c0, pc->ext_off, pc->last || pc->stop, d_exit, stream), who);
            HIP_TRY(lzs_hip_d2h(res, d_exit, sizeof(res), stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            c_first = res[0];
            ext_now = res[1] ? pc->ext_off : 0;
            total += ((uint64_t)res[3] << 32) | res[2];
        }
    }
    for (uint32_t k = 0; k < nseg; k++) {
        entry[k] = k * STREAM_SEG > c_first ? k * STREAM_SEG : c_first;
        dirty[k] = 1; exitp[k] = entry[k]; nbits[k] = 0; openi[2 * k] = openi[2 * k + 1] = 0;
    }
    /* (a match still open after the nibbles covers all the data there is: no tokens in this piece) */
    for (uint32_t round = 0, ndirty = ext_now ? 0 : nseg; ndirty; round++) {
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_compress_segments(d_slots, slot_stride, d_in, (uint32_t)n, STREAM_SEG, nseg,
                                                 d_entry, d_dirty, d_exit, d_nbits, NULL, NULL, lim, pc ? d_open : NULL, stream), who);
        HIP_TRY(lzs_hip_d2h(exitp, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        /* a segment is entered where the one before stopped (its own start for segment 0) */
        const uint32_t was = ndirty;
        ndirty = 0;
        dirty[0] = 0;
        for (uint32_t k = 1; k < nseg; k++) {
            dirty[k] = exitp[k - 1] != entry[k];
            if (dirty[k]) { entry[k] = exitp[k - 1]; ndirty++; }
        }
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream: round %u compressed %u segments in %.2f ms; %u to redo\n", round, was, t1 - t0, ndirty); t0 = t1; }
    }
    if (!ext_now) {
        HIP_TRY(lzs_hip_d2h(nbits, d_nbits, sizeof(uint64_t) * nseg, stream), "hipMemcpy D2H");
        if (pc) HIP_TRY(lzs_hip_d2h(openi, d_open, sizeof(uint32_t) * 2 * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
    }
    if (pc) {
        pc->c_exit = ext_now ? c_first : exitp[nseg - 1];
        pc->ext_exit = ext_now;
        if (!pc->last && !pc->stop && !ext_now && pc->c_exit >= n && n > c_first) {
            /* The last token is a match that reaches the end of the data so far: it may go on in
             * the next piece.  Its full groups of 15 stand; the closing nibble is taken back and
             * the bytes it covered wait for more data (state COMPRESS_EXTENDED, :750-758). */
            uint32_t k = nseg;
            while (k > 0 && openi[2 * (k - 1)] == 0) k--;
            if (k == 0 || nbits[k - 1] < 4) {
                rc = fail(LZS_E_HIP, "%s: inconsistent state from the device (piece of %zu bytes from %u ends at %u, no open match reported)",
                          who, n, c_first, pc->c_exit);
                goto failed;
            }
            const uint32_t off = openi[2 * (k - 1)], start = openi[2 * (k - 1) + 1];
            const uint32_t rest = ((uint32_t)n - start - 8u) % 15u;
            nbits[k - 1] -= 4;
            HIP_TRY(lzs_hip_h2d(d_nbits, nbits, sizeof(uint64_t) * nseg, stream), "hipMemcpy H2D");
            pc->c_exit = (uint32_t)n - rest;
            pc->ext_exit = off;
        }
    }
    for (uint32_t k = 0; k < nseg; k++) { bitat[k] = total; total += nbits[k]; }
    if (pc) pc->nbits = total;
    HIP_TRY(lzs_hip_h2d(d_bitat, bitat, sizeof(uint64_t) * nseg, stream), "hipMemcpy H2D");
    if (!ext_now)
        HIP_TRY(lzs_hip_launch_stitch_segments(d_out, d_slots, slot_stride, d_bitat, d_nbits, nseg, end_marker, stream), who);
    /* segments whose bits did not fit their slot (a match running on for more than ~120 KB past
     * the segment): once more, ORed straight into place */
    uint32_t nbig = 0;
    for (uint32_t k = 0; k < nseg; k++) { dirty[k] = nbits[k] > 8u * (uint64_t)slot_stride; nbig += dirty[k]; }
    if (nbig) {
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_compress_segments(d_slots, slot_stride, d_in, (uint32_t)n, STREAM_SEG, nseg,
                                                 d_entry, d_dirty, d_exit, d_nbits, d_out, d_bitat, lim, NULL, stream), who);
    }
    if (debug) { lzs_hip_stream_sync(stream); t1 = now_ms(); fprintf(stderr, "liblzs stream: stitch %.2f ms\n", t1 - t0); t0 = t1; }
    result = end_marker ? (size_t)((total + 9 + 7) / 8)        /* end marker, padded to a byte */
                        : (size_t)((total + 7) / 8);            /* a piece: the last byte may be partial */
    if (result > cap) result = cap;                            /* cut at the capacity, prefix unchanged */
    if (!dev) HIP_TRY(lzs_hip_d2h(out, d_out, result, stream), "hipMemcpy D2H");
    HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
#undef HIP_TRY
    goto done;

failed:
    result = 0;
    if (rc == LZS_OK) rc = LZS_E_HIP;
    if (!dev) fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
done:
    { staging_t *s2 = staging_get(); if (s2) staging_trim(s2); }
    if (status) *status = rc;
    return result;
}

LZS_HIDDEN size_t stream_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status)
{
    return stream_compress_piece(out, cap, in, n, dev, status, NULL);
}

int lzs_compress_stream_device(void *d_out, size_t out_cap, size_t *out_len, const void *d_in, size_t in_len)
{
    if (!out_len) return fail(LZS_E_ARG, "lzs_compress_stream_device: out_len is NULL");
    *out_len = 0;
    if (!d_out || (!d_in && in_len)) return fail(LZS_E_ARG, "lzs_compress_stream_device: NULL buffer");
    if (((uintptr_t)d_out & 3u) != 0) return fail(LZS_E_ARG, "lzs_compress_stream_device: d_out is not 4-byte aligned");
    if (in_len == 0 || in_len > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "lzs_compress_stream_device: length must be 1..LZS_BLOCK_MAX");
    int rc = LZS_OK;
    *out_len = stream_compress((uint8_t *)d_out, out_cap, (const uint8_t *)d_in, in_len, 1, &rc);
    return rc;
}

/* Segment size for decompressing a stream of n compressed bytes: 4 KiB for long streams (1 GiB of
 * output, ms: text 53.7 / 51.8 / 53.9 at 2 / 4 / 8 KiB, high-entropy 56.1 / 51.8 / 51.1, low-entropy
 * 8.1 / 9.7 / 11.2 -- smaller segments scan and decode a little faster, and since the copies across
 * their borders are settled by chunks of segments their number costs little), smaller
 * for short ones so that they too are spread over many wavefronts (one wavefront decodes ~7 MB/s).
 * Measured (text, host buffers, ms; output size): 64 KiB 7.7 with one wavefront, 0.67 in 256-byte
 * segments; 256 KiB 30.8 / 1.0; 1 MiB 122.8 / 2.4 (6.3 in 8 KiB segments); 4 MiB 8.2 in 1 KiB
 * segments, 10.1 in 8 KiB ones. */
static uint32_t stream_dec_seg(size_t n)
{
    const uint32_t v = lzs_env()->dec_seg;
    size_t seg = v ? v : (n / 2048u + 255u) & ~(size_t)255u;
    if (!v && seg > 4096u) seg = 4096u;
    const size_t most = lzs_hip_dec_segment_bytes();
    if (seg < 256u) seg = 256u;
    if (seg > most) seg = most;
    return (uint32_t)(seg & ~(size_t)63u);
}

/* lzs_decompress() of one long stream by many wavefronts: see lzs_scan_stream_kernel.  Returns
 * SIZE_MAX if this path does not apply (output of 4 GiB or more) and the caller should decode
 * with one wavefront. */
/* A piece of a stream for lzs_decompress_incremental(): the input is `prefix` (the bytes that hold
 * the bits left over from the call before) followed by `in`; the walk starts in state `entry0`
 * (bit offset into the first byte, extension running, offset: the kernels' state word); copies may
 * reach back into `hist`, the last bytes produced before.  Only whole segments are decoded, and
 * only those before the first one in which the stream stops (end marker, unfinished token) or
 * which would overflow the output: the rest is the one wavefront's (lzs_decode_resume_kernel). */

LZS_HIDDEN size_t stream_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n, int dev, int *status, int concat,
                                dec_piece_t *dp)
{
    const char *who = dp ? "lzs_decompress_incremental" : dev ? "lzs_decompress_stream_device" : concat ? "lzs_decompress_concat" : "lzs_decompress";
    const uint32_t seg = stream_dec_seg(n);
    const uint32_t nseg = (uint32_t)((n + seg - 1) / seg);
    size_t result = 0;
    int e = 0, rc = LZS_OK;
    void *d_in = NULL, *d_out = NULL, *d_aux = NULL, *d_origin = NULL, *d_marks = NULL;
    uint32_t *entry = NULL, *exits = NULL, *count = NULL, *start = NULL;
    uint8_t *dirty = NULL, *ones = NULL;
    uint32_t *seen = NULL;
    tls_error[0] = 0;
    if (require_device() != LZS_OK) goto failed;
    staging_t *st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
    /* the tables that travel every round, in pinned memory */
    seen = (uint32_t *)staging_host_tables(st, ((size_t)nseg + 16u) * (5u * 4u + 2u));
    if (!seen) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
    entry = seen + nseg; exits = entry + nseg; count = exits + nseg; start = count + nseg;
    dirty = (uint8_t *)(start + nseg); ones = dirty + nseg;

#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    const size_t aux_bytes = (size_t)nseg * (4 + 4 + 4 + 4 + 1 + 1) + 128;
    if (dev) d_in = (void *)in; else e = staging_reserve(st, BUF_IN, n + 64, &d_in);
    if (!e) e = staging_reserve(st, BUF_AUX, aux_bytes, &d_aux);
    if (!e) e = staging_reserve(st, BUF_MARKS, (size_t)nseg * LZS_SCAN_MARK_WORDS * 4u, &d_marks);
    if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
    uint32_t *d_entry = (uint32_t *)d_aux;
    uint32_t *d_exit = d_entry + nseg;
    uint32_t *d_count = d_exit + nseg;
    uint32_t *d_start = d_count + nseg;
    uint32_t *d_counters = d_start + nseg;                     /* [0] bytes with an origin, [1] left open */
    uint8_t *d_dirty = (uint8_t *)(d_counters + 2);
    uint8_t *d_ones = d_dirty + nseg;

    const int debug = lzs_env()->stream_debug;
    double t0 = debug ? now_ms() : 0, t1;
    if (!dev) {
        const size_t pre = dp ? dp->prefix_len : 0;
        if (pre) HIP_TRY(lzs_hip_h2d(d_in, dp->prefix, pre, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d((uint8_t *)d_in + pre, in, n - pre, stream), "hipMemcpy H2D");
    }
    /* SCAN rounds: every segment entered at its first bit in the normal state, then corrected */
    for (uint32_t k = 0; k < nseg; k++) { entry[k] = 0; dirty[k] = 1; seen[k] = 0xFFFFFFFFu; }
    if (dp) entry[0] = dp->entry0;
    for (uint32_t round = 0, ndirty = nseg; ndirty; round++) {
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_scan_stream(d_in, (uint32_t)n, nseg, d_entry, d_dirty, d_exit, d_count,
                                           round == 0 ? d_ones : NULL, (uint32_t *)d_marks, round != 0 && !lzs_env()->no_marks, seg, concat, NULL, NULL, (uint32_t)n, stream), who);
        /* (exits and count lie one behind the other on both sides: one copy) */
        HIP_TRY(lzs_hip_d2h(exits, d_exit, 2 * sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        if (round == 0) HIP_TRY(lzs_hip_d2h(ones, d_ones, nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        const uint32_t was = ndirty;
        ndirty = 0;
        if (dirty[0]) seen[0] = entry[0];                       /* exits[k], count[k] belong to this entry */
        dirty[0] = 0;
        int ended = 0;
        int settled = 1;            /* every segment before k has been walked from its final entry */
        int host_made = 0;          /* exits / counts worked out here, not on the device */
        for (uint32_t k = 1; k < nseg; k++) {
            if (dirty[k]) seen[k] = entry[k];
            uint32_t want = exits[k - 1];
            if (want & LZS_SEG_STOP) {
                /* end marker or end of input before k -- believed only from a settled walk: one that
                 * was entered at a guessed bit reads end markers into the data now and then */
                if (settled) ended = 1; else want = entry[k];
            }
            if (ended) want = LZS_SEG_STOP;
            if (want != entry[k]) settled = 0;
            entry[k] = want;
            dirty[k] = 0;
            /* a segment behind the (current) end of the stream keeps what it reported for its last
             * entry: the end may turn out to be a misread of a walk that had not fallen in step */
            if (ended || want == seen[k]) continue;
            if (((want >> 8) & 1u) && (ones[k] == 2 || (ones[k] && (want & 3u) == 0)) && !lzs_env()->no_ones) {
                /* all 0xFF inside a running extension: nothing but nibbles of 15, one every 4 bits
                 * from the cursor on (which is up to 20 bits in if the match token itself straddles
                 * the border) for as long as they start inside the segment -- provided the last of
                 * them is all ones too, which reaches up to 3 bits into the next segment unless the
                 * cursor is a multiple of 4.  No need to walk it then. */
                const uint32_t r = want & 0xFFu;
                const uint32_t nibbles = (seg * 8u - r + 3u) / 4u;
                exits[k] = (want & ~0xFFu) | (r + 4u * nibbles - seg * 8u);
                count[k] = 15u * nibbles;
                seen[k] = want;
                host_made = 1;
                continue;
            }
            dirty[k] = 1;
            ndirty++;
        }
        /* what the host worked out itself must survive the next round's copy back */
        if (ndirty && host_made)
            HIP_TRY(lzs_hip_h2d(d_exit, exits, 2 * sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream decode: round %u scanned %u of %u segments in %.2f ms; %u to redo\n", round, was, nseg, t1 - t0, ndirty); t0 = t1; }
    }
    if (lzs_env()->verify_scan) {
        /* development check: every segment walked in full from its final entry must report what
         * the rounds arrived at (merged walks and the all-0xFF shortcut included) */
        uint32_t *ex2 = (uint32_t *)malloc(sizeof(uint32_t) * nseg), *cn2 = (uint32_t *)malloc(sizeof(uint32_t) * nseg);
        memset(dirty, 1, nseg);
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_scan_stream(d_in, (uint32_t)n, nseg, d_entry, d_dirty, d_exit, d_count, NULL, NULL, 0, seg, concat, NULL, NULL, (uint32_t)n, stream), who);
        HIP_TRY(lzs_hip_d2h(ex2, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_d2h(cn2, d_count, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        for (uint32_t k = 0; k < nseg; k++) {
            if (entry[k] & LZS_SEG_STOP) break;
            if (ex2[k] != exits[k] || cn2[k] != count[k])
                fprintf(stderr, "liblzs verify: segment %u entry %08x: rounds say exit %08x count %u, a full walk says %08x %u (all-ones %u)\n",
                        k, entry[k], exits[k], count[k], ex2[k], cn2[k], ones[k]);
        }
        free(ex2); free(cn2);
    }
    uint64_t total = 0;
    uint32_t ndec = nseg;                                      /* segments to decode */
    const uint32_t before = dp ? dp->hist_len : 0;             /* bytes in front of out[0] that copies may reach */
    for (uint32_t k = 0; k < nseg; k++) {
        if (dp && ((exits[k] & LZS_SEG_STOP) || (entry[k] & LZS_SEG_STOP) || total + count[k] > cap)) { ndec = k; break; }
        start[k] = (uint32_t)total + before;
        if (!(entry[k] & LZS_SEG_STOP)) total += count[k];
        if (total >= 0xFFFFFF00ull - 0x100000ull) break;
    }
    if (total >= 0xFFFFFF00ull - 0x100000ull) { result = SIZE_MAX; goto done; }   /* positions are 32-bit here */
    if (dp) { dp->seg = seg; dp->segs_done = ndec; dp->next_entry = ndec < nseg ? entry[ndec] : exits[nseg - 1]; }
    const uint32_t produce = (uint32_t)(total < cap ? total : cap);
    if (produce) {
        if (dev) d_out = out; else e = staging_reserve(st, BUF_OUT, (size_t)before + produce + 64, &d_out);
        if (!e) e = staging_reserve(st, BUF_KEEP, 4 * ((size_t)before + produce) + 64, &d_origin);
        if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
        if (before) {                                          /* the history: final bytes (origin "clean" = all ones) */
            HIP_TRY(lzs_hip_h2d(d_out, dp->hist, before, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_memset(d_origin, 0xFF, 4 * (size_t)before, stream), "hipMemset");
        }
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_start, start, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_memset(d_counters, 0, 8, stream), "hipMemset");
        HIP_TRY(lzs_hip_launch_decode_stream(d_out, before + produce, (uint32_t *)d_origin, d_counters, d_in, (uint32_t)n, (uint32_t)n,
                                             ndec, d_entry, d_start, seg, concat, NULL, NULL, NULL, NULL, stream), who);
        uint32_t open[2] = {0, 0};
        HIP_TRY(lzs_hip_d2h(open, d_counters, 8, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream decode: %u bytes decoded in %.2f ms, %u with an origin in another segment\n", produce, t1 - t0, open[0]); t0 = t1; }
        uint32_t left = open[0];
        /* All origins lie in the 2047 bytes in front of a segment start (the tails).  So: (1) a
         * workgroup per chunk of consecutive segments settles their tails in order, up to copies
         * of bytes in front of the chunk; (2) rounds of pointer jumping on the tails in front of
         * the chunks alone; (3) one pass over everything.  (3) alone, repeated, does the job too:
         * (1) and (2) are shortcuts, not conditions. */
        const int aligned = ((uintptr_t)d_out & 3u) == 0 && ((uintptr_t)d_origin & 15u) == 0;
        int tails = left && ndec > 1 && aligned && !lzs_env()->no_tails;
        const int had_tails = tails;
        uint32_t stride = 1, round = 1;
        if (tails && ndec >= 64 && !lzs_env()->no_chunks) {
            stride = (ndec + 2047u) / 2048u;                    /* <= 2048 workgroups: all resident at once */
            if (stride < 16u) stride = 16u;
            HIP_TRY(lzs_hip_launch_resolve_chunks(d_out, (uint32_t *)d_origin, before + produce, d_start, ndec, stride, round++, stream), who);
            if (debug) { HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize"); t1 = now_ms(); fprintf(stderr, "liblzs stream decode: tails by chunks of %u segments in %.2f ms\n", stride, t1 - t0); t0 = t1; }
        }
        for (; left && round < 250; round++) {
            HIP_TRY(lzs_hip_memset(d_counters + 1, 0, 4, stream), "hipMemset");
            if (tails) {
                /* three rounds to a look at the counter: a round on the tails is 30 - 130 us, the look costs as much */
                HIP_TRY(lzs_hip_launch_resolve_tails(d_out, (uint32_t *)d_origin, before + produce, d_start, ndec, stride, round, d_counters + 1, stream), who);
                for (int more = 0; more < 2; more++) {
                    round++;
                    HIP_TRY(lzs_hip_memset(d_counters + 1, 0, 4, stream), "hipMemset");
                    HIP_TRY(lzs_hip_launch_resolve_tails(d_out, (uint32_t *)d_origin, before + produce, d_start, ndec, stride, round, d_counters + 1, stream), who);
                }
            } else
                HIP_TRY(lzs_hip_launch_resolve_stream(d_out, (uint32_t *)d_origin, before + produce, round, d_counters + 1, had_tails, stream), who);
            HIP_TRY(lzs_hip_d2h(&left, d_counters + 1, 4, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs stream decode: resolve round %u (%s, stride %u) in %.2f ms, %u left\n", round, tails ? "tails" : "all", stride, t1 - t0, left); t0 = t1; }
            if (tails && !left) {                               /* the tails in front of the chunks are final: now everything */
                tails = 0;                                      /* else -- a tail byte is one jump from its value, any other */
                left = 1;                                       /* byte two (the pass stores no marks on what it finishes, so */
            }                                                   /* a tail byte still shows its origin to whoever comes by) */
        }
        if (left) { fail(LZS_E_HIP, "%s: origins did not resolve", who); goto failed; }
        if (!dev) HIP_TRY(lzs_hip_d2h(out, (uint8_t *)d_out + before, produce, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
    }
    result = produce;
#undef HIP_TRY
    goto done;

failed:
    result = 0;
    if (rc == LZS_OK) rc = LZS_E_HIP;
    if (!dev) fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
done:
    { staging_t *s2 = staging_get(); if (s2) staging_trim(s2); }
    if (status) *status = rc;
    return result;
}

/* A batch of blocks decompressed like one long stream: every block is cut into segments of its
 * own (tables tell the kernels where a segment starts, where its stream ends, where its block's
 * output begins and must end), the scan rounds run over all of them at once -- a block's first
 * segment is always entered at bit 0 in the normal state -- and the origins are resolved over the
 * whole strided output.  d_in / d_out are the staged device buffers of host_batch(). */
LZS_HIDDEN int batch_decompress_segments(staging_t *st, void *stream, const char *who, void *d_out, size_t d_out_stride,
                                     uint32_t cap32, uint32_t *out_len, uint32_t *d_len, const void *d_in, size_t d_in_stride,
                                     const uint32_t *in_len_each, uint32_t in_len, size_t nblocks)
{
    int e = 0, rc = LZS_OK;
    size_t total_in = 0;
    for (size_t b = 0; b < nblocks; b++) total_in += in_len_each ? in_len_each[b] : in_len;
    const uint32_t seg = stream_dec_seg(total_in / 4);        /* (smaller than for one stream of that size: measured) */
    uint32_t in_extent = 0;                                    /* the readable bytes at d_in: the kernels' loads are bounded by it */
    for (size_t b = 0; b < nblocks; b++) {
        const size_t to = b * d_in_stride + (in_len_each ? in_len_each[b] : in_len);
        if (to > in_extent) in_extent = (uint32_t)to;
    }
    uint32_t nseg = 0;
    for (size_t b = 0; b < nblocks; b++) nseg += ((in_len_each ? in_len_each[b] : in_len) + seg - 1) / seg;
    const uint32_t extent = (uint32_t)(nblocks * d_out_stride);
    memset(out_len, 0, sizeof(uint32_t) * nblocks);
    if (nseg == 0) return LZS_OK;
    /* host tables: 12 words and 3 bytes per segment */
    uint32_t *tab = (uint32_t *)malloc((size_t)nseg * (12 * 4 + 4));
    if (!tab) return fail(LZS_E_NOMEM, "%s: out of host memory", who);
    uint32_t *entry = tab, *exits = entry + nseg, *count = exits + nseg, *start = count + nseg;
    uint32_t *base = start + nseg, *end = base + nseg, *floor_ = end + nseg, *limit = floor_ + nseg;
    uint32_t *seen = limit + nseg, *blk = seen + nseg, *spare = blk + nseg;   /* (spare: two unused rows) */
    uint8_t *dirty = (uint8_t *)(spare + 2 * (size_t)nseg), *ones = dirty + nseg, *first = ones + nseg;
    void *d_aux = NULL, *d_marks = NULL, *d_origin = NULL;
#define HIP_TRY(call, what) do { e = (call); if (e) { rc = hip_fail(e, what); goto done; } } while (0)
    e = staging_reserve(st, BUF_AUX, (size_t)nseg * (8 * 4 + 2) + 128, &d_aux);
    if (!e) e = staging_reserve(st, BUF_MARKS, (size_t)nseg * LZS_SCAN_MARK_WORDS * 4u, &d_marks);
    if (!e) e = staging_reserve(st, BUF_KEEP, 4 * (size_t)extent + 64, &d_origin);
    if (e) { rc = fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto done; }
    uint32_t *d_entry = (uint32_t *)d_aux, *d_exit = d_entry + nseg, *d_count = d_exit + nseg, *d_start = d_count + nseg;
    uint32_t *d_base = d_start + nseg, *d_end = d_base + nseg, *d_floor = d_end + nseg, *d_limit = d_floor + nseg;
    uint32_t *d_counters = d_limit + nseg;
    uint8_t *d_dirty = (uint8_t *)(d_counters + 2), *d_ones = d_dirty + nseg;
    {
        uint32_t k = 0;
        for (size_t b = 0; b < nblocks; b++) {
            const uint32_t len = in_len_each ? in_len_each[b] : in_len;
            for (uint32_t at = 0; at < len; at += seg, k++) {
                base[k] = (uint32_t)(b * d_in_stride) + at;
                end[k] = (uint32_t)(b * d_in_stride) + len;
                floor_[k] = (uint32_t)(b * d_out_stride);
                limit[k] = floor_[k] + cap32;
                blk[k] = (uint32_t)b;
                first[k] = at == 0;
                entry[k] = 0; dirty[k] = 1; seen[k] = 0xFFFFFFFFu;
            }
        }
    }
    const int debug = lzs_env()->stream_debug;
    double t0 = debug ? now_ms() : 0, t1;
    HIP_TRY(lzs_hip_h2d(d_base, base, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_end, end, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    for (uint32_t round = 0, ndirty = nseg; ndirty; round++) {
        HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_h2d(d_dirty, dirty, nseg, stream), "hipMemcpy H2D");
        HIP_TRY(lzs_hip_launch_scan_stream(d_in, 0, nseg, d_entry, d_dirty, d_exit, d_count, round == 0 ? d_ones : NULL,
                                           (uint32_t *)d_marks, round != 0, seg, 0, d_base, d_end, in_extent, stream), who);
        HIP_TRY(lzs_hip_d2h(exits, d_exit, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        if (round == 0) HIP_TRY(lzs_hip_d2h(ones, d_ones, nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_d2h(count, d_count, sizeof(uint32_t) * nseg, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        for (uint32_t k = 0; k < nseg; k++) if (dirty[k]) seen[k] = entry[k];
        ndirty = 0;
        int ended = 0, settled = 1;
        for (uint32_t k = 0; k < nseg; k++) {                  /* as in stream_decompress(), block by block */
            dirty[k] = 0;
            if (first[k]) { ended = 0; settled = 1; continue; }
            uint32_t want = exits[k - 1];
            if (want & LZS_SEG_STOP) { if (settled) ended = 1; else want = entry[k]; }
            if (ended) want = LZS_SEG_STOP;
            if (want != entry[k]) settled = 0;
            entry[k] = want;
            if (ended || want == seen[k]) continue;
            if (((want >> 8) & 1u) && (ones[k] == 2 || (ones[k] && (want & 3u) == 0))) {
                const uint32_t r = want & 0xFFu;
                const uint32_t nibbles = (seg * 8u - r + 3u) / 4u;
                exits[k] = (want & ~0xFFu) | (r + 4u * nibbles - seg * 8u);
                count[k] = 15u * nibbles;
                seen[k] = want;
                continue;
            }
            dirty[k] = 1;
            ndirty++;
        }
        if (ndirty) {
            HIP_TRY(lzs_hip_h2d(d_exit, exits, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_h2d(d_count, count, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
        }
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs batch decode: %zu blocks, %u segments of %u; round %u in %.2f ms, %u to redo\n", nblocks, nseg, seg, round, t1 - t0, ndirty); t0 = t1; }
    }
    {
        uint64_t total = 0;
        for (uint32_t k = 0; k < nseg; k++) {
            if (first[k]) total = 0;
            start[k] = floor_[k] + (uint32_t)(total < cap32 ? total : cap32);
            if (!(entry[k] & LZS_SEG_STOP)) total += count[k];
            out_len[blk[k]] = (uint32_t)(total < cap32 ? total : cap32);
        }
    }
    HIP_TRY(lzs_hip_h2d(d_entry, entry, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_start, start, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_floor, floor_, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_h2d(d_limit, limit, sizeof(uint32_t) * nseg, stream), "hipMemcpy H2D");
    HIP_TRY(lzs_hip_memset(d_counters, 0, 8, stream), "hipMemset");
    HIP_TRY(lzs_hip_memset(d_origin, 0xFF, 4 * (size_t)extent, stream), "hipMemset");      /* everything "clean" */
    HIP_TRY(lzs_hip_launch_decode_stream(d_out, extent, (uint32_t *)d_origin, d_counters, d_in, 0, in_extent, nseg, d_entry, d_start,
                                         seg, 0, d_base, d_end, d_floor, d_limit, stream), who);
    {
        uint32_t open[2] = {0, 0};
        HIP_TRY(lzs_hip_d2h(open, d_counters, 8, stream), "hipMemcpy D2H");
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs batch decode: tables + memset + decode in %.2f ms, %u bytes with an origin elsewhere\n", t1 - t0, open[0]); t0 = t1; }
        /* origins never leave their block: one workgroup per block resolves them to the end */
        HIP_TRY(lzs_hip_h2d(d_len, out_len, sizeof(uint32_t) * nblocks, stream), "hipMemcpy H2D");
        if (open[0]) HIP_TRY(lzs_hip_launch_resolve_blocks(d_out, (uint32_t *)d_origin, d_out_stride, d_len, (uint32_t)nblocks, stream), who);
        HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        if (debug) { t1 = now_ms(); fprintf(stderr, "liblzs batch decode: resolve in %.2f ms\n", t1 - t0); t0 = t1; }
    }
#undef HIP_TRY
done:
    free(tab);
    return rc;
}

int lzs_decompress_stream_device(void *d_out, size_t out_cap, size_t *out_len, const void *d_in, size_t in_len)
{
    if (!out_len) return fail(LZS_E_ARG, "lzs_decompress_stream_device: out_len is NULL");
    *out_len = 0;
    if ((!d_out && out_cap) || (!d_in && in_len)) return fail(LZS_E_ARG, "lzs_decompress_stream_device: NULL buffer");
    if (in_len == 0 || out_cap == 0) return LZS_OK;
    if (in_len > LZS_BLOCK_MAX) return fail(LZS_E_ARG, "lzs_decompress_stream_device: stream exceeds LZS_BLOCK_MAX");
    int rc = LZS_OK;
    const size_t got = stream_decompress((uint8_t *)d_out, out_cap, (const uint8_t *)d_in, in_len, 1, &rc, 0, NULL);
    if (got == SIZE_MAX) return fail(LZS_E_ARG, "lzs_decompress_stream_device: output of 4 GiB or more");
    *out_len = got;
    return rc;
}

