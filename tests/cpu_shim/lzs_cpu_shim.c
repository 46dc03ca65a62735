/*
 * tests/cpu_shim/lzs_cpu_shim.c -- a CPU implementation of csrc/lzs_hip_shim.h.  TEST INFRASTRUCTURE ONLY.
 *
 * The product's host code (lzs_host.c, lzs_stream.c, lzs_incremental.c, lzs_pipeline.c, lzs_hostcodec.c: ~2 400 lines of
 * staging, pinned-piece rings, worker threads, carry state across pieces, dirty-segment re-entry) only ever ran against
 * the real device, where no sanitizer can look at it (GPU AddressSanitizer is not available on the pool, and a race or
 * an out-of-bounds read that lands in mapped memory passes every parity test).  Linked with THIS file instead of
 * lzs_kernels.hip, the unchanged host sources run under -fsanitize=address,undefined and, separately, -fsanitize=thread
 * on the CPU (tests/test_sanitizers.py builds and runs both):
 *   * "device" memory is malloc'ed memory, so an overrun of a staging buffer is a heap overflow ASan sees;
 *   * streams and events are synchronous (every copy is done when its call returns; an event is complete once recorded);
 *   * the block launches (compress, decompress, compact) are backed by the oracle (oracle/lzs_oracle.c), block by block,
 *     with the kernels' contracts: fixed-stride slots, lengths cut at the capacity, nothing past a slot touched;
 *   * the segment launches of lzs_stream.c's compress side (segments, stitch, extend-resume) and the incremental
 *     decoder's launch (decode-resume) are restated here from the kernels' documented contracts, serially;
 *   * the many-wavefront decompression launches (scan / decode / resolve) return hipErrorNotSupported: the harness runs
 *     with LZS_ONE_WAVE=1, which keeps those calls on the one-wavefront route, as tools/README.md says.
 * Nothing here is the product, nothing in the product links this.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lzs_hip_shim.h"

size_t   lzs_oracle_compress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);
size_t   lzs_oracle_decompress(uint8_t *out, size_t cap, const uint8_t *in, size_t n);
unsigned lzs_oracle_search(const uint8_t *in, size_t n, size_t c, unsigned *best_off);

#define E_NOT_SUPPORTED 801         /* hipErrorNotSupported */
#define E_OUT_OF_MEMORY 2           /* hipErrorOutOfMemory */

/* ---- HIP's sticky last error, modelled: a failed allocation stays the answer of "get last error" until fetched, and
 * every launcher returns the last error after its launch (what ADVICE r04's stale-error finding is about) */
static _Thread_local int g_last_error;
static int launched(void) { const int e = g_last_error; g_last_error = 0; return e; }
void lzs_hip_clear_error(void) { g_last_error = 0; }

int lzs_hip_device_count(int *count) { *count = 1; return 0; }
int lzs_hip_describe(char *buf, size_t cap) { snprintf(buf, cap, "cpu shim (tests/cpu_shim): the oracle behind the launchers, synchronous streams"); return 0; }
const char *lzs_hip_strerror(int e) { return e == E_NOT_SUPPORTED ? "operation not supported (cpu shim)" : e == E_OUT_OF_MEMORY ? "out of memory (cpu shim)" : e ? "error (cpu shim)" : "no error"; }

int lzs_hip_total_memory(size_t *bytes) { *bytes = (size_t)8 << 30; return 0; }    /* (1/32 of it is below the library's floor of 640 MiB) */
/* what is outstanding, for the tests of lzs_release_thread_cache() and of the per-thread limit: bytes of "device" memory and
 * of pinned host memory, and the count of live allocations of either kind (malloc_usable_size: the heap's own bookkeeping) */
#include <malloc.h>
static size_t g_dev_bytes, g_host_bytes, g_live;
void lzs_shim_outstanding(size_t *dev_bytes, size_t *host_bytes, size_t *live)
{
    if (dev_bytes) *dev_bytes = __atomic_load_n(&g_dev_bytes, __ATOMIC_RELAXED);
    if (host_bytes) *host_bytes = __atomic_load_n(&g_host_bytes, __ATOMIC_RELAXED);
    if (live) *live = __atomic_load_n(&g_live, __ATOMIC_RELAXED);
}
static int shim_alloc(void **p, size_t bytes, size_t *counter)
{
    if (bytes > ((size_t)1 << 40)) { *p = NULL; g_last_error = E_OUT_OF_MEMORY; return E_OUT_OF_MEMORY; }
    /* 0xCD: what lies in fresh device memory is nobody's zeros */
    *p = malloc(bytes ? bytes : 1);
    if (!*p) { g_last_error = E_OUT_OF_MEMORY; return E_OUT_OF_MEMORY; }
    memset(*p, 0xCD, bytes ? bytes : 1);
    __atomic_add_fetch(counter, malloc_usable_size(*p), __ATOMIC_RELAXED);
    __atomic_add_fetch(&g_live, 1, __ATOMIC_RELAXED);
    return 0;
}
static void shim_free(void *p, size_t *counter)
{
    if (!p) return;
    __atomic_sub_fetch(counter, malloc_usable_size(p), __ATOMIC_RELAXED);
    __atomic_sub_fetch(&g_live, 1, __ATOMIC_RELAXED);
    free(p);
}
int lzs_hip_malloc(void **p, size_t bytes) { return shim_alloc(p, bytes, &g_dev_bytes); }
int lzs_hip_free(void *p) { shim_free(p, &g_dev_bytes); return 0; }
int lzs_hip_host_malloc(void **p, size_t bytes) { return shim_alloc(p, bytes, &g_host_bytes); }
int lzs_hip_host_malloc_staging(void **p, size_t bytes) { return shim_alloc(p, bytes, &g_host_bytes); }
int lzs_hip_host_free(void *p) { shim_free(p, &g_host_bytes); return 0; }

int lzs_hip_stream_create(void **s) { *s = malloc(8); return *s ? 0 : E_OUT_OF_MEMORY; }
int lzs_hip_stream_destroy(void *s) { free(s); return 0; }
int lzs_hip_stream_sync(void *s) { (void)s; return 0; }
int lzs_hip_event_create(void **e) { *e = calloc(1, 8); return *e ? 0 : E_OUT_OF_MEMORY; }
int lzs_hip_event_destroy(void *e) { free(e); return 0; }
int lzs_hip_event_record(void *e, void *s) { (void)s; *(volatile int *)e = 1; return 0; }
int lzs_hip_event_sync(void *e) { (void)e; return 0; }
int lzs_hip_event_done(void *e) { (void)e; return 1; }
int lzs_hip_stream_wait_event(void *s, void *e) { (void)s; (void)e; return 0; }
int lzs_hip_h2d(void *d, const void *s, size_t n, void *st) { (void)st; if (n) memcpy(d, s, n); return 0; }
int lzs_hip_d2h(void *d, const void *s, size_t n, void *st) { (void)st; if (n) memcpy(d, s, n); return 0; }
int lzs_hip_d2d(void *d, const void *s, size_t n, void *st) { (void)st; if (n) memmove(d, s, n); return 0; }
int lzs_hip_memset(void *d, int v, size_t n, void *st) { (void)st; if (n) memset(d, v, n); return 0; }
int lzs_hip_words_to_host(uint32_t *h, const uint32_t *d, size_t nwords, void *st) { (void)st; if (nwords) memcpy(h, d, 4 * nwords); return launched(); }
int lzs_hip_chain_mode(void *stream, int *mode) { (void)stream; *mode = 0; return 0; }
int lzs_hip_load_check_state(int dev) { (void)dev; return 0; }
unsigned lzs_hip_dec_segment_bytes(void) { return 8192; }

/* ---- block launches: block b is the one-shot call on block b (the kernels' contract) */
int lzs_hip_launch_compress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len, const void *d_in, size_t in_stride,
                            const uint32_t *d_in_len, uint32_t in_len, uint32_t nblocks, void *stream)
{
    (void)stream;
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t n = d_in_len ? d_in_len[b] : in_len;
        d_out_len[b] = (uint32_t)lzs_oracle_compress((uint8_t *)d_out + (size_t)b * out_stride, out_cap, (const uint8_t *)d_in + (size_t)b * in_stride, n);
    }
    return launched();
}
int lzs_hip_classify_blocks(uint32_t *d_codes, const void *d_in, size_t in_stride, const uint32_t *d_in_len, uint32_t in_len, uint32_t nblocks, void *stream)
{ (void)d_in; (void)in_stride; (void)d_in_len; (void)in_len; (void)stream; for (uint32_t b = 0; b < nblocks; b++) d_codes[b] = 0xFFFFFFF1u; return launched(); }
int lzs_hip_launch_decompress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len, const void *d_in, size_t in_stride,
                              const uint32_t *d_in_len, uint32_t in_len, uint32_t nblocks, void *stream)
{
    (void)stream;
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t n = d_in_len ? d_in_len[b] : in_len;
        d_out_len[b] = (uint32_t)lzs_oracle_decompress((uint8_t *)d_out + (size_t)b * out_stride, out_cap, (const uint8_t *)d_in + (size_t)b * in_stride, n);
    }
    return launched();
}
int lzs_hip_launch_decompress_concat(void *a, size_t b, uint32_t c, uint32_t *d, const void *e, size_t f, const uint32_t *g, uint32_t h, uint32_t i, void *j)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; return E_NOT_SUPPORTED; }

int lzs_hip_launch_compact(void *d_dense, uint64_t *d_offsets, const void *d_slots, size_t slot_stride, const uint32_t *d_len, uint32_t nblocks, void *stream)
{
    (void)stream;
    uint64_t at = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
        d_offsets[b] = at;
        memcpy((uint8_t *)d_dense + at, (const uint8_t *)d_slots + (size_t)b * slot_stride, d_len[b]);
        at += d_len[b];
    }
    d_offsets[nblocks] = at;
    return launched();
}

/* ---- MSB-first bits ORed into zeroed memory at a bit offset (what the segment kernels and the stitch do) */
static void or_bits(uint8_t *dst, size_t limit_bytes, uint64_t at, uint32_t value, unsigned width)
{
    for (unsigned i = 0; i < width; i++) {
        const uint64_t bit = at + i;
        if ((value >> (width - 1u - i)) & 1u) {
            if ((bit >> 3) < limit_bytes) dst[bit >> 3] |= (uint8_t)(0x80u >> (bit & 7u));
        }
    }
}
static unsigned lcp_at(const uint8_t *in, size_t a, size_t b, unsigned lim) { unsigned k = 0; while (k < lim && in[a + k] == in[b + k]) k++; return k; }

/* One segment of a stream (lzs_compress_segments_kernel's contract, csrc/kernels/compress_wg.inc): the stream in[0..n) from
 * token start c0 up to the first token start >= e; bits counted from bit `head0` of dst (zeroed) and stored while they
 * fall inside `limit` bytes; reports the exit position, the bit count and -- open_info -- {offset, start} of the last
 * token if that is a match whose nibbles reach n exactly. */
static void segment_job(uint8_t *dst, size_t limit, unsigned head0, const uint8_t *in, uint32_t n, uint32_t c0, uint32_t e,
                        uint32_t *exit_pos, uint64_t *nbits, uint32_t *open_info)
{
    uint64_t at = head0;
    uint32_t c = c0;
    if (open_info) { open_info[0] = 0; open_info[1] = 0; }
    while (c < e) {
        unsigned off = 0;
        const unsigned len = lzs_oracle_search(in, n, c, &off);
        if (len < 2) { or_bits(dst, limit, at, in[c], 9); at += 9; c++; continue; }
        const unsigned first = len < 8 ? len : 8;
        if (off <= 127) { or_bits(dst, limit, at, (3u << 7) | off, 9); at += 9; } else { or_bits(dst, limit, at, (2u << 11) | off, 13); at += 13; }
        if (first <= 4) { or_bits(dst, limit, at, first - 2, 2); at += 2; } else { or_bits(dst, limit, at, 7 + first, 4); at += 4; }
        const uint32_t start = c;
        c += first;
        if (first == 8) {
            unsigned x;
            do {
                const unsigned lim = n - c < 15 ? n - c : 15;
                x = lcp_at(in, c, c - off, lim);
                or_bits(dst, limit, at, x, 4); at += 4;
                c += x;
            } while (x == 15);
            if (open_info && c == n) { open_info[0] = off; open_info[1] = start; }
        }
    }
    *exit_pos = c;
    *nbits = at - head0;
}

int lzs_hip_launch_compress_segments(void *d_slots, size_t slot_stride, const void *d_in, uint32_t n, uint32_t seg, uint32_t nseg,
                                     const uint32_t *d_entry, const uint8_t *d_dirty, uint32_t *d_exit, uint64_t *d_nbits,
                                     void *d_out, const uint64_t *d_bit_at, uint32_t lim, uint32_t *d_open, void *stream)
{
    (void)stream;
    for (uint32_t k = 0; k < nseg; k++) {
        if (d_dirty && !d_dirty[k]) continue;
        const uint32_t s = k * seg;
        const uint32_t e = s + seg < lim ? s + seg : lim;
        const uint32_t c0 = d_entry[k];
        uint32_t *open = d_open ? d_open + 2 * k : NULL;
        if (c0 >= e) { d_exit[k] = c0; d_nbits[k] = 0; if (open) { open[0] = 0; open[1] = 0; } continue; }
        if (!d_out) {
            uint8_t *slot = (uint8_t *)d_slots + (size_t)k * slot_stride;
            memset(slot, 0, slot_stride);                        /* (the kernel's bit ring starts as zeros: a slot holds its bits and nothing else up to them) */
            segment_job(slot, slot_stride, 0, (const uint8_t *)d_in, n, c0, e, &d_exit[k], &d_nbits[k], open);
        } else {
            /* a segment whose bits did not fit its slot: ORed straight into the output at its bit offset */
            segment_job((uint8_t *)d_out + (d_bit_at[k] >> 3), (size_t)-1, (unsigned)(d_bit_at[k] & 7u), (const uint8_t *)d_in, n, c0, e, &d_exit[k], &d_nbits[k], open);
        }
    }
    return launched();
}

int lzs_hip_launch_stitch_segments(void *d_out, const void *d_slots, size_t slot_stride, const uint64_t *d_bit_at, const uint64_t *d_nbits,
                                   uint32_t nseg, int end_marker, void *stream)
{
    (void)stream;
    for (uint32_t k = 0; k < nseg; k++) {
        if (d_nbits[k] > 8u * (uint64_t)slot_stride) continue;   /* (ORed in directly by the second segment launch) */
        const uint8_t *slot = (const uint8_t *)d_slots + (size_t)k * slot_stride;
        for (uint64_t i = 0; i < d_nbits[k]; i++)
            if ((slot[i >> 3] >> (7u - (i & 7u))) & 1u) {
                const uint64_t bit = d_bit_at[k] + i;
                ((uint8_t *)d_out)[bit >> 3] |= (uint8_t)(0x80u >> (bit & 7u));
            }
    }
    if (end_marker && nseg) or_bits((uint8_t *)d_out, (size_t)-1, d_bit_at[nseg - 1] + d_nbits[nseg - 1], 0x180u, 9);
    return launched();
}

int lzs_hip_launch_extend_resume(void *d_out, uint32_t bit0, const void *d_in, uint32_t n, uint32_t c0, uint32_t off, int last,
                                 uint32_t *d_result, void *stream)
{
    (void)stream;
    const uint8_t *in = (const uint8_t *)d_in;
    uint32_t c = c0;
    while (c < n && in[c] == in[c - off]) c++;
    const uint32_t run = c - c0;
    const int open = c == n && !last;
    const uint32_t full = run / 15u;
    uint64_t at = bit0;
    for (uint32_t i = 0; i < full; i++, at += 4) or_bits((uint8_t *)d_out, (size_t)-1, at, 15, 4);
    if (!open) { or_bits((uint8_t *)d_out, (size_t)-1, at, run - 15u * full, 4); at += 4; }
    const uint64_t bits = at - bit0;
    d_result[0] = open ? c0 + 15u * full : c;
    d_result[1] = open ? 1u : 0u;
    d_result[2] = (uint32_t)bits; d_result[3] = (uint32_t)(bits >> 32);
    return launched();
}

/* ---- the incremental decoder's launch (lzs_decode_resume_kernel's contract, csrc/kernels/compact_resume.inc): one
 * call's worth of decoding from the state block, the same stop rules at token granularity */
int lzs_hip_launch_decode_resume(lzs_dec_resume_t *st, const void *d_in, uint32_t n, void *d_out, uint32_t cap, void *stream)
{
    (void)stream;
    const uint8_t *in = (const uint8_t *)d_in;
    uint8_t *out = (uint8_t *)d_out;
    /* the history and what is produced as one array, the way the kernel's ring sees them */
    const uint32_t base = st->hist_len;
    uint8_t *ring = (uint8_t *)malloc((size_t)base + cap + 1);
    if (!ring) return E_OUT_OF_MEMORY;
    memcpy(ring, st->hist, base);
    uint64_t bits = (uint64_t)st->bitq << 32;
    uint32_t have = st->qlen;
    const uint32_t carried = have;
    uint32_t off = st->off, rem = st->rem, ipos = 0, count = base, status = 0;
    int extended = st->extended != 0;
    const uint32_t limit = base + cap;
    for (;;) {
        while (have <= 32 && ipos < n) {                        /* a word at a time, like the kernel */
            const uint32_t nb = n - ipos < 4 ? n - ipos : 4;
            uint32_t w = 0;
            for (uint32_t i = 0; i < nb; i++) w |= (uint32_t)in[ipos + i] << (24 - 8 * i);
            bits |= (uint64_t)w << (32 - have);
            have += 8 * nb;
            ipos += 4;
        }
        if (have == 0) { status |= LZS_INC_INPUT_FINISHED | LZS_INC_INPUT_STARVED; break; }
        if (rem) {
            if (count == limit) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }
            ring[count] = count >= off ? ring[count - off] : 0;
            count++; rem--;
            continue;
        }
        int starved = 0;
        if (extended) {
            if (have < 4) starved = 1;
            else { const uint32_t e = (uint32_t)(bits >> 60); bits <<= 4; have -= 4; rem = e; if (e != 15) extended = 0; }
        } else if ((bits >> 63) == 0) {
            if (have < 9) starved = 1;
            else if (count >= limit) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }
            else { ring[count++] = (uint8_t)(bits >> 55); bits <<= 9; have -= 9; }
        } else {
            const int is_short = (int)((bits >> 62) & 1u);
            const uint32_t used = is_short ? 9u : 13u;
            if (have < used) starved = 1;
            else {
                const uint32_t o = is_short ? (uint32_t)(bits >> 55) & 0x7Fu : (uint32_t)(bits >> 51) & 0x7FFu;
                if (o == 0) {
                    bits <<= used; have -= used;
                    if (is_short) { const uint32_t pad = have & 7u; bits <<= pad; have -= pad; status |= LZS_INC_END_MARKER; break; }
                    off = 0;
                } else {
                    const uint32_t code = (uint32_t)((bits << used) >> 60);
                    const uint32_t width = code < 0xCu ? 2u : 4u;
                    if (have < used + width) starved = 1;
                    else {
                        const uint32_t len = code < 0xCu ? 2u + (code >> 2) : 5u + (code - 0xCu);
                        bits <<= used + width; have -= used + width;
                        off = o; rem = len; extended = len == 8;
                    }
                }
            }
        }
        if (starved) { status |= LZS_INC_INPUT_STARVED; break; }
    }
    memcpy(out, ring + base, count - base);
    const uint32_t fed = ipos < n ? ipos : n;
    const uint32_t consumed = carried + 8u * fed - have;
    const uint32_t fed_left = consumed >= carried ? have : 8u * fed;
    const uint32_t back = (status & LZS_INC_INPUT_STARVED) ? 0u : fed_left >> 3;
    have -= 8u * back;
    const uint32_t hist_len = count < 2047u ? count : 2047u;
    memcpy(st->hist, ring + count - hist_len, hist_len);
    st->bitq = (uint32_t)(bits >> 32) & (have ? ~0u << (32u - have) : 0u);
    st->qlen = have; st->off = off; st->rem = rem; st->extended = extended ? 1u : 0u;
    st->hist_len = hist_len; st->in_used = fed - back; st->out_made = count - base; st->status = status;
    free(ring);
    return launched();
}

/* ---- the many-wavefront decompression: not modelled (LZS_ONE_WAVE=1 keeps the harness off it) */
int lzs_hip_launch_scan_stream(const void *a, uint32_t b, uint32_t c, const uint32_t *d, const uint8_t *e, uint32_t *f, uint32_t *g, uint8_t *h,
                               uint32_t *i, int j, uint32_t k, int l, const uint32_t *m, const uint32_t *n, uint32_t o, void *p)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; (void)k; (void)l; (void)m; (void)n; (void)o; (void)p; return E_NOT_SUPPORTED; }
int lzs_hip_launch_decode_stream(void *a, uint32_t b, uint32_t *c, uint32_t *d, const void *e, uint32_t f, uint32_t g, uint32_t h, const uint32_t *i,
                                 const uint32_t *j, uint32_t k, int l, const uint32_t *m, const uint32_t *n, const uint32_t *o, const uint32_t *p, void *q)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; (void)k; (void)l; (void)m; (void)n; (void)o; (void)p; (void)q; return E_NOT_SUPPORTED; }
int lzs_hip_launch_resolve_blocks(void *a, uint32_t *b, size_t c, const uint32_t *d, uint32_t e, void *f)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; return E_NOT_SUPPORTED; }
int lzs_hip_launch_resolve_stream(void *a, uint32_t *b, uint32_t c, uint32_t d, uint32_t *e, int f, void *g)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; return E_NOT_SUPPORTED; }
int lzs_hip_launch_resolve_tails(void *a, uint32_t *b, uint32_t c, const uint32_t *d, uint32_t e, uint32_t f, uint32_t g, uint32_t *h, void *i)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; return E_NOT_SUPPORTED; }
int lzs_hip_launch_resolve_chunks(void *a, uint32_t *b, uint32_t c, const uint32_t *d, uint32_t e, uint32_t f, uint32_t g, void *h)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; return E_NOT_SUPPORTED; }
