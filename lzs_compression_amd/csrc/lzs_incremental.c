/*
 * lzs_incremental.c -- the reference's incremental interface (c/src/liblzs/lzs.h:90-232) served by
 * the device: what must survive between calls lives in the caller's parameter block.
 */
#include "lzs_internal.h"

/* ------------------------------------------------------- incremental interface: decoding */
/* reference lzs-decompression.c:420-743.  What the reference keeps in its private members is
 * kept here, in the same bytes of the caller's block, in this layout; every call ships it to
 * the device with the input and back (lzs_decode_resume_kernel). */
typedef struct __attribute__((packed)) {
    uint32_t bitq;                  /* bits of an unfinished token, left-aligned */
    uint16_t off;                   /* copy in progress: offset */
    uint16_t hist_len;              /* bytes in hist[], oldest first */
    uint8_t  qlen, rem, extended;   /* bits in bitq; copy bytes left; a length nibble follows */
    uint8_t  hist[LZS_MAX_HISTORY_SIZE];
    uint8_t  big_log;               /* how much input a pass of the many-wavefront path takes: 1 MiB << big_log (learnt over the calls) */
} dec_priv_t;
#define DEC_PRIV_AT 36u
#define DEC_SMALL     16384u        /* calls up to this much input and output take the short way */
#define INC_DEC_STREAM_MIN 16384u   /* pieces from this size on go to many wavefronts first */
#define DEC_STATE_PAD 2112u         /* sizeof(lzs_dec_resume_t) rounded up to 64 */
#define INC_BOX_BYTES (2 * (size_t)DEC_SMALL + DEC_STATE_PAD + 64)
_Static_assert(sizeof(lzs_dec_resume_t) <= DEC_STATE_PAD, "state fits its slot");
_Static_assert(sizeof(LzsDecompressParameters_t) == 2096, "size of the reference's LzsDecompressParameters_t");
_Static_assert(sizeof(LzsCompressParameters_t) == 14432, "size of the reference's LzsCompressParameters_t");
_Static_assert(DEC_PRIV_AT + sizeof(dec_priv_t) <= sizeof(LzsDecompressParameters_t), "private state fits");

/* one line on stderr per thread and failure text, not one per call of a loop that keeps calling */
static __thread char inc_noted[64];
static void inc_failed_note(const char *who)
{
    if (strncmp(inc_noted, tls_error, sizeof(inc_noted) - 1) == 0 && inc_noted[0]) return;
    memcpy(inc_noted, tls_error, sizeof(inc_noted) - 1); inc_noted[sizeof(inc_noted) - 1] = 0;
    fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
}

void lzs_decompress_init(LzsDecompressParameters_t *p)
{
    inc_noted[0] = 0;               /* a new stream: its first failure is reported again */
    if (!p) return;
    p->status = LZS_D_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
}

/* The call on the host route (lzs_hostcodec.c): the one wavefront's work -- same state, same stop rules -- done by the
 * calling thread, straight from the caller's input into the caller's output. */
static size_t dec_incremental_host(LzsDecompressParameters_t *p, dec_priv_t *pv)
{
    size_t made = 0;
    for (;;) {
        const size_t take = p->inLength < ((size_t)1 << 30) ? p->inLength : ((size_t)1 << 30);
        const size_t cap = p->outLength < 0xF0000000u ? p->outLength : 0xF0000000u;
        /* (the state lives in the caller's block, packed: its fields by value, the history in place) */
        uint32_t bitq = pv->bitq, qlen = pv->qlen, off = pv->off, rem = pv->rem, extended = pv->extended, hist_len = pv->hist_len;
        uint32_t in_used = 0, out_made = 0, status = 0;
        hostcodec_decode_resume_fields(&bitq, &qlen, &off, &rem, &extended, pv->hist, &hist_len, p->inPtr, (uint32_t)take, p->outPtr, (uint32_t)cap,
                                       &in_used, &out_made, &status);
        pv->bitq = bitq; pv->qlen = (uint8_t)qlen; pv->off = (uint16_t)off; pv->rem = (uint8_t)rem;
        pv->extended = (uint8_t)extended; pv->hist_len = (uint16_t)hist_len;
        p->inPtr += in_used;   p->inLength -= in_used;
        p->outPtr += out_made; p->outLength -= out_made;
        made += out_made;
        /* stopped only because of this loop's own limits: go on */
        if ((status & LZS_INC_INPUT_STARVED) && p->inLength) continue;
        if ((status & LZS_INC_NO_OUTPUT_SPACE) && p->outLength) continue;
        p->status = (uint8_t)status;
        return made;
    }
}

size_t lzs_decompress_incremental(LzsDecompressParameters_t *p)
{
    const char *who = "lzs_decompress_incremental";
    if (!p) return 0;
    dec_priv_t *pv = (dec_priv_t *)((uint8_t *)p + DEC_PRIV_AT);
    size_t made = 0;
    int e = 0;
    p->status = LZS_D_STATUS_NONE;
    tls_error[0] = 0;
    /* nothing to read and no bit queued: the answer needs no device (:475-478; the reference
     * stops there even with a copy pending) */
    if (p->inLength == 0 && pv->qlen == 0) {
        p->status = LZS_D_STATUS_INPUT_FINISHED | LZS_D_STATUS_INPUT_STARVED;
        return 0;
    }
    staging_t *st = NULL;
    if ((p->inLength && !p->inPtr) || (p->outLength && !p->outPtr)) {
        fail(LZS_E_ARG, "%s: NULL buffer", who);
        goto failed;                /* terminal like every other failure: a loop that never looks at ERROR ends */
    }
    /* a small piece: the host route (by size on a box with its device -- require_device() first -- or by name) */
    if (route_on_host(p->inLength, INC_DEC_HOST_MAX)) {
        if (lzs_env()->route != LZS_ROUTE_HOST && require_device() != LZS_OK) goto failed;
        return dec_incremental_host(p, pv);
    }
    if (require_device() != LZS_OK) goto failed;
    st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
#define HIP_TRY(call, what) do { e = (call); if (e) { hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    lzs_dec_resume_t h;
    memset(&h, 0, sizeof(h));
    size_t wave_limit = (size_t)16 << 20;
    /* how much input one pass of the many-wavefront path looks at: it scans ALL of it but decodes
     * only up to the first end marker, so a buffer of concatenated streams (what lzs-compress
     * writes with -b) would be scanned once per marker in full; starts small, doubles only while
     * the pass before was used up to its end
     * (round 6) ... and what was learnt stays in the block for the next call: a stream without early markers that is fed in
     * pieces of 16 MiB is decoded in one pass a piece, not in five (1 + 2 + 4 + 8 + 1 MiB, each with its scan, decode and
     * resolve rounds: 12 ms for an 8 MiB piece, profiles/r06/inc_dec_stages.txt) */
    size_t big_limit = (size_t)1 << (20u + (pv->big_log <= 8u ? pv->big_log : 0u));
    for (;;) {
        /* A large piece first goes to many wavefronts (stream_decompress) as far as
         * whole segments can be decoded; what is left -- the segment with the end marker, the
         * unfinished token at the end of the input, the last bytes before the output is full, a
         * copy still running -- is the one wavefront's below. */
        if (pv->rem == 0 && p->inLength >= INC_DEC_STREAM_MIN && p->outLength >= 4096u && !lzs_env()->one_wave) {
            const size_t big = p->inLength < big_limit ? p->inLength : big_limit;
            const uint32_t nb = (pv->qlen + 7u) / 8u;
            uint8_t pre[4] = {0, 0, 0, 0};
            const uint32_t v = pv->qlen ? pv->bitq >> (32u - pv->qlen) : 0u;     /* the queued bits, right-aligned */
            for (uint32_t i = 0; i < nb; i++) pre[i] = (uint8_t)(v >> (8u * (nb - 1u - i)));
            dec_piece_t dp;
            memset(&dp, 0, sizeof(dp));
            dp.prefix = pre; dp.prefix_len = nb;
            dp.entry0 = (8u * nb - pv->qlen) | ((uint32_t)(pv->extended != 0) << 8) | ((pv->extended ? (uint32_t)pv->off : 0u) << 9);
            dp.hist = pv->hist; dp.hist_len = pv->hist_len;
            const size_t room = p->outLength < 0xE0000000u ? p->outLength : 0xE0000000u;
            int rc = LZS_OK;
            const size_t got = stream_decompress(p->outPtr, room, p->inPtr, nb + big, 0, &rc, 0, &dp);
            if (rc != LZS_OK) goto failed_quiet;
            if (got != SIZE_MAX && dp.segs_done > 0) {
                const size_t at = (size_t)dp.segs_done * dp.seg + ((dp.next_entry & 0xFFu) >> 3);   /* in prefix + input */
                const uint32_t b = dp.next_entry & 7u;
                if (at < nb || at - nb + (b ? 1u : 0u) > big || (dp.next_entry & LZS_SEG_STOP) || got > room) {
                    fail(LZS_E_HIP, "%s: inconsistent state from the device", who);
                    goto failed;
                }
                pv->qlen = b ? (uint8_t)(8u - b) : 0;
                pv->bitq = b ? (uint32_t)p->inPtr[at - nb] << (24u + b) : 0u;
                pv->extended = (uint8_t)((dp.next_entry >> 8) & 1u);
                if (pv->extended) pv->off = (uint16_t)((dp.next_entry >> 9) & 0x7FFu);
                /* the history: the last 2047 bytes of what was there and what came now */
                if (got >= LZS_MAX_HISTORY_SIZE) {
                    memcpy(pv->hist, p->outPtr + got - LZS_MAX_HISTORY_SIZE, LZS_MAX_HISTORY_SIZE);
                    pv->hist_len = LZS_MAX_HISTORY_SIZE;
                } else {
                    const size_t keep = (size_t)pv->hist_len + got > LZS_MAX_HISTORY_SIZE ? LZS_MAX_HISTORY_SIZE - got : pv->hist_len;
                    memmove(pv->hist, pv->hist + pv->hist_len - keep, keep);
                    memcpy(pv->hist + keep, p->outPtr, got);
                    pv->hist_len = (uint16_t)(keep + got);
                }
                const size_t used = at - nb + (b ? 1u : 0u);
                if (used + 4u * (size_t)dp.seg >= big) {
                    /* used up to its end: the next pass may take twice as much -- if this one was cut by its limit, not by the piece */
                    if (big == big_limit && big_limit < ((size_t)256 << 20)) { big_limit *= 2; pv->big_log++; }
                } else if (used < big / 4u && pv->big_log > 0u) {
                    /* an end marker early in the pass (concatenated streams): scan less next time */
                    big_limit /= 2; pv->big_log--;
                }
                p->inPtr += used;  p->inLength -= used;
                p->outPtr += got;  p->outLength -= got;
                made += got;
                if (p->inLength == 0 && pv->qlen == 0) { p->status = LZS_D_STATUS_INPUT_FINISHED | LZS_D_STATUS_INPUT_STARVED; break; }
            }
            /* not even one whole segment this time: the wavefront takes the next stretch */
            wave_limit = 4u * (size_t)(dp.seg ? dp.seg : 8192u);
        }
        /* A copy still running from the call before keeps a large piece from the many wavefronts (they start at a
         * token): the one wavefront then only finishes it and walks a short stretch -- not the whole piece, 0.6 us a
         * token (half a MiB: 176 ms instead of 0.7) -- and the loop comes round to the many. */
        if (pv->rem != 0 && p->inLength >= INC_DEC_STREAM_MIN && p->outLength >= 4096u && !lzs_env()->one_wave && wave_limit > 2048u)
            wave_limit = 2048u;
        /* one launch takes at most 16 MiB of input; its output is bounded by 30x that
         * (a length nibble stands for 15 bytes) */
        const size_t take = p->inLength < wave_limit ? p->inLength : wave_limit;
        const size_t most = 30u * (take + 4u) + 64u;
        const size_t cap = p->outLength < most ? p->outLength : most;
        void *d_in = NULL, *d_out = NULL, *d_state = NULL;
        h.bitq = pv->bitq; h.qlen = pv->qlen; h.off = pv->off; h.rem = pv->rem;
        h.extended = pv->extended; h.hist_len = pv->hist_len;
        memcpy(h.hist, pv->hist, pv->hist_len);
        if (take <= DEC_SMALL && cap <= DEC_SMALL) {
            /* A small call is all latency: one copy in ([input | state], the input right-aligned
             * before the state), one launch, one copy out ([state | output]), one wait. */
            /* The box is pinned host memory the device reads and writes in place (hipHostMalloc:
             * mapped, coherent): no copies are queued at all -- input and state are put there, the
             * one wavefront works on them over the bus, and the wait for the launch is the only
             * round trip (two queued copies and their bookkeeping were most of the 0.11 ms). */
            if (!st->host_box) {
                void *pinned = NULL;
                e = lzs_hip_host_malloc(&pinned, INC_BOX_BYTES);
                if (e) { fail(LZS_E_NOMEM, "%s: pinned host allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
                st->host_box = (uint8_t *)pinned;
            }
            uint8_t *box = st->host_box;
            const size_t in_at = DEC_SMALL - ((take + 3u) & ~(size_t)3u);
            memcpy(box + in_at, p->inPtr, take);
            h.in_used = h.out_made = h.status = 0;
            memcpy(box + DEC_SMALL, &h, sizeof(h));
            HIP_TRY(lzs_hip_launch_decode_resume((lzs_dec_resume_t *)(box + DEC_SMALL), box + in_at, (uint32_t)take,
                                                 box + DEC_SMALL + DEC_STATE_PAD, (uint32_t)cap, stream), who);
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            memcpy(&h, box + DEC_SMALL, sizeof(h));
            if (h.in_used > take || h.out_made > cap || h.hist_len > LZS_MAX_HISTORY_SIZE) {
                fail(LZS_E_HIP, "%s: inconsistent state from the device", who);
                goto failed;
            }
            memcpy(p->outPtr, box + DEC_SMALL + DEC_STATE_PAD, h.out_made);
        } else {
            e = staging_reserve(st, BUF_IN, take + 64, &d_in);
            if (!e) e = staging_reserve(st, BUF_OUT, cap + 64, &d_out);
            if (!e) e = staging_reserve(st, BUF_AUX, sizeof(h), &d_state);
            if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
            HIP_TRY(lzs_hip_h2d(d_state, &h, sizeof(h), stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_h2d(d_in, p->inPtr, take, stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_launch_decode_resume((lzs_dec_resume_t *)d_state, d_in, (uint32_t)take, d_out, (uint32_t)cap, stream), who);
            HIP_TRY(lzs_hip_d2h(&h, d_state, sizeof(h), stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
            if (h.in_used > take || h.out_made > cap || h.hist_len > LZS_MAX_HISTORY_SIZE) {
                fail(LZS_E_HIP, "%s: inconsistent state from the device", who);
                goto failed;
            }
            HIP_TRY(lzs_hip_d2h(p->outPtr, d_out, h.out_made, stream), "hipMemcpy D2H");
            HIP_TRY(lzs_hip_stream_sync(stream), "hipStreamSynchronize");
        }
        pv->bitq = h.bitq; pv->qlen = (uint8_t)h.qlen; pv->off = (uint16_t)h.off; pv->rem = (uint8_t)h.rem;
        pv->extended = (uint8_t)h.extended; pv->hist_len = (uint16_t)h.hist_len;
        memcpy(pv->hist, h.hist, h.hist_len);
        p->inPtr += h.in_used;   p->inLength -= h.in_used;
        p->outPtr += h.out_made; p->outLength -= h.out_made;
        made += h.out_made;
        /* stopped only because of this loop's own limits: go on */
        if ((h.status & LZS_INC_INPUT_STARVED) && p->inLength) continue;
        if ((h.status & LZS_INC_NO_OUTPUT_SPACE) && p->outLength && cap == most) continue;
        p->status = (uint8_t)h.status;
        break;
    }
#undef HIP_TRY
    staging_trim(st);
    return made;

failed:
    inc_failed_note(who);
failed_quiet:                       /* (stream_decompress has reported by itself) */
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
    /* Terminal also for a caller that never looks at the ERROR flag (the reference's tools loop
     * until the input is used up and STARVED is reported: utils/lzs-decompress.c:82-121): the
     * input is dropped, so such a loop runs out instead of spinning on the same bytes for ever. */
    if (p->inPtr) p->inPtr += p->inLength;
    p->inLength = 0;
    p->status = LZS_D_STATUS_ERROR | LZS_D_STATUS_INPUT_STARVED | LZS_D_STATUS_INPUT_FINISHED;
    return made;
}

/* ---------------------------------------------------- incremental interface: compression */
/* reference lzs-compression.c:479-823.  Between calls the caller's block holds, in place of the
 * reference's ring and hash tables: the last bytes already encoded (what the next piece's chains
 * are built from: lzs_compress_segments_kernel warms up over the window before its first token),
 * the <= 15 bytes after them that wait for more look-ahead, the offset of a long match still
 * running, the bits of the last, partial output byte, and output that found no room.  Every call
 * encodes what the data so far decides, as one piece of the stream on the device
 * (stream_compress_piece).
 *
 * Two blocks share the code below (enc_core_t says where their members are): the reference's
 * LzsCompressParameters_t (14432 bytes: one array that holds 2304 bytes of history and then either
 * up to 10 KiB of input collected before the device is asked, or up to 11.7 KiB of output that
 * found no room) and its LzsSimpleCompressParameters_t (2112 bytes: the window, the look-ahead and
 * nine bytes of output that found no room -- every call that can decide a token goes to the
 * device, and input is only taken as far as its worst-case output fits the caller's buffer and
 * those nine bytes). */
#define INC_HIST      2304u
#define INC_UNDECIDED 15u           /* what a piece leaves undecided at most (LZS_MAX_LOOK_AHEAD_LEN) */
#define INC_BUF       14368u        /* one array for both: history + bytes not yet encoded, or history + output that found no room */
#define INC_CARRY_MAX (INC_BUF - INC_HIST)      /* room for bytes not yet encoded ... */
#define INC_ACCUM     10240u        /* ... small pieces are collected up to here before the device is asked */
#define INC_PEND_AT   (INC_HIST + INC_UNDECIDED + 1u)   /* after a device call data[] ends below here: the parked output starts here */
#define INC_PEND_MAX  (INC_BUF - INC_PEND_AT)
typedef struct __attribute__((packed)) {
    uint32_t data_len;              /* bytes in data[]: history, then carry_len bytes not yet encoded */
    uint32_t carry_len;
    uint32_t pend_pos, pend_len;    /* output waiting in data[INC_PEND_AT + pend_pos .. INC_PEND_AT + pend_len): while there is
                                     * any, no input is collected (data_len < INC_PEND_AT), so the two never meet */
    uint16_t ext_off;               /* != 0: inside a long match at this offset */
    uint8_t  bit_len, bit_val;      /* bits of the partial last output byte, left-aligned */
    uint8_t  marker_waiting;        /* the end marker is among the waiting output */
    uint8_t  data[INC_BUF];
} enc_priv_t;
#define ENC_PRIV_AT 40u
_Static_assert(ENC_PRIV_AT + sizeof(enc_priv_t) <= sizeof(LzsCompressParameters_t), "private state fits");

#define SIMPLE_PEND_MAX 9u
typedef struct __attribute__((packed)) {
    uint16_t data_len;              /* history (<= 2047), then carry_len bytes not yet encoded (<= 15) */
    uint16_t ext_off;
    uint8_t  carry_len, bit_len, bit_val;     /* bit_len: bits 0-2; bit 7: the end marker is among the waiting output */
    uint8_t  data[LZS_MAX_HISTORY_SIZE + INC_UNDECIDED];
    uint8_t  pend_at;               /* output waiting in pend[pend_at & 15 .. pend_at >> 4) */
    uint8_t  pend[SIMPLE_PEND_MAX];
} simple_priv_t;
#define SIMPLE_PRIV_AT 33u
_Static_assert(sizeof(LzsSimpleCompressParameters_t) == 2112, "size of the reference's LzsSimpleCompressParameters_t");
_Static_assert(SIMPLE_PRIV_AT + sizeof(simple_priv_t) <= sizeof(LzsSimpleCompressParameters_t), "private state fits");

/* one parameter block as the shared code sees it */
typedef struct {
    const uint8_t **inPtr; uint8_t **outPtr; size_t *inLength, *outLength; uint8_t *status;
    uint8_t *data; uint32_t hist_keep, accum;     /* history kept between calls; input collected up to here */
    uint8_t *pend; uint32_t pend_cap;             /* output that found no room (NULL, 0: there is none) */
    uint32_t data_len, carry_len, pend_pos, pend_len, ext_off, bit_len, bit_val, marker_waiting;
} enc_core_t;

void lzs_compress_init_full(LzsCompressParameters_t *p)
{
    inc_noted[0] = 0;               /* a new stream: its first failure is reported again */
    if (!p) return;
    p->status = LZS_C_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
}

void lzs_compress_init_quick(LzsCompressParameters_t *p) { lzs_compress_init_full(p); }

void lzs_simple_compress_init(LzsSimpleCompressParameters_t *p)
{
    inc_noted[0] = 0;
    if (!p) return;
    p->status = LZS_C_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
}

/* hand `len` bytes to the caller's buffer, what does not fit to pend[] (room was reserved) */
static size_t inc_deliver(enc_core_t *s, const uint8_t *src, size_t len)
{
    const size_t now = len < *s->outLength ? len : *s->outLength;
    memcpy(*s->outPtr, src, now);
    *s->outPtr += now; *s->outLength -= now;
    if (len > now) memcpy(s->pend, src + now, len - now);
    s->pend_pos = 0; s->pend_len = (uint32_t)(len - now);
    return now;
}

static size_t inc_compress_core(enc_core_t *s, bool add_end_marker, const char *who)
{
    size_t made = 0;
    uint8_t *tmp = NULL;
    *s->status = LZS_C_STATUS_NONE;
    tls_error[0] = 0;
    if ((*s->inLength && !*s->inPtr) || (*s->outLength && !*s->outPtr) ||
        s->data_len > s->hist_keep + (s->accum > INC_UNDECIDED ? INC_CARRY_MAX : INC_UNDECIDED) || s->carry_len > s->data_len ||
        s->pend_len > s->pend_cap || s->pend_pos > s->pend_len) {
        fail(LZS_E_ARG, "%s: NULL buffer or a parameter block that was not initialised", who);
        goto failed;                /* terminal like every other failure: a loop that never looks at ERROR ends */
    }
    /* no device, no stream: say so at the first call, not when the collected input is flushed */
    if (lzs_env()->route != LZS_ROUTE_HOST && require_device() != LZS_OK) goto failed;
    /* output still waiting from the call before goes first (:574-588) */
    if (s->pend_pos < s->pend_len) {
        const size_t have = s->pend_len - s->pend_pos;
        const size_t now = have < *s->outLength ? have : *s->outLength;
        memcpy(*s->outPtr, s->pend + s->pend_pos, now);
        *s->outPtr += now; *s->outLength -= now; s->pend_pos += (uint32_t)now; made += now;
        if (s->pend_pos < s->pend_len) {
            *s->status = LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE;
            return made;
        }
        s->pend_pos = s->pend_len = 0;
        if (s->marker_waiting) {
            s->marker_waiting = 0;
            *s->status = LZS_C_STATUS_END_MARKER | (*s->inLength ? 0 : LZS_C_STATUS_INPUT_FINISHED | LZS_C_STATUS_INPUT_STARVED);
            return made;
        }
    }
    int starved_for_room = 0;
    for (;;) {
        /* Take as much input as the room for its output allows: 9 bits a byte at worst (what is
         * waiting undecided included), 8 bytes for the partial byte and the end marker; into the
         * caller's buffer and then into pend[]; one piece is at most 1 GiB. */
        const size_t room = (*s->outLength < ((size_t)1 << 40) ? *s->outLength : ((size_t)1 << 40)) + s->pend_cap;
        /* (at most 9 bits per byte a token covers -- a literal; <= 7 bits carried in, the end marker's 9 and <= 7 of padding) */
        const size_t worst = 8u * room >= 27u ? (8u * room - 27u) / 9u : 0u;       /* input bytes whose output surely fits */
        size_t take;
        int last;
        uint32_t stop = 0;
        const uint32_t c0 = s->data_len - s->carry_len;
        if (worst > s->carry_len) {
            const size_t fits = worst - s->carry_len;
            take = *s->inLength < fits ? *s->inLength : fits;
            if (take > ((size_t)1 << 30)) take = (size_t)1 << 30;
            last = add_end_marker && take == *s->inLength;
        } else {
            /* Little room (the low-memory block parks 9 bytes; lzs-compression-simple.c:435-647 goes on with
             * outLength >= 1, and so must this -- VERDICT r03): nine bits per BYTE promise nothing here, but
             * the tokens a piece decides can be counted.  k token starts cover k - 1 bytes before the last of
             * them begins (<= 9 bits a byte: a literal), and that last one is <= 17 bits and a nibble per 15
             * bytes of the <= k + 15 there are; 7 bits may be carried in.  So the piece takes just the input
             * that makes K starts decidable (what stays undecided still fits the block), and when the stream
             * is to be finished but its marker and padding would not fit as well, the last bytes go out in
             * pieces that know where the data ends but start no token past `stop`. */
            size_t K = 0;
            /* (K + 1 tried: 9 bits for each of the K starts before the last, 17 for that one, and the nibbles of a match
             * that runs in from the piece before as well as of one that runs out) */
            while (7u + 9u * K + 17u + 8u * ((K + 1u + INC_UNDECIDED) / 15u + 1u) <= 8u * room) K++;
            if (K == 0) { starved_for_room = 1; break; }            /* (room < 4 bytes: not with a pend[] of 9) */
            const size_t want = K + INC_UNDECIDED - s->carry_len;  /* >= K >= 1 */
            take = *s->inLength < want ? *s->inLength : want;
            const size_t span = s->carry_len + take;
            last = add_end_marker && take == *s->inLength;
            if (last && 7u + 9u * span + 16u > 8u * room) {         /* all of it AND the marker: too much for one call */
                last = 0;
                stop = c0 + (uint32_t)(K < span ? K : span);
                starved_for_room = 1;                               /* (more output is waiting for room: say so) */
            }
        }
        const size_t n = (size_t)s->data_len + take;
        if (!last && !stop && n - c0 <= s->accum) {
            /* Too little to decide the next token (:641-647) -- or just little: a call costs
             * ~0.12 ms whatever its size, so pieces like the reference tools' 512 bytes are
             * collected in the block until there are `accum` bytes of them -- up to 10 KiB, what the block's one array
             * holds beside the history -- or the stream is finished. */
            memcpy(s->data + s->data_len, *s->inPtr, take);
            s->data_len += (uint32_t)take; s->carry_len += (uint32_t)take;
            *s->inPtr += take; *s->inLength -= take;
            break;
        }
        piece_t pc;
        memset(&pc, 0, sizeof(pc));
        pc.prefix = s->data; pc.prefix_len = s->data_len;
        pc.c0 = c0; pc.ext_off = s->ext_off; pc.bit0 = s->bit_len; pc.first = (uint8_t)s->bit_val; pc.last = last; pc.stop = stop;
        const size_t cap = LZS_COMPRESSED_MAX(n - c0) + 16;
        tmp = (uint8_t *)malloc(cap);
        if (!tmp) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
        int rc = LZS_OK;
        /* a piece with little to decide: the host route (lzs_hostcodec.c), same contract */
        size_t got;
        if (route_on_host(n - c0, INC_ENC_HOST_MAX)) {
            got = hostcodec_compress_piece(tmp, cap, *s->inPtr, n, &pc);
            if (got == SIZE_MAX) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
        } else {
            got = stream_compress_piece(tmp, cap, *s->inPtr, n, 0, &rc, &pc);
            if (rc != LZS_OK) goto failed_quiet;
        }
        const size_t whole = last ? got : (size_t)(pc.nbits / 8);
        if (whole > got || whole > room || pc.c_exit > n || pc.c_exit < c0 || (last ? pc.c_exit != n : n - pc.c_exit > INC_UNDECIDED)) {
            fail(LZS_E_HIP, "%s: inconsistent state from the device (piece of %zu bytes from %u: %zu bytes out, %llu bits, ends at %u, room %zu)",
                 who, n, c0, got, (unsigned long long)pc.nbits, pc.c_exit, room);
            goto failed;
        }
        /* the new history and carry: bytes [c_exit - hist_keep, n) of prefix + input -- first, because
         * output that finds no room is parked in the same array, above where this ends */
        {
            const size_t from = pc.c_exit > s->hist_keep ? pc.c_exit - s->hist_keep : 0;
            uint8_t keep[INC_HIST + INC_UNDECIDED + 1u];
            size_t k = 0;
            for (size_t i = from; i < n; ) {
                if (i < s->data_len) { const size_t m = (s->data_len < n ? s->data_len : n) - i; memcpy(keep + k, s->data + i, m); k += m; i += m; }
                else { const size_t m = n - i; memcpy(keep + k, *s->inPtr + (i - s->data_len), m); k += m; i += m; }
            }
            memcpy(s->data, keep, k);
            s->data_len = (uint32_t)k;
            s->carry_len = (uint32_t)(n - pc.c_exit);
        }
        made += inc_deliver(s, tmp, whole);
        s->bit_len = last ? 0 : (uint32_t)(pc.nbits & 7u);
        s->bit_val = s->bit_len ? (uint32_t)(tmp[whole] & (0xFF00u >> s->bit_len)) : 0;
        s->ext_off = pc.ext_exit;
        free(tmp); tmp = NULL;
        *s->inPtr += take; *s->inLength -= take;
        if (last) {
            if (s->pend_len) s->marker_waiting = 1;
            else *s->status |= LZS_C_STATUS_END_MARKER;
            break;
        }
        if (s->pend_len || *s->inLength == 0 || stop) break;
    }
    if (s->pend_len || starved_for_room) *s->status |= LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE;
    if (*s->inLength == 0) *s->status |= LZS_C_STATUS_INPUT_FINISHED | LZS_C_STATUS_INPUT_STARVED;
    return made;

failed:
    inc_failed_note(who);
failed_quiet:
    free(tmp);
    /* Terminal also for a caller that never looks at the ERROR flag (the reference's tool loops
     * until END_MARKER: utils/lzs-compress.c:91-134): the input is dropped and END_MARKER set
     * next to ERROR, so such a loop ends instead of spinning on the same bytes for ever. */
    if (*s->inPtr) *s->inPtr += *s->inLength;
    *s->inLength = 0;
    *s->status = LZS_C_STATUS_ERROR | LZS_C_STATUS_END_MARKER | LZS_C_STATUS_INPUT_STARVED | LZS_C_STATUS_INPUT_FINISHED;
    return made;
}

size_t lzs_compress_incremental(LzsCompressParameters_t *p, bool add_end_marker)
{
    if (!p) return 0;
    enc_priv_t *pv = (enc_priv_t *)((uint8_t *)p + ENC_PRIV_AT);
    enc_core_t s = { &p->inPtr, &p->outPtr, &p->inLength, &p->outLength, &p->status,
                     pv->data, INC_HIST, INC_ACCUM, pv->data + INC_PEND_AT, INC_PEND_MAX,
                     pv->data_len, pv->carry_len, pv->pend_pos, pv->pend_len, pv->ext_off, pv->bit_len, pv->bit_val, pv->marker_waiting };
    const size_t made = inc_compress_core(&s, add_end_marker, "lzs_compress_incremental");
    pv->data_len = s.data_len; pv->carry_len = s.carry_len; pv->pend_pos = s.pend_pos; pv->pend_len = s.pend_len;
    pv->ext_off = (uint16_t)s.ext_off; pv->bit_len = (uint8_t)s.bit_len; pv->bit_val = (uint8_t)s.bit_val;
    pv->marker_waiting = (uint8_t)s.marker_waiting;
    return made;
}

/* reference lzs-compression-simple.c: the low-memory compressor with the same output
 * (lzs.h:224-227).  Same calls, same stream; the 2112-byte block has no room to collect input or
 * to park output, so every call with >= 16 bytes to decide is a device call, and a call only takes
 * the input whose worst-case output fits outLength and the block's nine spare bytes.  Any outLength
 * >= 1 makes progress, like the reference's (lzs-compression-simple.c:435-647): with little room a
 * call takes just the input that makes a few token starts decidable (their worst case is counted by
 * tokens, not by bytes: DESIGN.md 3.7), and the final flush goes out in pieces that stop early. */
size_t lzs_simple_compress_incremental(LzsSimpleCompressParameters_t *p, bool add_end_marker)
{
    if (!p) return 0;
    simple_priv_t *pv = (simple_priv_t *)((uint8_t *)p + SIMPLE_PRIV_AT);
    enc_core_t s = { &p->inPtr, &p->outPtr, &p->inLength, &p->outLength, &p->status,
                     pv->data, LZS_MAX_HISTORY_SIZE, INC_UNDECIDED, pv->pend, SIMPLE_PEND_MAX,
                     pv->data_len, pv->carry_len, pv->pend_at & 15u, pv->pend_at >> 4, pv->ext_off, pv->bit_len & 7u, pv->bit_val,
                     pv->bit_len >> 7 };
    const size_t made = inc_compress_core(&s, add_end_marker, "lzs_simple_compress_incremental");
    pv->data_len = (uint16_t)s.data_len; pv->carry_len = (uint8_t)s.carry_len;
    pv->ext_off = (uint16_t)s.ext_off; pv->bit_len = (uint8_t)(s.bit_len | (s.marker_waiting << 7)); pv->bit_val = (uint8_t)s.bit_val;
    pv->pend_at = (uint8_t)(s.pend_pos | (s.pend_len << 4));
    return made;
}

size_t lzs_simple_compress(uint8_t *a_pOutData, size_t a_outBufferSize, const uint8_t *a_pInData, size_t a_inLen)
{
    return lzs_compress(a_pOutData, a_outBufferSize, a_pInData, a_inLen);
}
