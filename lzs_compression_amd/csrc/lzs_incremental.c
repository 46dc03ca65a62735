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

void lzs_decompress_init(LzsDecompressParameters_t *p)
{
    if (!p) return;
    p->status = LZS_D_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
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
    if ((p->inLength && !p->inPtr) || (p->outLength && !p->outPtr)) {
        fail(LZS_E_ARG, "%s: NULL buffer", who);
        p->status = LZS_D_STATUS_ERROR;
        return 0;
    }
    if (require_device() != LZS_OK) goto failed;
    staging_t *st = staging_get();
    if (!st) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
#define HIP_TRY(call, what) do { e = (call); if (e) { hip_fail(e, what); goto failed; } } while (0)
    if (!st->stream) HIP_TRY(lzs_hip_stream_create(&st->stream), "hipStreamCreate");
    void *stream = st->stream;
    lzs_dec_resume_t h;
    memset(&h, 0, sizeof(h));
    size_t wave_limit = (size_t)16 << 20;
    for (;;) {
        /* A large piece first goes to many wavefronts (stream_decompress) as far as
         * whole segments can be decoded; what is left -- the segment with the end marker, the
         * unfinished token at the end of the input, the last bytes before the output is full, a
         * copy still running -- is the one wavefront's below. */
        if (pv->rem == 0 && p->inLength >= INC_DEC_STREAM_MIN && p->outLength >= 4096u && !getenv("LZS_ONE_WAVE")) {
            const size_t big = p->inLength < ((size_t)256 << 20) ? p->inLength : ((size_t)256 << 20);
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
            if (rc != LZS_OK) { p->status = LZS_D_STATUS_ERROR; return made; }
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
                p->inPtr += used;  p->inLength -= used;
                p->outPtr += got;  p->outLength -= got;
                made += got;
                if (p->inLength == 0 && pv->qlen == 0) { p->status = LZS_D_STATUS_INPUT_FINISHED | LZS_D_STATUS_INPUT_STARVED; break; }
            }
            /* not even one whole segment this time: the wavefront takes the next stretch */
            wave_limit = 4u * (size_t)(dp.seg ? dp.seg : 8192u);
        }
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
            if (!st->host_box) st->host_box = (uint8_t *)malloc(INC_BOX_BYTES);
            uint8_t *box = st->host_box;
            if (!box) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
            uint8_t *d_box = NULL;
            e = staging_reserve(st, BUF_AUX, INC_BOX_BYTES, (void **)&d_box);
            if (e) { fail(LZS_E_NOMEM, "%s: device allocation failed: %s", who, lzs_hip_strerror(e)); goto failed; }
            const size_t in_at = DEC_SMALL - ((take + 3u) & ~(size_t)3u);
            memcpy(box + in_at, p->inPtr, take);
            memcpy(box + DEC_SMALL, &h, sizeof(h));
            HIP_TRY(lzs_hip_h2d(d_box + in_at, box + in_at, DEC_SMALL - in_at + sizeof(h), stream), "hipMemcpy H2D");
            HIP_TRY(lzs_hip_launch_decode_resume((lzs_dec_resume_t *)(d_box + DEC_SMALL), d_box + in_at, (uint32_t)take,
                                                 d_box + DEC_SMALL + DEC_STATE_PAD, (uint32_t)cap, stream), who);
            HIP_TRY(lzs_hip_d2h(box + DEC_SMALL, d_box + DEC_SMALL, DEC_STATE_PAD + cap, stream), "hipMemcpy D2H");
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
    p->status = LZS_D_STATUS_ERROR;
    fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
    { staging_t *s2 = staging_get(); if (s2 && s2->stream) lzs_hip_stream_sync(s2->stream); }
    return made;
}

/* ---------------------------------------------------- incremental interface: compression */
/* reference lzs-compression.c:479-823.  Between calls the caller's block holds, in place of the
 * reference's ring and hash tables: the last INC_HIST bytes already encoded (what the next
 * piece's chains are built from: lzs_compress_segments_kernel warms up over 2176 + 64 bytes
 * before its first token), the <= 15 bytes after them that wait for more look-ahead, the offset
 * of a long match still running, the bits of the last, partial output byte, and output that
 * found no room.  Every call encodes what the data so far decides, as one piece of the stream
 * on the device (stream_compress_piece). */
#define INC_HIST      2304u
#define INC_CARRY_MAX 3600u         /* room for bytes not yet encoded ... */
#define INC_ACCUM     3072u         /* ... small pieces are collected up to here before the device is asked */
#define INC_UNDECIDED 15u           /* what a piece leaves undecided at most (LZS_MAX_LOOK_AHEAD_LEN) */
#define INC_PEND_MAX  8192u
typedef struct __attribute__((packed)) {
    uint32_t data_len;              /* bytes in data[]: history, then carry_len bytes not yet encoded */
    uint32_t carry_len;
    uint32_t pend_pos, pend_len;    /* output waiting in pend[pend_pos .. pend_len) */
    uint16_t ext_off;               /* != 0: inside a long match at this offset */
    uint8_t  bit_len, bit_val;      /* bits of the partial last output byte, left-aligned */
    uint8_t  marker_waiting;        /* the end marker is among the waiting output */
    uint8_t  data[INC_HIST + INC_CARRY_MAX];
    uint8_t  pend[INC_PEND_MAX];
} enc_priv_t;
#define ENC_PRIV_AT 40u
_Static_assert(ENC_PRIV_AT + sizeof(enc_priv_t) <= sizeof(LzsCompressParameters_t), "private state fits");

void lzs_compress_init_full(LzsCompressParameters_t *p)
{
    if (!p) return;
    p->status = LZS_C_STATUS_NONE;
    memset(p->reserved_, 0, sizeof(p->reserved_));
}

void lzs_compress_init_quick(LzsCompressParameters_t *p) { lzs_compress_init_full(p); }

/* hand `len` bytes to the caller's buffer, what does not fit to pend[] (room was reserved) */
static size_t inc_deliver(LzsCompressParameters_t *p, enc_priv_t *pv, const uint8_t *src, size_t len)
{
    const size_t now = len < p->outLength ? len : p->outLength;
    memcpy(p->outPtr, src, now);
    p->outPtr += now; p->outLength -= now;
    memcpy(pv->pend, src + now, len - now);
    pv->pend_pos = 0; pv->pend_len = (uint32_t)(len - now);
    return now;
}

size_t lzs_compress_incremental(LzsCompressParameters_t *p, bool add_end_marker)
{
    const char *who = "lzs_compress_incremental";
    if (!p) return 0;
    enc_priv_t *pv = (enc_priv_t *)((uint8_t *)p + ENC_PRIV_AT);
    size_t made = 0;
    uint8_t *tmp = NULL;
    p->status = LZS_C_STATUS_NONE;
    tls_error[0] = 0;
    if ((p->inLength && !p->inPtr) || (p->outLength && !p->outPtr) ||
        pv->data_len > sizeof(pv->data) || pv->carry_len > pv->data_len || pv->pend_len > INC_PEND_MAX || pv->pend_pos > pv->pend_len) {
        fail(LZS_E_ARG, "%s: NULL buffer or a parameter block that was not initialised", who);
        p->status = LZS_C_STATUS_ERROR;
        return 0;
    }
    /* no device, no stream: say so at the first call, not when the collected input is flushed */
    if (require_device() != LZS_OK) {
        fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
        p->status = LZS_C_STATUS_ERROR;
        return 0;
    }
    /* output still waiting from the call before goes first (:574-588) */
    if (pv->pend_pos < pv->pend_len) {
        const size_t have = pv->pend_len - pv->pend_pos;
        const size_t now = have < p->outLength ? have : p->outLength;
        memcpy(p->outPtr, pv->pend + pv->pend_pos, now);
        p->outPtr += now; p->outLength -= now; pv->pend_pos += (uint32_t)now; made += now;
        if (pv->pend_pos < pv->pend_len) {
            p->status = LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE;
            return made;
        }
        pv->pend_pos = pv->pend_len = 0;
        if (pv->marker_waiting) {
            pv->marker_waiting = 0;
            p->status = LZS_C_STATUS_END_MARKER | (p->inLength ? 0 : LZS_C_STATUS_INPUT_FINISHED | LZS_C_STATUS_INPUT_STARVED);
            return made;
        }
    }
    for (;;) {
        /* Take as much input as the room for its output allows: 9 bits a byte at worst, into the
         * caller's buffer and then into pend[]; one piece is at most 1 GiB. */
        const size_t room = (p->outLength < ((size_t)1 << 40) ? p->outLength : ((size_t)1 << 40)) + INC_PEND_MAX;
        const size_t fits = (8u * room - 64u) / 9u - pv->carry_len;
        size_t take = p->inLength < fits ? p->inLength : fits;
        if (take > ((size_t)1 << 30)) take = (size_t)1 << 30;
        const int last = add_end_marker && take == p->inLength;
        const size_t n = (size_t)pv->data_len + take;
        const uint32_t c0 = pv->data_len - pv->carry_len;
        if (!last && n - c0 <= INC_ACCUM) {
            /* Too little to decide the next token (:641-647) -- or just little: a call costs
             * ~0.12 ms whatever its size, so pieces like the reference tools' 512 bytes are
             * collected in the block until there are 3 KiB of them (or the stream is finished). */
            memcpy(pv->data + pv->data_len, p->inPtr, take);
            pv->data_len += (uint32_t)take; pv->carry_len += (uint32_t)take;
            p->inPtr += take; p->inLength -= take;
            break;
        }
        piece_t pc;
        memset(&pc, 0, sizeof(pc));
        pc.prefix = pv->data; pc.prefix_len = pv->data_len;
        pc.c0 = c0; pc.ext_off = pv->ext_off; pc.bit0 = pv->bit_len; pc.first = pv->bit_val; pc.last = last;
        const size_t cap = LZS_COMPRESSED_MAX(n - c0) + 16;
        tmp = (uint8_t *)malloc(cap);
        if (!tmp) { fail(LZS_E_NOMEM, "%s: out of host memory", who); goto failed; }
        int rc = LZS_OK;
        const size_t got = stream_compress_piece(tmp, cap, p->inPtr, n, 0, &rc, &pc);
        if (rc != LZS_OK) goto failed_quiet;
        const size_t whole = last ? got : (size_t)(pc.nbits / 8);
        if (whole > got || whole > room || pc.c_exit > n || pc.c_exit < c0 || (last ? pc.c_exit != n : n - pc.c_exit > INC_UNDECIDED)) {
            fail(LZS_E_HIP, "%s: inconsistent state from the device (piece of %zu bytes from %u: %zu bytes out, %llu bits, ends at %u, room %zu)",
                 who, n, c0, got, (unsigned long long)pc.nbits, pc.c_exit, room);
            goto failed;
        }
        made += inc_deliver(p, pv, tmp, whole);
        pv->bit_len = last ? 0 : (uint8_t)(pc.nbits & 7u);
        pv->bit_val = pv->bit_len ? (uint8_t)(tmp[whole] & (0xFF00u >> pv->bit_len)) : 0;
        pv->ext_off = (uint16_t)pc.ext_exit;
        free(tmp); tmp = NULL;
        /* the new history and carry: bytes [c_exit - INC_HIST, n) of prefix + input */
        {
            const size_t from = pc.c_exit > INC_HIST ? pc.c_exit - INC_HIST : 0;
            uint8_t keep[INC_HIST + INC_UNDECIDED + 1u];
            size_t k = 0;
            for (size_t i = from; i < n; ) {
                if (i < pv->data_len) { const size_t m = (pv->data_len < n ? pv->data_len : n) - i; memcpy(keep + k, pv->data + i, m); k += m; i += m; }
                else { const size_t m = n - i; memcpy(keep + k, p->inPtr + (i - pv->data_len), m); k += m; i += m; }
            }
            memcpy(pv->data, keep, k);
            pv->data_len = (uint32_t)k;
            pv->carry_len = (uint32_t)(n - pc.c_exit);
        }
        p->inPtr += take; p->inLength -= take;
        if (last) {
            if (pv->pend_len) pv->marker_waiting = 1;
            else p->status |= LZS_C_STATUS_END_MARKER;
            break;
        }
        if (pv->pend_len || p->inLength == 0) break;
    }
    if (pv->pend_len) p->status |= LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE;
    if (p->inLength == 0) p->status |= LZS_C_STATUS_INPUT_FINISHED | LZS_C_STATUS_INPUT_STARVED;
    return made;

failed:
    fprintf(stderr, "liblzs: %s failed: %s\n", who, tls_error);
failed_quiet:
    free(tmp);
    p->status = LZS_C_STATUS_ERROR;
    return made;
}

