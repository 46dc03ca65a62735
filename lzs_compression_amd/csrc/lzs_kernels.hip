// lzs_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the LZS one-shot path,
// plus the extern-"C" shim the C host library calls.
//
// Path and contract (what must come out, bit for bit):
//   reference c/src/liblzs/lzs-compression.c:249-467   lzs_compress()
//   reference c/src/liblzs/lzs-decompression.c:156-412 lzs_decompress()
// The encoder decision rule is SURVEY.md Appendix A.2: at each token start c, take the
// NEAREST offset in 1..min(c,2047) that maximises min(common_prefix, min(remaining,12));
// a match whose first length code is 8 is then extended at that same offset in nibbles
// of up to 15 bytes.  The search is a pure function of (input, c), which is what makes
// the lane-parallel scan below legal.
//
// Execution model: ONE 64-lane wavefront per independent block (4 waves per 256-thread
// workgroup, no inter-wave communication, no barriers).  Per wave, in LDS:
//   ring[4096]  the input's sliding window: position p lives at ring[p & 4095]; it always
//               holds [c-2047, c+64) -- history plus look-ahead -- and is refilled 1 KiB at
//               a time by coalesced 16-byte-per-lane loads from HBM;
//   stage[256]  output staging, drained to HBM as one coalesced 4-byte-per-lane store.
// No MFMA: this is byte search and bit packing, not a contraction.
//
// gfx950 only.  No CUDA compatibility layer, no alternate code paths.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "lzs_hip_shim.h"

namespace {

constexpr uint32_t kWindow     = 2047;   // farthest offset (11 bits)      lzs.h:60
constexpr uint32_t kSearchCap  = 12;     // search stops improving here    lzs-compression.c:62
constexpr uint32_t kTokenMax   = 8;      // first length code carries <=8  lzs-common.h:52
constexpr uint32_t kNibbleMax  = 15;     // extension nibble "continue"    lzs-common.h:53
constexpr uint32_t kShortMax   = 127;    // 7-bit offsets                  lzs-common.h:43

constexpr uint32_t kRing       = 4096;   // bytes, power of two >= window + tile + look-ahead
constexpr uint32_t kRingMask   = kRing - 1;
constexpr uint32_t kRingWords  = kRing / 4;
constexpr uint32_t kTile       = 1024;   // 64 lanes x 16 B per refill
constexpr uint32_t kLookAhead  = 64;     // bytes past c guaranteed resident (>= 15)
constexpr uint32_t kStage      = 256;    // output staging bytes per wave
constexpr uint32_t kWavesPerWG = 4;

struct __attribute__((aligned(16))) WaveLds {
    uint32_t ring[kRingWords];
    uint32_t stage[kStage / 4];
};

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
        v = v > o ? v : o;
    }
    return v;
}

// 16 input bytes at block position p (multiple of 16), zero past n.
__device__ __forceinline__ uint4 load16(const uint8_t *src, uint32_t p, uint32_t n, bool aligned16, bool aligned4 = false)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p + 16 <= n && aligned16) {
        v = *reinterpret_cast<const uint4 *>(src + p);
    } else if (p + 16 <= n && aligned4) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(src + p);
        v = make_uint4(w[0], w[1], w[2], w[3]);
    } else if (p < n) {
        uint32_t w[4] = {0, 0, 0, 0};
        uint32_t m = n - p < 16 ? n - p : 16;
        for (uint32_t k = 0; k < m; k++)
            w[k >> 2] |= (uint32_t)src[p + k] << (8 * (k & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

// 12 bytes of the ring starting at block position q (any alignment), little-endian words.
__device__ __forceinline__ void ring_read12(const uint32_t *ring, uint32_t q,
                                            uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
    const uint32_t a = (q & kRingMask) >> 2;
    const uint32_t s = q & 3;
    const uint32_t d0 = ring[a];
    const uint32_t d1 = ring[(a + 1) & (kRingWords - 1)];
    const uint32_t d2 = ring[(a + 2) & (kRingWords - 1)];
    const uint32_t d3 = ring[(a + 3) & (kRingWords - 1)];
    w0 = __builtin_amdgcn_alignbyte(d1, d0, s);
    w1 = __builtin_amdgcn_alignbyte(d2, d1, s);
    w2 = __builtin_amdgcn_alignbyte(d3, d2, s);
}

__device__ __forceinline__ uint32_t ring_byte(const uint32_t *ring, uint32_t q)
{
    return reinterpret_cast<const uint8_t *>(ring)[q & kRingMask];
}

// Equal leading bytes (0..4) given the XOR of two little-endian words.
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x)
{
    const uint32_t t = ((uint32_t)__builtin_ffs((int)x) - 1u) >> 3;   // x==0 -> huge
    return t < 4u ? t : 4u;
}

// ---------------------------------------------------------------------------------
// Output bit sink: MSB-first bits -> big-endian words in LDS -> coalesced HBM stores.
// All state is wave-uniform.  Truncation rule of lzs-compression.c:304-313: bytes at
// or past `cap` are dropped, nothing past the buffer is touched.
// ---------------------------------------------------------------------------------
struct Sink {
    uint64_t acc;       // pending bits, right-aligned
    uint32_t nbits;     // < 32 between calls
    uint32_t fill;      // bytes in stage[]
    uint32_t flushed;   // bytes already handed to HBM (multiple of kStage)
    uint8_t *dst;
    uint32_t cap;
    bool     aligned4;
};

__device__ __forceinline__ void sink_flush_full(Sink &s, uint32_t *stage, uint32_t lane)
{
    __builtin_amdgcn_wave_barrier();
    const uint32_t v = stage[lane];
    const uint32_t at = s.flushed + 4 * lane;
    if (s.aligned4 && at + 4 <= s.cap) {
        *reinterpret_cast<uint32_t *>(s.dst + at) = v;
    } else {
        for (uint32_t k = 0; k < 4; k++)
            if (at + k < s.cap) s.dst[at + k] = (uint8_t)(v >> (8 * k));
    }
    __builtin_amdgcn_wave_barrier();
    s.flushed += kStage;
    s.fill = 0;
}

__device__ __forceinline__ void sink_put(Sink &s, uint32_t *stage, uint32_t lane, uint32_t value, uint32_t width)
{
    s.acc = (s.acc << width) | value;
    s.nbits += width;
    if (s.nbits >= 32) {
        s.nbits -= 32;
        const uint32_t word = (uint32_t)(s.acc >> s.nbits);
        if (lane == 0) stage[s.fill >> 2] = __builtin_bswap32(word);
        s.fill += 4;
        if (s.fill == kStage) sink_flush_full(s, stage, lane);
    }
}

// End marker 1 1 0000000, zero pad to a byte, drain (lzs-compression.c:449-466); the
// returned length is cut at the capacity like every byte before it.
__device__ __forceinline__ void sink_finish(Sink &s, uint32_t *stage, uint32_t lane, uint32_t *len_out)
{
    sink_put(s, stage, lane, 0x180u, 9);
    if (s.nbits & 7u) sink_put(s, stage, lane, 0u, 8u - (s.nbits & 7u));
    uint8_t *stage8 = reinterpret_cast<uint8_t *>(stage);
    const uint32_t tail = s.nbits >> 3;                 // 0..3 whole bytes left in acc
    if (lane < tail) stage8[s.fill + lane] = (uint8_t)(s.acc >> (s.nbits - 8 - 8 * lane));
    s.fill += tail;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < s.fill; i += 64) {
        const uint32_t at = s.flushed + i;
        if (at < s.cap) s.dst[at] = stage8[i];
    }
    const uint32_t total = s.flushed + s.fill;
    if (lane == 0) *len_out = total < s.cap ? total : s.cap;
}

// ---------------------------------------------------------------------------------
// lzs_compress() per block, variant "scan": every token start scans all offsets, 64 per
// round, nearest first.  Simple and data-independent; kept as the A/B baseline.
// reference lzs-compression.c:249-467
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerWG * 64)
void lzs_compress_blocks_scan_kernel(uint8_t *__restrict__ out, size_t out_stride, uint32_t out_cap,
                                uint32_t *__restrict__ out_len,
                                const uint8_t *__restrict__ in, size_t in_stride,
                                const uint32_t *__restrict__ in_len, uint32_t in_len_uniform,
                                uint32_t nblocks)
{
    __shared__ WaveLds lds[kWavesPerWG];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv   = uniform(threadIdx.x >> 6);   // wave-uniform, and the compiler knows it
    const uint32_t b    = blockIdx.x * kWavesPerWG + wv;
    if (b >= nblocks) return;

    WaveLds &L = lds[wv];
    const uint8_t *src = in + (size_t)b * in_stride;
    const uint32_t n   = in_len ? in_len[b] : in_len_uniform;
    const bool src16   = ((uintptr_t)src & 15u) == 0;

    Sink s;
    s.acc = 0; s.nbits = 0; s.fill = 0; s.flushed = 0;
    s.dst = out + (size_t)b * out_stride;
    s.cap = out_cap;
    s.aligned4 = ((uintptr_t)s.dst & 3u) == 0;

    uint32_t c = 0;        // start of the next token
    uint32_t loaded = 0;   // ring holds [loaded-4096, loaded); multiple of kTile

    while (c < n) {
        // Keep [c, c+kLookAhead) resident.  (lzs-compression.c works on the caller's flat
        // buffer; the ring is our LDS image of it.)
        while (loaded < n && loaded < c + kLookAhead) {
            const uint32_t p = loaded + 16 * lane;
            const uint4 v = load16(src, p, n, src16);
            *reinterpret_cast<uint4 *>(&L.ring[(p & kRingMask) >> 2]) = v;
            loaded += kTile;
        }
        __builtin_amdgcn_wave_barrier();
        if (s.flushed >= s.cap) break;                 // output full: lzs-compression.c:306-309

        // ---- search: lzs-compression.c:322-363 (== lzs-compression-simple.c:264-278)
        const uint32_t remaining = n - c;
        const uint32_t lim = remaining < kSearchCap ? remaining : kSearchCap;
        uint32_t best = 0;                             // (len << 11) | (2047 - off)
        if (lim >= 2) {
            uint32_t t0, t1, t2;
            ring_read12(L.ring, c, t0, t1, t2);
            const uint32_t reach = c < kWindow ? c : kWindow;
            for (uint32_t base = 0; base < reach; base += 64) {
                const uint32_t off = base + lane + 1;          // nearest offsets first
                const bool valid = off <= reach;
                uint32_t w0, w1, w2;
                ring_read12(L.ring, c - (valid ? off : 1u), w0, w1, w2);
                const uint32_t e0 = eq_bytes(w0 ^ t0);
                const uint32_t e1 = eq_bytes(w1 ^ t1);
                const uint32_t e2 = eq_bytes(w2 ^ t2);
                uint32_t len = e0 + (e0 == 4 ? e1 + (e1 == 4 ? e2 : 0u) : 0u);
                len = len < lim ? len : lim;
                const uint32_t key = (valid && len >= 2) ? ((len << 11) | (kWindow - off)) : 0u;
                best = best > key ? best : key;
                // a candidate at the cap in this round beats everything farther away (:341-344)
                if (__ballot(valid && len == lim) != 0ull) break;
            }
            best = wave_max(best);
        }
        best = uniform(best);

        const uint32_t len = best >> 11;
        if (len < 2) {
            // ---- literal: 0 bbbbbbbb (:365-375)
            sink_put(s, L.stage, lane, ring_byte(L.ring, c) & 0xFFu, 9);
            c += 1;
            continue;
        }
        // ---- match head: 1, offset, first length code (:376-409)
        const uint32_t off = kWindow - (best & kWindow);
        const uint32_t first = len < kTokenMax ? len : kTokenMax;
        if (off <= kShortMax) sink_put(s, L.stage, lane, (3u << 7) | off, 9);
        else                  sink_put(s, L.stage, lane, (2u << 11) | off, 13);
        if (first <= 4) sink_put(s, L.stage, lane, first - 2, 2);
        else            sink_put(s, L.stage, lane, 0xCu + (first - 5), 4);
        c += first;
        if (first == kTokenMax) {
            // ---- extension nibbles at the same offset (:417-431)
            uint32_t e;
            do {
                while (loaded < n && loaded < c + kLookAhead) {
                    const uint32_t p = loaded + 16 * lane;
                    const uint4 v = load16(src, p, n, src16);
                    *reinterpret_cast<uint4 *>(&L.ring[(p & kRingMask) >> 2]) = v;
                    loaded += kTile;
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t rem = n - c;
                const uint32_t elim = rem < kNibbleMax ? rem : kNibbleMax;
                const bool differs = lane < elim &&
                                     ring_byte(L.ring, c + lane) != ring_byte(L.ring, c + lane - off);
                const uint64_t stop = __ballot(differs) | (1ull << elim);
                e = uniform((uint32_t)__builtin_ctzll(stop));
                sink_put(s, L.stage, lane, e, 4);
                c += e;
            } while (e == kNibbleMax);
        }
    }

    sink_finish(s, L.stage, lane, &out_len[b]);
}

// ---------------------------------------------------------------------------------
// Lane-parallel output: a 4096-bit ring of big-endian words in LDS.  Tokens are OR-ed in
// at their bit offsets (many lanes at once, offsets from a wave prefix sum of the token
// widths), and each 2048-bit half is drained to HBM with one coalesced store as soon as it
// is complete.  State is wave-uniform.  Truncation as lzs-compression.c:304-313.
// ---------------------------------------------------------------------------------
constexpr uint32_t kBitWords = 128;                  // 512 B

struct BitRing {
    uint32_t flushed;    // bytes handed to HBM (multiple of 256)
    uint32_t head;       // bits appended past `flushed` (< 4096)
    uint8_t *dst;
    uint32_t cap;
    bool     aligned4;
};

__device__ __forceinline__ uint32_t br_at(const BitRing &e) { return ((e.flushed << 3) + e.head) & 4095u; }

// OR the low `width` (1..32) bits of `value`, MSB first, at ring bit offset `at`.
__device__ __forceinline__ void br_or(uint32_t *words, uint32_t at, uint32_t value, uint32_t width)
{
    const uint32_t sh = at & 31u, d = at >> 5;
    const uint64_t v = (uint64_t)value << (64u - sh - width);
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    __hip_atomic_fetch_or(&words[d], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lo) __hip_atomic_fetch_or(&words[(d + 1) & (kBitWords - 1)], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void br_flush(BitRing &e, uint32_t *words, uint32_t lane)
{
    while (e.head >= 2048u) {
        __builtin_amdgcn_wave_barrier();
        const uint32_t slot = ((e.flushed >> 8) & 1u) * 64u + lane;
        const uint32_t v = __builtin_bswap32(words[slot]);
        words[slot] = 0;
        const uint32_t at = e.flushed + 4 * lane;
        if (e.aligned4 && at + 4 <= e.cap) {
            *reinterpret_cast<uint32_t *>(e.dst + at) = v;
        } else {
            for (uint32_t k = 0; k < 4; k++)
                if (at + k < e.cap) e.dst[at + k] = (uint8_t)(v >> (8 * k));
        }
        __builtin_amdgcn_wave_barrier();
        e.flushed += 256u;
        e.head -= 2048u;
    }
}

// one field from the whole wave (wave-uniform value/width)
__device__ __forceinline__ void br_put(BitRing &e, uint32_t *words, uint32_t lane, uint32_t value, uint32_t width)
{
    if (lane == 0) br_or(words, br_at(e), value, width);
    e.head += width;
    br_flush(e, words, lane);
}

// Inclusive prefix sum across the wave with DPP adds only (no LDS round trips):
// row_shr 1/2/4/8 scan each row of 16, row_bcast 15/31 carry the row totals upward.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// Emit the tokens that start at the marked lanes of one 64-position chunk (lane = position);
// `r` is the lane's search result (len << 11 | off, len < 8 here), `byte` its input byte.
// Token formats: lzs-compression.c:365-409.
__device__ __forceinline__ void br_emit_chunk(BitRing &e, uint32_t *words, uint32_t lane,
                                              uint64_t marks, uint32_t r, uint32_t byte)
{
    if (marks == 0ull) return;
    const bool mine = (marks >> lane) & 1ull;
    const uint32_t len = r >> 11, off = r & kWindow;
    uint32_t value = byte & 0xFFu, width = 9;                         // 0 bbbbbbbb
    if (len >= 2) {
        const uint32_t ov = off <= kShortMax ? ((3u << 7) | off) : ((2u << 11) | off);
        const uint32_t ow = off <= kShortMax ? 9u : 13u;
        const uint32_t lv = len <= 4 ? len - 2 : 0xCu + (len - 5);
        const uint32_t lw = len <= 4 ? 2u : 4u;
        value = (ov << lw) | lv;
        width = ow + lw;
    }
    const uint32_t w = mine ? width : 0u;
    const uint32_t incl = wave_inclusive_sum(w);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (mine) br_or(words, (br_at(e) + incl - w) & 4095u, value, width);
    e.head += total;
    br_flush(e, words, lane);
}

// End marker 1 1 0000000, zero pad to a byte, drain (lzs-compression.c:449-466).
__device__ __forceinline__ void br_finish(BitRing &e, uint32_t *words, uint32_t lane, uint32_t *len_out)
{
    br_put(e, words, lane, 0x180u, 9);
    e.head = (e.head + 7u) & ~7u;                                     // pad bits are already zero
    br_flush(e, words, lane);
    __builtin_amdgcn_wave_barrier();
    const uint32_t nbytes = e.head >> 3;
    for (uint32_t i = lane; i < nbytes; i += 64) {
        const uint32_t bit = ((e.flushed << 3) + 8 * i) & 4095u;
        const uint32_t v = (words[bit >> 5] >> (24u - (bit & 24u))) & 0xFFu;
        if (e.flushed + i < e.cap) e.dst[e.flushed + i] = (uint8_t)v;
    }
    const uint32_t total = e.flushed + nbytes;
    if (lane == 0) *len_out = total < e.cap ? total : e.cap;
}

// ---------------------------------------------------------------------------------
// lzs_compress() per block, variant "chain" (the default).
//
// The search rule is a pure function of (input, position), so it is hoisted out of the
// serial parse and made position-parallel.  One wave owns one block and works in rounds
// over a POOL of 512 consecutive positions:
//   1. BUILD   insert the pool's positions, 64 per instruction, into two
//              previous-occurrence chains: one keyed by a hash of the 3 bytes at the
//              position, one by a hash of 2.  Only offsets whose first bytes match can
//              give a match, so each chain is a complete, nearest-first candidate list for
//              its length class (the reference's chains, lzs-compression.c:328-361,435-443,
//              rest on the same fact with a 2-byte key).
//   2. SEARCH  lanes pull positions from the pool as they become free (walk lengths vary
//              a lot, so lane = position would idle most lanes).  A position first walks
//              its 3-byte chain through the whole window: every offset with a common
//              prefix >= 3 is on it, so the nearest-longest rule (:337-345) is decided
//              there whenever any match >= 3 exists.  Otherwise the answer is the nearest
//              true 2-byte match: the first verified candidate on the 2-byte chain.
//   3. PARSE   the greedy token loop (:365-431) runs on the scalar unit over the pool's
//              results and packs bits.
// Positions swallowed by a long match are built but not searched.
//
// Chain storage, per wave, in LDS: head3[1024], head2[1024] (latest position per hash),
// link3[2560], link2[2560] (per position, ring of 40 batches: distance to the previous
// position with the same hash; anything > 2047 means none).
// Lanes of one build instruction that share a hash are chained by ONE ds_wrxchg_rtn: on
// gfx950 same-address LDS atomics of one wave instruction apply in ascending lane order,
// so each lane receives its nearest lower neighbour (or the older head).  That ordering
// is a measured property, not an ISA promise: tools/probes/lds_order_probe.hip and
// tests/test_gpu_lds_order.py check it on the device over 2.6e5 conflict patterns.
// ---------------------------------------------------------------------------------
#ifdef LZS_PROFILE
// Diagnostic build only (tools/probes/prof_compress): per-phase cycle sums over all waves.
__device__ unsigned long long lzs_prof[32];
#define PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[32] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}
#define PROF_MARK(i) do { unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_t; prof_t = t_; } while (0)
#define PROF_COUNT(i, v) do { prof_acc[i] += (v); } while (0)
#define PROF_T0 unsigned long long prof_u = __builtin_readcyclecounter()
#define PROF_T1(i) do { prof_acc[i] += __builtin_readcyclecounter() - prof_u; } while (0)
#define PROF_T0B unsigned long long prof_v = __builtin_readcyclecounter()
#define PROF_T1B(i) do { prof_acc[i] += __builtin_readcyclecounter() - prof_v; } while (0)
#define PROF_T0C unsigned long long prof_w = __builtin_readcyclecounter()
#define PROF_T1C(i) do { prof_acc[i] += __builtin_readcyclecounter() - prof_w; } while (0)
#define PROF_STAMP0 unsigned long long prof_s = __builtin_readcyclecounter()
#define PROF_STAMP(i) do { unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_s; prof_s = t_; } while (0)
#define PROF_DONE do { if (lane == 0) for (int i_ = 0; i_ < 32; i_++) atomicAdd(&lzs_prof[i_], prof_acc[i_]); } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_COUNT(i, v)
#define PROF_T0
#define PROF_T1(i)
#define PROF_T0B
#define PROF_T1B(i)
#define PROF_T0C
#define PROF_T1C(i)
#define PROF_STAMP0
#define PROF_STAMP(i)
#define PROF_DONE
#endif

constexpr uint32_t kPool      = 512;             // positions hoisted per round
constexpr uint32_t kLinkN     = 2560;            // 40 x 64 >= 2047 + kPool
#ifndef LZS_HEAD3_BITS
#define LZS_HEAD3_BITS 10
#endif
#ifndef LZS_HEAD2_BITS
#define LZS_HEAD2_BITS 10
#endif
constexpr uint32_t kHead3     = 1u << LZS_HEAD3_BITS;
constexpr uint32_t kHead2     = 1u << LZS_HEAD2_BITS;
constexpr uint32_t kNoLink    = 0xFFFFu;
#ifndef LZS_REFILL_MIN
#define LZS_REFILL_MIN 32
#endif
constexpr uint32_t kRefillMin = LZS_REFILL_MIN;  // idle lanes that justify a refill pass

struct __attribute__((aligned(16))) ChainLds {
    uint32_t ring[kRingWords + 4];               // +16 B mirror of ring[0..15]: reads never wrap
    uint32_t head3[kHead3];
    uint32_t head2[kHead2];
    uint16_t link3[kLinkN];
    uint16_t link2[kLinkN];
    uint16_t res[kPool];                         // (len << 11) | off per pool position
    uint32_t bits[kBitWords];                    // output bit ring
};

__device__ __forceinline__ void ringm_read12(const uint32_t *ring, uint32_t q,
                                             uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
    const uint32_t a = (q & kRingMask) >> 2;     // words a..a+3 exist thanks to the mirror
    const uint32_t d0 = ring[a], d1 = ring[a + 1], d2 = ring[a + 2], d3 = ring[a + 3];
    w0 = __builtin_amdgcn_alignbyte(d1, d0, q);  // v_alignbyte_b32 shifts by the low two bits of q
    w1 = __builtin_amdgcn_alignbyte(d2, d1, q);
    w2 = __builtin_amdgcn_alignbyte(d3, d2, q);
}

__device__ __forceinline__ void chain_refill(ChainLds &L, const uint8_t *src, uint32_t n, bool src16,
                                             uint32_t lane, uint32_t &loaded, uint32_t need)
{
    while (loaded < n && loaded < need) {
        const uint32_t p = loaded + 16 * lane;
        const uint4 v = load16(src, p, n, src16);
        const uint32_t at = (p & kRingMask) >> 2;
        *reinterpret_cast<uint4 *>(&L.ring[at]) = v;
        if (at == 0) *reinterpret_cast<uint4 *>(&L.ring[kRingWords]) = v;
        loaded += kTile;
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t link_slot_base(uint32_t B) { return ((B >> 6) % 40u) * 64u; }

// BUILD for the 64 positions starting at B (multiple of 64).
__device__ __forceinline__ void chain_build(ChainLds &L, uint32_t B, uint32_t n, uint32_t lane)
{
    const uint32_t p = B + lane;
    const uint32_t a = (p & kRingMask) >> 2;
    const uint32_t t0 = __builtin_amdgcn_alignbyte(L.ring[a + 1], L.ring[a], p & 3);
    const uint32_t h3 = ((t0 & 0xFFFFFFu) * 0x9E3779B1u) >> (32 - LZS_HEAD3_BITS);           // 10 bits
    const uint32_t h2 = (((t0 & 0xFFFFu) * 40503u) >> 6) & (kHead2 - 1);  // 10 bits
    uint32_t d3 = kNoLink, d2 = kNoLink;
    // ds_wrxchg_rtn_b32: lanes sharing a slot are served in ascending lane order (see above)
    if (p + 2 < n) {
        const uint32_t old = __hip_atomic_exchange(&L.head3[h3], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t d = p - old;                       // old == ~0u (never set) -> p + 1: too far
        d3 = d < kNoLink ? d : kNoLink;
    }
    if (p + 1 < n) {
        const uint32_t old = __hip_atomic_exchange(&L.head2[h2], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t d = p - old;
        d2 = d < kNoLink ? d : kNoLink;
    }
    const uint32_t slot = link_slot_base(B) + lane;
    L.link3[slot] = (uint16_t)d3;
    L.link2[slot] = (uint16_t)d2;
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64)
void lzs_compress_blocks_kernel(uint8_t *__restrict__ out, size_t out_stride, uint32_t out_cap,
                                uint32_t *__restrict__ out_len,
                                const uint8_t *__restrict__ in, size_t in_stride,
                                const uint32_t *__restrict__ in_len, uint32_t in_len_uniform,
                                uint32_t nblocks)
{
    __shared__ ChainLds L;
    const uint32_t lane = threadIdx.x;
    const uint32_t b    = blockIdx.x;
    if (b >= nblocks) return;

    const uint8_t *src = in + (size_t)b * in_stride;
    const uint32_t n   = in_len ? in_len[b] : in_len_uniform;
    const bool src16   = ((uintptr_t)src & 15u) == 0;

    BitRing s;
    s.flushed = 0; s.head = 0;
    s.dst = out + (size_t)b * out_stride;
    s.cap = out_cap;
    s.aligned4 = ((uintptr_t)s.dst & 3u) == 0;

    L.bits[lane] = 0; L.bits[64 + lane] = 0;
    for (uint32_t i = lane; i < kHead3; i += 64) L.head3[i] = ~0u;
    for (uint32_t i = lane; i < kHead2; i += 64) L.head2[i] = ~0u;
    __builtin_amdgcn_wave_barrier();

    uint32_t c = 0;          // start of the next token
    uint32_t loaded = 0;     // ring holds [loaded-4096, loaded)
    uint32_t next = 0;       // next batch of 64 positions to build
    PROF_DECL;

    while (c < n) {
        if (s.flushed >= s.cap) break;                        // output full (:306-309)
        PROF_MARK(4);
        // batches wholly behind c (inside a match): keep the chains complete, no search
        while (next + 64 <= c) {
            chain_refill(L, src, n, src16, lane, loaded, c + 80);
            chain_build(L, next, n, lane);
            next += 64;
        }
        // ---- BUILD the pool [Pb, Pe)
        const uint32_t Pb = next;
        const uint32_t Pe = Pb + kPool < ((n + 63u) & ~63u) ? Pb + kPool : ((n + 63u) & ~63u);
        chain_refill(L, src, n, src16, lane, loaded, (Pe > c ? Pe : c) + 80);
        PROF_MARK(0);
        for (uint32_t B = Pb; B < Pe; B += 64) chain_build(L, B, n, lane);
        PROF_MARK(1);
        next = Pe;
        const uint32_t pend = Pe < n ? Pe : n;
        const uint32_t slot0 = link_slot_base(Pb);

        // ---- SEARCH with lanes pulling positions from the pool (:322-363).
        // The loop body is written branch-free (selects, unconditional in-range LDS reads):
        // one wave alone on its SIMD pays for every instruction, scalar mask juggling included.
        {
            uint32_t nextp = c > Pb ? c : Pb;
            bool busy = false, three = false;
            uint32_t p = Pb, t0 = 0, t1 = 0, t2 = 0, lim = 0, reach = 0, myslot = slot0, first2 = kNoLink;
            uint32_t cum = 0, dist = kNoLink, best_len = 0, best_off = 0;
            for (;;) {
                const uint64_t idle = __ballot(!busy);
                const uint32_t nidle = (uint32_t)__builtin_popcountll(idle);
                if (nextp < pend && (nidle >= kRefillMin || nidle == 64u)) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    const uint32_t np = nextp + rank;
                    const bool take = !busy && np < pend;
                    const uint32_t pp = take ? np : p;                  // every lane reads in range
                    uint32_t n0, n1, n2;
                    ringm_read12(L.ring, pp, n0, n1, n2);
                    uint32_t sl = slot0 + (pp - Pb);
                    sl = sl >= kLinkN ? sl - kLinkN : sl;
                    const uint32_t l3 = L.link3[sl], l2 = L.link2[sl];
                    const uint32_t nlim = n - pp < kSearchCap ? n - pp : kSearchCap;
                    p = pp; myslot = sl;
                    t0 = take ? n0 : t0; t1 = take ? n1 : t1; t2 = take ? n2 : t2;
                    lim = take ? nlim : lim;
                    reach = take ? (pp < kWindow ? pp : kWindow) : reach;
                    three = take ? nlim >= 3 : three;
                    first2 = take ? (nlim >= 2 ? l2 : kNoLink) : first2;
                    dist = take ? (nlim >= 3 ? l3 : (nlim >= 2 ? l2 : kNoLink)) : dist;
                    cum = take ? 0u : cum;
                    best_len = take ? 0u : best_len;
                    best_off = take ? 0u : best_off;
                    busy = busy || take;
                    nextp += nidle;
                }
                if (__ballot(busy) == 0ull) break;
                PROF_COUNT(5, 1);
                PROF_COUNT(6, __builtin_popcountll(__ballot(busy)));

                const uint32_t cum2 = cum + dist;
                const bool inwin = busy && cum2 <= reach;
                uint32_t w0, w1, w2;
                ringm_read12(L.ring, p - (inwin ? cum2 : 0u), w0, w1, w2);
                int32_t at = (int32_t)myslot - (int32_t)(inwin ? cum2 : 0u);
                at = at < 0 ? at + (int32_t)kLinkN : at;
                const uint32_t nd = three ? L.link3[at] : L.link2[at];
                const uint32_t e0 = eq_bytes(w0 ^ t0);
                const uint32_t e1 = eq_bytes(w1 ^ t1);
                const uint32_t e2 = eq_bytes(w2 ^ t2);
                uint32_t len = e0 + (e0 == 4 ? e1 + (e1 == 4 ? e2 : 0u) : 0u);
                len = len < lim ? len : lim;
                // 3-byte chain: first strictly longer match wins, the cap ends the walk (:337-345);
                // 2-byte chain: the first verified candidate is the answer
                const bool better = three ? len > best_len : len >= 2;
                const bool takeit = inwin && better;
                best_len = takeit ? len : best_len;
                best_off = takeit ? cum2 : best_off;
                const bool ended = !inwin || (three ? len == lim : len >= 2);
                // nothing >= 3 in the whole window: restart on the 2-byte chain
                const bool fallback = busy && ended && three && best_len < 3;
                const bool finish = busy && ended && !fallback;
                cum = fallback ? 0u : cum2;
                dist = fallback ? first2 : nd;
                best_len = fallback ? 0u : best_len;
                best_off = fallback ? 0u : best_off;
                three = three && !fallback;
                if (finish) L.res[p - Pb] = (uint16_t)((best_len << 11) | best_off);
                busy = busy && !finish;
            }
        }
        __builtin_amdgcn_wave_barrier();
        PROF_MARK(2);

        // ---- PARSE + PACK, 64 positions (one chunk) at a time.  The greedy chase over token
        // starts is scalar and touches one register per token; the tokens it marks are
        // encoded and written by all lanes at once.  Only matches that reach the extended
        // length field (first code 8) are handled one by one.
        for (uint32_t k = (c - Pb) >> 6; Pb + 64 * k < pend; k++) {
            const uint32_t base = Pb + 64 * k;
            const uint32_t cend = base + 64 < pend ? base + 64 : pend;
            if (c >= cend) continue;
            if (s.flushed >= s.cap) break;
            PROF_T0;
            const uint32_t r = L.res[64 * k + lane];
            const uint32_t byte = ring_byte(L.ring, base + lane);
            const uint32_t len_l = r >> 11;
            // The chase runs on the scalar unit over bit planes of the per-position step
            // (bytes a token starting there consumes, 1..8), so it needs no per-token lane read.
            const uint32_t nvalid = cend - base;              // lanes past the pool end hold stale results
            const uint64_t vmask = nvalid >= 64u ? ~0ull : ((1ull << nvalid) - 1ull);
            const uint32_t sm1 = len_l < 2 ? 0u : (len_l < kTokenMax ? len_l - 1 : kTokenMax - 1);
            const uint64_t matches = __ballot(len_l >= 2) & vmask;
            const uint64_t isext = __ballot(len_l >= kTokenMax) & vmask;
            const uint64_t s0 = __ballot(sm1 & 1u), s1 = __ballot(sm1 & 2u), s2 = __ballot(sm1 & 4u);
            uint64_t marks = 0;
            PROF_T1(9);
            PROF_T0B;
            while (c < cend) {
                const uint32_t li = c - base;
                const uint64_t ahead = matches >> li;
                if ((ahead & 1ull) == 0ull) {                 // a run of literals: mark it whole
                    const uint32_t left = cend - c;
                    const uint32_t run = ahead ? (uint32_t)__builtin_ctzll(ahead) : 64u;
                    const uint32_t nlit = run < left ? run : left;
                    marks |= (nlit >= 64u ? ~0ull : ((1ull << nlit) - 1ull)) << li;
                    c += nlit;
                    continue;
                }
                if (((isext >> li) & 1ull) == 0ull) {
                    marks |= 1ull << li;
                    c += 1u + (uint32_t)((s0 >> li) & 1ull) + 2u * (uint32_t)((s1 >> li) & 1ull)
                            + 4u * (uint32_t)((s2 >> li) & 1ull);
                    continue;
                }
                // ---- extended match (:411-431): flush what is marked, then this token
                PROF_COUNT(8, 1);
                br_emit_chunk(s, L.bits, lane, marks, r, byte);
                marks = 0;
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)li) & kWindow;
                if (off <= kShortMax) br_put(s, L.bits, lane, (((3u << 7) | off) << 4) | 0xFu, 13);
                else                  br_put(s, L.bits, lane, (((2u << 11) | off) << 4) | 0xFu, 17);
                c += kTokenMax;
                bool more = true;
                while (more) {                               // up to 60 bytes = 4 nibbles per round
                    chain_refill(L, src, n, src16, lane, loaded, c + 64);
                    // keep the chains current while c runs ahead of the built range: every
                    // batch now wholly behind c is inserted while its bytes are in the ring
                    while (next + 128 <= c) {
                        chain_build(L, next, n, lane);
                        next += 64;
                    }
                    const uint32_t rem = n - c;
                    const uint32_t span = rem < 60u ? rem : 60u;
                    const bool differs = lane < span &&
                                         ring_byte(L.ring, c + lane) != ring_byte(L.ring, c + lane - off);
                    const uint64_t stopmask = __ballot(differs) | (1ull << span);
                    const uint32_t m = uniform((uint32_t)__builtin_ctzll(stopmask));  // equal bytes <= span
                    c += m;
                    const uint32_t full = m / kNibbleMax;                             // nibbles of 15
                    if (full) br_put(s, L.bits, lane, (1u << (4 * full)) - 1u, 4 * full);
                    // 60 equal bytes = four full nibbles and the match may go on; anything
                    // shorter ends it with a last nibble of 0..14 (0 when it ended on a
                    // multiple of 15 or at the end of the input)
                    more = (m == 60u);
                    if (!more) br_put(s, L.bits, lane, m % kNibbleMax, 4);
                    if (s.flushed >= s.cap) more = false;
                }
            }
            PROF_T1B(10);
            PROF_T0C;
            br_emit_chunk(s, L.bits, lane, marks, r, byte);
            PROF_T1C(11);
        }
        PROF_MARK(3);
        PROF_COUNT(7, 1);
    }
    br_finish(s, L.bits, lane, &out_len[b]);
    PROF_DONE;
}

// ---------------------------------------------------------------------------------
// lzs_compress() per block, variant "wg" (the default): the same algorithm as "chain",
// but ONE 256-THREAD WORKGROUP PER BLOCK.  The four waves share a single set of LDS tables
// (ring, heads, links), which quadruples the waves a CU can hold for a given LDS budget,
// and every phase is lane-parallel over 256 threads, separated by workgroup barriers:
// Pools of 512 positions are pipelined: pool k is built and searched while pool k-1 is
// parsed and packed (walks of pool k still running are carried into the next round).
//   BUILD   HASH: buckets and "inserted at all" per position, dealt over the four waves;
//           CHAIN: wave 0 / wave 1 chain the 3-byte / 2-byte buckets by ordered exchange.
//           Positions inside a run of one byte value (same byte before, 13 equal bytes
//           ahead) are not inserted -- they can only ever win as offset 1, the seed of
//           every search;
//   SEARCH  the four waves pull positions from one shared counter; one branch-free step
//           per candidate for all 64 lanes;
//   EXTEND  a match that fills the search cap is measured on to <= 59 more bytes so that
//           its token is complete (only the first of each run of such positions compares
//           bytes); longer ones are left "open";
//   PARSE   the greedy chain of token starts (lzs-compression.c:301-447) by pointer
//           doubling inside 64-position chunks (in registers, for every possible entry),
//           a walk over the 8 chunk exits, and lane m taking the m-th token of its chunk;
//   PACK    tokens are encoded by their lanes, a workgroup prefix sum of the bit widths
//           places them, and complete 256-byte quarters of the bit ring go out as
//           coalesced stores.  Only open matches (long runs) are finished by wave 0
//           alone, 240 bytes per step.
// DESIGN.md section 3.1 has the LDS table, the measurements and what bounds the kernel.
// ---------------------------------------------------------------------------------
constexpr uint32_t kWgThreads  = 256;
// 3-byte buckets: as many as the LDS left over at five workgroups per CU holds (31.0 KB per
// workgroup still fits five, 31.5 KB does not; not a power of two: the hash is scaled into the range).  A quarter fewer collisions than 1024 buckets, and a
// collision costs a whole SEARCH step.
constexpr uint32_t kWgHead3    = 1256;
constexpr uint32_t kWgPool     = 512;
constexpr uint32_t kWgLinkN    = 3072;            // 48 x 64 >= 2047 + 2 * kWgPool: the pool in SEARCH and the one before
constexpr uint32_t kWgResN     = 2 * kWgPool;     // results of two consecutive pools, by position & (kWgResN - 1)
constexpr uint32_t kWgBitWords = 256;             // 8192-bit ring, quarters of 2048 bits
constexpr uint32_t kOpen       = 1023;            // jump code of an open match
constexpr uint32_t kExtOpen    = 63;              // extension code of an open match
#ifndef LZS_SUBSTEPS
#define LZS_SUBSTEPS 3
#endif
constexpr uint32_t kExtMax     = 59;              // longest extension resolved in SEARCH

struct __attribute__((aligned(16))) BlkLds {
    uint32_t ring[kRingWords + 4];                // +16 B mirror of ring[0..15]
    uint32_t head3[kWgHead3];
    uint32_t head2[kHead2];
    uint16_t link3[kWgLinkN];
    uint16_t link2[kWgLinkN];
    uint32_t res[kWgResN];                        // off | len << 11 | ext << 15
    uint16_t exitfn[kWgPool];                     // per entry position: where the chain leaves its chunk
    uint32_t bits[kWgBitWords];                   // output bit ring
    uint32_t chunk_bits[8];
    uint32_t nextp;                               // SEARCH work counter
    uint32_t bcast[8];
#ifdef LZS_PAD_LDS
    uint32_t pad[LZS_PAD_LDS / 4];                // occupancy experiments only
#endif
};

// Bucket of the 3 bytes in the low 24 bits of t.
__device__ __forceinline__ uint32_t wg_hash3(uint32_t t) { return __umulhi((t & 0xFFFFFFu) * 0x9E3779B1u, kWgHead3); }

__device__ __forceinline__ uint32_t wg_slot_base(uint32_t B) { return ((B >> 6) % (kWgLinkN / 64u)) * 64u; }

// Equal leading bytes of two 12-byte strings given the XOR of their little-endian words;
// 12 or more (a large number) when all are equal -- callers clamp to their own limit.
__device__ __forceinline__ uint32_t lcp12(uint32_t x0, uint32_t x1, uint32_t x2)
{
    // v_ffbl_b32 gives -1 for 0, which survives the OR: the first set bit of x2:x1:x0, or huge
    uint32_t f0, f1, f2;
    asm("v_ffbl_b32 %0, %1" : "=v"(f0) : "v"(x0));
    asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"(x1));
    asm("v_ffbl_b32 %0, %1" : "=v"(f2) : "v"(x2));
    const uint32_t a = f1 | 32u, b = f2 | 64u;
    const uint32_t m = f0 < a ? (f0 < b ? f0 : b) : (a < b ? a : b);   // v_min3_u32
    return m >> 3;                                                      // all equal -> huge
}

// OR `width` (1..32) bits of `value`, MSB first, at bit offset `at` of a ring of `words` words.
__device__ __forceinline__ void bits_or(uint32_t *ring, uint32_t words, uint32_t at, uint32_t value, uint32_t width)
{
    const uint32_t sh = at & 31u, d = (at >> 5) & (words - 1);
    const uint64_t v = (uint64_t)value << (64u - sh - width);
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    __hip_atomic_fetch_or(&ring[d], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lo) __hip_atomic_fetch_or(&ring[(d + 1) & (words - 1)], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Output state: wave-uniform and identical in all four waves.
struct WgOut {
    uint32_t flushed;    // bytes handed to HBM (multiple of 256)
    uint32_t head;       // bits appended past `flushed`
    uint8_t *dst;
    uint32_t cap;
    bool     aligned4;
    // Segments of one long stream only: `cap` above stays unlimited (the bits are counted to the
    // end whatever happens), stores stop at `limit` (the slot), or -- `ored` -- go to zeroed memory
    // shared with the neighbouring segments by atomic OR.  A block: limit = cap.
    uint32_t limit;
    bool     ored;
};

__device__ __forceinline__ uint32_t wg_bit_at(const WgOut &o) { return ((o.flushed << 3) + o.head) & 8191u; }

// Store one complete quarter (2048 bits) of the ring from the calling wave and clear it.
__device__ __forceinline__ void wg_store_quarter(const WgOut &o, BlkLds &L, uint32_t lane)
{
    const uint32_t slot = ((o.flushed >> 8) & 3u) * 64u + lane;
    const uint32_t v = __builtin_bswap32(L.bits[slot]);
    L.bits[slot] = 0;
    const uint32_t at = o.flushed + 4 * lane;
    if (o.ored) {                                             // o.dst is 4-aligned in this mode
        if (v) atomicOr(reinterpret_cast<unsigned int *>(o.dst + at), v);
        return;
    }
    if (o.aligned4 && at + 4 <= o.limit) {
        *reinterpret_cast<uint32_t *>(o.dst + at) = v;
    } else {
        for (uint32_t k = 0; k < 4; k++)
            if (at + k < o.limit) o.dst[at + k] = (uint8_t)(v >> (8 * k));
    }
}

// ---- helpers used by wave 0 alone while it finishes an open match
__device__ __forceinline__ void wave_refill(BlkLds &L, const uint8_t *src, uint32_t n, bool src16,
                                            uint32_t lane, uint32_t &loaded, uint32_t need)
{
    while (loaded < n && loaded < need) {
        const uint32_t p = loaded + 16 * lane;
        const uint4 v = load16(src, p, n, src16);
        const uint32_t at = (p & kRingMask) >> 2;
        *reinterpret_cast<uint4 *>(&L.ring[at]) = v;
        if (at == 0) *reinterpret_cast<uint4 *>(&L.ring[kRingWords]) = v;
        loaded += kTile;
    }
    __builtin_amdgcn_wave_barrier();
}

template <class T> __device__ __forceinline__ T opaque(T x) { asm volatile("" : "+v"(x)); return x; }

// Where an idle lane's stores go: a word of its own in exitfn[], which is only live inside PARSE.
__device__ __forceinline__ uint32_t *wg_dummy(BlkLds &L) { return reinterpret_cast<uint32_t *>(L.exitfn) + threadIdx.x; }

// BUILD in two phases, so that the arithmetic is done once per position instead of once per wave.
// HASH: the batches of [B0, Be) are dealt to the four waves; a wave computes, for each position
// of its batches, both bucket numbers and whether the position is inserted at all (not inside a
// run, gram within the input), and leaves them as a record in the result slot of that position
// (free until SEARCH).  Whether a position is inside a run is read off ballots of "this byte
// equals the next" over three neighbouring batches.
// CHAIN (after a barrier): wave 0 keeps the 3-byte chain and wave 1 the 2-byte chain, so every
// bucket is chained by one wave's in-order instruction stream; waves 2 and 3 wait (the kernel is
// bound by VALU issue, and chaining four ways only repeated the record decoding).  No
// exec-masked regions and no branches inside the loops: a lane that must not insert exchanges
// with its dummy word instead.
__device__ __forceinline__ void wg_hash_range(BlkLds &L, uint32_t B0, uint32_t Be, uint32_t n, uint32_t lane, uint32_t wave)
{
    const auto text4 = [&](uint32_t q) {
        const uint32_t a = (q & kRingMask) >> 2;
        return __builtin_amdgcn_alignbyte(L.ring[a + 1], L.ring[a], q);
    };
    // bit l: byte B+l equals byte B+l+1, and both are input
    const auto eqnext = [&](uint32_t q, uint32_t t) {
        const uint32_t diff = ((t ^ (t >> 8)) & 0xFFu) | (q + 1u < n ? 0u : 1u);
        return __builtin_amdgcn_ballot_w64(diff == 0u);
    };
    // lane l needs the 13 bits from bit l-1 of (enext : ecur : eprev >> 63)
    const bool first = lane == 0u, low = lane <= 32u;
    const uint32_t shift = (lane + 31u) & 31u;
    // two batches of the wave side by side (straight-line code: their LDS reads overlap); one
    // that lies past Be is computed all the same and stored to the dummy word
    for (uint32_t B = B0 + 64u * wave; B < Be; B += 512u) {
        uint32_t tprev[2], tcur[2], tnext[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t p = B + 256u * j + lane;
            tprev[j] = text4(p - 64u); tcur[j] = text4(p); tnext[j] = text4(p + 64u);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t Bj = B + 256u * j, p = Bj + lane;
            const uint64_t eprev = Bj >= 64u ? eqnext(p - 64u, tprev[j]) : 0ull;
            const uint64_t ecur = eqnext(p, tcur[j]), enext = eqnext(p + 64u, tnext[j]);
            const uint32_t w0 = (uint32_t)(eprev >> 32), w1 = (uint32_t)ecur, w2 = (uint32_t)(ecur >> 32), w3 = (uint32_t)enext;
            const uint32_t lo = first ? w0 : (low ? w1 : w2), hi = first ? w1 : (low ? w2 : w3);
            const uint32_t run = __builtin_amdgcn_alignbit(hi, lo, shift) & 0x1FFFu;
            // Interior of a run: the same byte before, and 13 equal bytes ahead.  Such a position is
            // dominated by p+1 as a candidate for every later position, and its own search ends at
            // offset 1 (full cap), so it is neither inserted nor does it need a link.
            const uint32_t skip = opaque(run == 0x1FFFu ? 1u : 0u);
            const uint32_t h3 = wg_hash3(tcur[j]);
            const uint32_t h2 = (((tcur[j] & 0xFFFFu) * 40503u) >> 6) & (kHead2 - 1);
            const uint32_t out3 = skip | (p + 2u < n ? 0u : 1u), out2 = skip | (p + 1u < n ? 0u : 1u);
            uint32_t *const to = Bj < Be ? &L.res[p & (kWgResN - 1)] : wg_dummy(L);
            *to = h3 | (h2 << 11) | (out3 << 21) | (out2 << 22);
        }
    }
}

// CHAIN for `K` consecutive batches from B: all reads, then all exchanges, then all links, so the
// LDS latencies are paid once per group.
template <int K>
__device__ __forceinline__ void wg_chain_group(BlkLds &L, uint32_t B, uint32_t &slot, uint32_t lane,
                                               uint32_t *heads, uint16_t *links, uint32_t hshift, uint32_t hmask, uint32_t oshift)
{
    uint32_t *const dummy = wg_dummy(L);
    uint32_t rec[K], old[K], out[K];
    uint16_t *la[K];
#pragma unroll
    for (int j = 0; j < K; j++) rec[j] = L.res[(B + 64u * j + lane) & (kWgResN - 1)];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t p = B + 64u * j + lane;
        const uint32_t h = (rec[j] >> hshift) & hmask;
        out[j] = (rec[j] >> oshift) & 1u;
        uint32_t *const ha = out[j] == 0u ? &heads[h] : dummy;
        old[j] = __hip_atomic_exchange(ha, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        la[j] = &links[slot + lane];
        slot = slot + 64u >= kWgLinkN ? slot + 64u - kWgLinkN : slot + 64u;
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t p = B + 64u * j + lane;
        uint32_t d = p - old[j] < kNoLink ? p - old[j] : kNoLink;
        d = out[j] == 0u ? d : kNoLink;
        *la[j] = (uint16_t)d;
    }
}

__device__ __forceinline__ void wg_chain_range(BlkLds &L, uint32_t B0, uint32_t Be, uint32_t lane, uint32_t wave)
{
    if (wave >= 2u) return;                                    // waves 2 and 3 only wait: the kernel is VALU-bound
    const bool two = wave == 1u;                               // uniform
    uint32_t *const heads = two ? L.head2 : L.head3;
    uint16_t *const links = two ? L.link2 : L.link3;
    const uint32_t hshift = two ? 11u : 0u, hmask = two ? 0x3FFu : 0x7FFu, oshift = two ? 22u : 21u;
    uint32_t slot = wg_slot_base(B0);
    uint32_t B = B0;
    for (; B + 256u <= Be; B += 256u) wg_chain_group<4>(L, B, slot, lane, heads, links, hshift, hmask, oshift);
    for (; B < Be; B += 64u) wg_chain_group<1>(L, B, slot, lane, heads, links, hshift, hmask, oshift);
}

// EXTEND: a match that fills the search cap (12) may run on, and its token is only complete with
// the whole length (:417-431).  SEARCH leaves bits 15.. of such results empty; this pass fills them
// for the pool about to be parsed, for both chunks of the calling wave.  Consecutive positions
// inside one long match all report 12 at the same offset, and their totals differ by one per
// position, so only the first of each such run (per 64-position chunk) compares bytes -- up to
// kTokenMax + kExtMax + 1 of them -- and the others derive theirs.  Beyond that the match is
// "open" (finished serially in PARSE), and so is everything derived from an open one.
__device__ __forceinline__ void wg_extend(BlkLds &L, uint32_t Pb, uint32_t entry, uint32_t npos, uint32_t n, uint32_t lane, uint32_t wave,
                                          uint32_t (&rr)[2])
{
    constexpr uint32_t kRoom = kTokenMax + kExtMax + 1;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t gi = 256u * h + 64u * wave + lane;
        rr[h] = 0;
        if (64u * wave + 256u * h >= npos) continue;           // chunk past the end of the pool (uniform)
        const uint32_t p = Pb + gi;
        // SEARCH stored (len << 16) - offset; offset 0 = no match
        const uint32_t k = L.res[p & (kWgResN - 1)];
        const uint32_t off = (0u - k) & 0xFFFFu;
        const uint32_t len = off ? (k + 0xFFFFu) >> 16 : 0u;
        const uint32_t r = off | (len << 11);
        // positions before the entry were not searched: what is stored there is stale
        const bool need = gi >= entry && gi < npos && len == kSearchCap && n - p > kSearchCap;
        const uint64_t needs = __builtin_amdgcn_ballot_w64(need);
        if (needs == 0ull) {
            rr[h] = r | ((len > kTokenMax ? len - kTokenMax : 0u) << 15);
            continue;
        }
        const uint32_t left = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane - 1u) << 2), (int)r);
        // the run goes on if the position before is complete the same way: 12 at the same offset
        const bool head = need && !(lane > 0u && ((needs >> (lane - 1u)) & 1ull) != 0ull && ((left ^ r) & 0x7FFFu) == 0u);
        uint32_t total = kSearchCap;
        if (head) {
            const uint32_t room = n - p < kRoom ? n - p : kRoom;
            for (;;) {
                uint32_t a0, a1, a2, b0, b1, b2;
                ringm_read12(L.ring, p + total, a0, a1, a2);
                ringm_read12(L.ring, p + total - off, b0, b1, b2);
                uint32_t e = lcp12(a0 ^ b0, a1 ^ b1, a2 ^ b2);
                e = e < 12u ? e : 12u;
                e = e < room - total ? e : room - total;
                total += e;
                if (e < 12u || total >= room) break;
            }
        }
        const uint64_t heads = __builtin_amdgcn_ballot_w64(head) & ((2ull << lane) - 1ull);
        const uint32_t hp = heads ? 63u - (uint32_t)__builtin_clzll(heads) : lane;   // head of my run
        const uint32_t th = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hp << 2), (int)total);
        uint32_t ext = len >= kTokenMax ? len - kTokenMax : 0u;
        if (need) ext = th >= kRoom ? kExtOpen : th - (lane - hp) - kTokenMax;
        rr[h] = r | (ext << 15);
    }
}

// One lane's SEARCH state.  It lives in registers across pools: a walk that is still going when
// its wave runs out of fresh positions is carried into the SEARCH of the next pool.
//
// The step is written for LATENCY, not instruction count: with 5 workgroups per CU a wave has at
// most four neighbours on its SIMD, and measured on gfx950 a lone wave pays 4 cycles per VALU or
// SALU instruction but ~28 for compare -> s_and -> select, ~35 for an exec-masked region and ~40
// for a branch on a vector compare (tools/probes/lat_probe.hip).  So there is no boolean algebra
// on lane masks and no branch inside a step: every predicate is one compare consumed by selects,
// conditions are folded into the data (an out-of-window candidate compares with length cap 0),
// the best match is one v_max over a packed key, and idle lanes run along harmlessly, storing to
// a dummy word.
struct Walk {
    uint32_t p, t0, t1, t2;      // position and its 12 bytes
    uint32_t nreach, myslot;     // ~(window reach): a distance d is inside iff -d > ~reach; link slot of p
    uint32_t nback;              // MINUS the distance of the candidate to look at next
    // best so far as (len << 16) - offset (signed compares): candidates come nearest first, so a
    // longer match wins and an equally long farther one does not.  0x20000 = "3-byte chain, nothing
    // yet" (only a length >= 3 beats it), 0x10000 = "2-byte chain, nothing yet".
    int32_t key;
    uint32_t cap;                // length that ends the walk: min(remaining, 12), or 2 on the 2-byte chain
    uint32_t nfirst2;            // MINUS the distance where the 2-byte chain starts (-kNoLink: nowhere)
    const uint16_t *links;       // link array being followed
    uint32_t *resp;              // where the result goes; the dummy word while the lane is idle
};
constexpr int32_t kKeyNone3 = 0x20000, kKeyNone2 = 0x10000;

// SEARCH: every wave pulls positions of [Pb, pend) from L.nextp.  A wave leaves when the pool has
// no fresh position left and none of its lanes still walks for a position before Pb (those belong
// to the pool that is parsed next); walks for positions of this pool may be left unfinished.
// The longest walk of a pool (about 30 steps in text, against 5.5 on average) then no longer
// holds up 255 other lanes at the end of every pool.  Results are stored as raw keys; EXTEND
// turns them into off | len << 11 | ext << 15.
#ifdef LZS_PROFILE
#define SEARCH_PROF_PARAMS , unsigned long long *prof_acc
#define SEARCH_PROF_ARGS , prof_acc
#else
#define SEARCH_PROF_PARAMS
#define SEARCH_PROF_ARGS
#endif
__device__ __forceinline__ void wg_search(BlkLds &L, Walk &W, uint32_t Pb, uint32_t pend, uint32_t n, uint32_t lane SEARCH_PROF_PARAMS)
{
    const uint32_t slot0 = wg_slot_base(Pb);
    uint32_t *const dummy = wg_dummy(L);
    bool pool_done = false;
    uint32_t p = W.p, t0 = W.t0, t1 = W.t1, t2 = W.t2, nreach = W.nreach, myslot = W.myslot;
    uint32_t nback = W.nback, cap = W.cap, nfirst2 = W.nfirst2;
    int32_t key = W.key;
    const uint16_t *links = W.links;
    uint32_t *resp = W.resp;
    for (;;) {
        const uint64_t idle = __builtin_amdgcn_ballot_w64(resp == dummy);
        const uint32_t nidle = (uint32_t)__builtin_popcountll(idle);
        if (!pool_done && nidle >= kRefillMin) {
            uint32_t basep = 0;
            if (lane == 0) basep = __hip_atomic_fetch_add(&L.nextp, nidle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            basep = uniform(basep);
            pool_done = basep + nidle >= pend;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                  __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t np = basep + rank;
            const bool take = resp == dummy && np < pend;
            PROF_COUNT(9, 1);
            PROF_COUNT(10, __builtin_popcountll(__builtin_amdgcn_ballot_w64(take)));
            PROF_T0;
            if (take) {                                                  // exec-masked: direct writes, no selects
                p = np;
                ringm_read12(L.ring, p, t0, t1, t2);
                const uint32_t before = ring_byte(L.ring, p - 1u);
                uint32_t sl = slot0 + (p - Pb);
                sl = sl >= kWgLinkN ? sl - kWgLinkN : sl;
                myslot = sl;
                const uint32_t l3 = L.link3[sl], l2 = L.link2[sl];
                const uint32_t lim = n - p < kSearchCap ? n - p : kSearchCap;
                // offset 1 first: common prefix of the text with itself shifted by one byte.  It
                // can only reach 2 if the byte before equals the next two, which is rare in text:
                // the 12-byte form is computed only when some lane needs it.
                const bool maybe1 = p >= 1u && ((t0 ^ ((t0 << 8) | before)) & 0xFFFFu) == 0;
                uint32_t len1 = 0;
                if (__builtin_amdgcn_ballot_w64(maybe1) != 0ull) {
                    len1 = lcp12(t0 ^ ((t0 << 8) | before), t1 ^ ((t1 << 8) | (t0 >> 24)), t2 ^ ((t2 << 8) | (t1 >> 24)));
                    len1 = len1 < lim ? len1 : lim;
                }
                const bool seeded = maybe1 && len1 >= 2u;
                const bool capped = seeded && len1 == lim;               // nothing nearer or longer exists
                const uint32_t reach = p < kWindow ? p : kWindow;
                nreach = ~reach;
                // chains that are empty inside the window are skipped here, not discovered by a step
                const bool walk3 = lim >= 3u && !capped && l3 <= reach;
                const bool walk2 = lim >= 2u && !capped && !seeded && l2 <= reach;
                nfirst2 = 0u - (walk2 ? l2 : kNoLink);
                nback = walk3 ? 0u - l3 : nfirst2;
                // 3-byte chain: only a longer match than the seed (or than 2) counts and the cap ends
                // the walk (:337-345); 2-byte chain: the first verified candidate is the answer
                key = seeded ? (int32_t)((len1 << 16) - 1u) : (walk3 ? kKeyNone3 : kKeyNone2);
                cap = walk3 ? lim : 2u;
                links = walk3 ? L.link3 : L.link2;
                uint32_t *const slot = &L.res[p & (kWgResN - 1)];
                *slot = (uint32_t)key;                                   // final if there is no chain to walk
                resp = (walk3 || walk2) ? slot : dummy;
            }
            PROF_T1(14);
        }
        if (pool_done && __builtin_amdgcn_ballot_w64(resp != dummy && p < Pb) == 0ull) break;
        PROF_COUNT(11, 1);
        PROF_COUNT(8 + 4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(resp != dummy)));
        PROF_T0B;
#pragma unroll
        for (int sub = 0; sub < LZS_SUBSTEPS; sub++) {
            // One step = one candidate, for all 64 lanes alike.
            const bool inwin = nback > nreach;
            const uint32_t nb = inwin ? nback : 0u;           // out of the window: look at p itself ...
            const uint32_t capx = inwin ? cap : 0u;           // ... and count none of it
            uint32_t w0, w1, w2;
            ringm_read12(L.ring, p + nb, w0, w1, w2);
            const uint32_t at = myslot + nb;
            const uint32_t nd = links[at < at + kWgLinkN ? at : at + kWgLinkN];
            uint32_t len = lcp12(w0 ^ t0, w1 ^ t1, w2 ^ t2);
            len = len < capx ? len : capx;
            const int32_t cand = (int32_t)((len << 16) + nb);
            key = key > cand ? key : cand;
            *resp = (uint32_t)key;
            // the walk ends at the cap, or where the chain leaves the window (capx = 0 = len)
            const int32_t kend = opaque(len == capx ? key : 0);
            // nothing on the 3-byte chain: restart on the 2-byte chain
            const bool fallback = kend == kKeyNone3;
            const int32_t kfin = opaque(fallback ? 0 : kend);
            key = fallback ? kKeyNone2 : key;
            cap = fallback ? 2u : cap;
            nback = fallback ? nfirst2 : nback - nd;
            links = fallback ? L.link2 : links;
            resp = kfin != 0 ? dummy : resp;                   // ended for good: the lane is idle
        }
        PROF_T1B(15);
    }
    W.p = p; W.t0 = t0; W.t1 = t1; W.t2 = t2; W.nreach = nreach; W.myslot = myslot;
    W.nback = nback; W.key = key; W.cap = cap; W.nfirst2 = nfirst2;
    W.links = links; W.resp = resp;
}

// What one workgroup compresses: the stream `src[0..n)` from token start `c0` up to (not
// including) the first token start >= `cend`.  A whole block: c0 = w0 = 0, cend = n.  A segment of
// a longer stream (lzs_compress_segments_kernel): the chains are first built from `w0` (a window
// before the segment), there is no end marker, and where the last token ends and how many bits
// were written is reported for the stitching.
struct WgJob {
    const uint8_t *src;
    uint32_t n, cend, c0, w0;
    bool last;                      // append the end marker
    uint32_t *out_len;              // bytes written (a block), or null
    uint32_t *exit_pos;             // a segment: first token start >= cend ...
    unsigned long long *nbits;      // ... and the bits emitted up to there; or null
    uint32_t *open_info;            // a piece of a stream that will go on (lzs_compress_incremental):
                                    // {offset, start} of the job's last token if that is a match
                                    // reaching the end of the data so far (n), {0, 0} otherwise; or null
};

__device__ __forceinline__ void wg_compress_job(BlkLds &L, const WgJob &job, WgOut o)
{
    const uint32_t tid  = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = uniform(tid >> 6);
    const uint8_t *src = job.src;
    const uint32_t n   = job.n;
    const uint32_t cend = job.cend;
    const uint32_t head0 = o.head;
    const bool src16   = ((uintptr_t)src & 15u) == 0;
    const bool src4    = ((uintptr_t)src & 3u) == 0;

    for (uint32_t i = tid; i < kWgHead3; i += kWgThreads) L.head3[i] = ~0u;
    for (uint32_t i = tid; i < kHead2; i += kWgThreads) L.head2[i] = ~0u;
    L.bits[tid] = 0;
    __syncthreads();

    if (job.open_info && tid == 0) { job.open_info[0] = 0; job.open_info[1] = 0; }
    uint32_t c = job.c0;         // start of the next token
    uint32_t loaded = job.w0;    // ring holds [loaded-4096, loaded)
    uint32_t next = job.w0;      // next batch of 64 positions to build
    PROF_DECL;

    // The pools are pipelined: while pool k is in SEARCH, pool k-1 (searched in the round before,
    // except for the walks carried over, which end in this round's SEARCH) waits for PARSE + PACK.
    Walk W;
    W.p = 0; W.t0 = W.t1 = W.t2 = 0; W.nreach = ~0u; W.myslot = 0;
    W.nback = 0u - kNoLink; W.key = 0; W.cap = 0; W.nfirst2 = 0u - kNoLink;
    W.links = L.link2; W.resp = wg_dummy(L);
    bool pending = false;                    // a searched pool waits for PARSE
    uint32_t Pb = 0, pend = 0;               // that pool
    const uint32_t nup = (cend + 63u) & ~63u;                 // no pool starts past the job's end

    for (;;) {
        if (o.flushed >= o.cap) break;
        if (!pending && (c >= cend || next >= nup)) break;
        PROF_MARK(4);
        // ---- the next pool [Sb, Se): REFILL (128 threads x 4 B per half KiB) and BUILD
        const bool fresh = next < nup;
        uint32_t Sb = next;
        if (!pending) while (Sb + 64 <= c) Sb += 64;          // batches wholly behind c: built, not searched
        const uint32_t Se = fresh ? (Sb + kWgPool < nup ? Sb + kWgPool : nup) : Sb;
        if (fresh) {
            // Not further ahead than needed: the walks carried over still read 2047 bytes
            // back from the pool before, and the ring holds 4096.
            const uint32_t need = (Se > c ? Se : c) + 96;
            while (loaded < n && loaded < need) {
                // half a KiB by 32 lanes of one wave (the waves take turns): the kernel is bound
                // by VALU issue, and four waves computing addresses for 4 bytes each cost four times this
                if (wave == ((loaded >> 9) & 3u) && lane < 32u) {
                    const uint32_t p = loaded + 16 * lane;
                    const uint4 v = load16(src, p, n, src16, src4);
                    const uint32_t at = (p & kRingMask) >> 2;
                    *reinterpret_cast<uint4 *>(&L.ring[at]) = v;
                    if (at == 0) *reinterpret_cast<uint4 *>(&L.ring[kRingWords]) = v;
                }
                loaded += kTile / 2;
            }
            __syncthreads();
            PROF_MARK(0);
            // Normally one pool; after an open match also the batches it ran over (at most the
            // 2112 positions that can still be candidates, see below), half a KiB at a time: the
            // HASH records live in the result slots, which are free then (no pool is pending).
            { PROF_T0; PROF_COUNT(17, (Se - next) >> 6);
              for (uint32_t R = next; R < Se; R += kWgPool) {
                  const uint32_t R2 = R + kWgPool < Se ? R + kWgPool : Se;
                  if (R != next) __syncthreads();
                  wg_hash_range(L, R, R2, n, lane, wave);
                  __syncthreads();
                  wg_chain_range(L, R, R2, lane, wave);
              }
              PROF_T1(16); }
            next = Se;
        }
        const uint32_t send = Se < cend ? Se : cend;             // (cend <= n; tokens start before it)
        if (tid == 0) L.nextp = (!pending && c > Sb) ? c : Sb;
        __syncthreads();
        PROF_MARK(1);

        // ---- SEARCH the new pool; ends every walk for the pending one
        wg_search(L, W, Sb, send, n, lane SEARCH_PROF_ARGS);
        PROF_MARK(2);
        __syncthreads();
        PROF_MARK(5);

        if (!pending) {                                        // nothing to parse yet
            Pb = Sb; pend = send; pending = fresh;
            continue;
        }

        // ---- PARSE + PACK; repeated after each open match that ends inside the pool.
        // The greedy chain of token starts is resolved hierarchically.  Inside each 64-position
        // chunk (two per wave, lane = position) pointer doubling over next[i] = i + bytes
        // consumed at i runs in registers (ds_bpermute, no barrier) and yields, for EVERY
        // possible entry position, where the chain leaves the chunk.  One short walk over the
        // 8 published exit functions then tells each chunk its actual entry, and lane m finds
        // the m-th token start of its chunk from the kept doubling tables -- so lane order is
        // token order and the bit offsets are a prefix sum.
        const uint32_t npos = pend - Pb;
        PROF_STAMP0;
        uint32_t rr[2];                                        // results of this thread's two positions, complete
        wg_extend(L, Pb, c - Pb, npos, n, lane, wave, rr);
        PROF_STAMP(20);
        while (c < pend && o.flushed < o.cap) {
            const uint32_t entry = c - Pb;
            // Written for latency (see Walk): the two chunks of a wave go through every
            // dependent cross-lane step side by side, predicates are single compares consumed
            // by selects, and nothing is exec-masked.
            uint32_t T[2][6], t[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t cb = 256u * h + 64u * wave;                 // pool-relative start of the chunk
                const uint32_t inside = cb < npos ? (npos - cb < 64u ? npos - cb : 64u) : 0u;   // its positions in the pool
                const uint32_t r = rr[h];
                const uint32_t len = (r >> 11) & 15u, ext = r >> 15;
                uint32_t step = len < kTokenMax ? len : kTokenMax + ext;
                step = len < 2u ? 1u : step;
                const uint32_t land = lane + step;
                // codes: < 64 next start inside the chunk; 0x100|j chain leaves at chunk offset j
                // (j >= 64, or the end of the pool); 0x200|i open match at chunk offset i
                uint32_t x = land < inside ? land : (0x100u | land);
                x = ext == kExtOpen ? (0x200u | lane) : x;                 // ext is 63 only for an open match
                x = lane < inside ? x : (0x100u | lane);
                t[h] = x;
            }
#pragma unroll
            for (int d = 0; d < 6; d++) {
                uint32_t via[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    T[h][d] = t[h];
                    via[h] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(t[h] << 2), (int)t[h]);
                }
#pragma unroll
                for (int h = 0; h < 2; h++) t[h] = t[h] < 64u ? via[h] : t[h];
            }
            // published per entry position: the pool-relative position the chain goes to next
            // (>= npos: the pool is done), or 0x8000 | the position of an open match
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t cb = 256u * h + 64u * wave;
                const uint32_t v = (t[h] & 0x200u) ? (0x8000u | (cb + (t[h] & 63u))) : cb + (t[h] & 0xFFu);
                L.exitfn[cb + lane] = (uint16_t)v;
            }
            PROF_STAMP(21);
            __syncthreads();
            PROF_STAMP(22);
            // complete quarters of the bit ring go out, one wave each: the bits of the round before
            // are all in (barrier above), and nothing is added before the next barrier
            while (o.head >= 2048u) {
                if (wave == ((o.flushed >> 8) & 3u)) wg_store_quarter(o, L, lane);
                o.flushed += 256u;
                o.head -= 2048u;
            }
            // follow the published exits from the entry, one hop per chunk, in every lane alike
            // (on the scalar unit: the kernel is VALU-bound, and this walk is the same in all four waves)
            uint32_t node[2] = {0x100u, 0x100u};
            uint32_t last = entry;
            while (last < npos) {
                const uint32_t k = last >> 6;
                if (k == wave) node[0] = last & 63u;
                if (k == wave + 4u) node[1] = last & 63u;
                last = uniform((uint32_t)L.exitfn[last]);
            }
            const uint32_t chain_end = (last & 0x8000u) ? kOpen : last;
            const uint32_t open_at = last & 0x7FFFu;
            PROF_STAMP(23);
            PROF_MARK(3);
            // ---- PACK: lane m takes the m-th token of its chunk (formats: lzs-compression.c:365-431)
#pragma unroll
            for (int d = 0; d < 6; d++) {
                uint32_t via[2];
#pragma unroll
                for (int h = 0; h < 2; h++)
                    via[h] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(node[h] << 2), (int)T[h][d]);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t on = node[h] < 64u ? via[h] : node[h];
                    node[h] = ((lane >> d) & 1u) ? on : node[h];
                }
            }
            uint32_t valhi[2], vallo[2], width[2], incl[2], rtok[2], btok[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                rtok[h] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(node[h] << 2), (int)rr[h]);
                btok[h] = ring_byte(L.ring, Pb + 256u * h + 64u * wave + (node[h] & 63u)) & 0xFFu;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t r = rtok[h];
                const uint32_t len = (r >> 11) & 15u, ext = r >> 15, off = r & kWindow;
                // match: offset field, first length code, extension nibbles 15 15 .. last
                const uint32_t ov = off <= kShortMax ? ((3u << 7) | off) : ((2u << 11) | off);
                const uint32_t ow = off <= kShortMax ? 9u : 13u;
                const uint32_t first = len < kTokenMax ? len : kTokenMax;
                const uint32_t lv = first <= 4u ? first - 2u : first + 7u;
                const uint32_t lw = first <= 4u ? 2u : 4u;
                const uint32_t full = (ext * 137u) >> 11;                  // ext / 15 for ext < 64
                uint32_t tv = (((1u << (4u * full)) - 1u) << 4) | (ext - kNibbleMax * full);
                uint32_t tw = 4u * (full + 1u);
                tw = len < kTokenMax ? 0u : tw;
                tv = len < kTokenMax ? 0u : tv;
                uint32_t hv = (ov << lw) | lv, hw = ow + lw;
                hv = len < 2u ? btok[h] : hv;                              // literal: 0 bbbbbbbb
                hw = len < 2u ? 9u : hw;
                // not a token of this round: no node, or the open match (finished by wave 0 below)
                uint32_t w = hw + tw;
                w = ext == kExtOpen ? 0u : w;
                w = node[h] < 64u ? w : 0u;
                const uint64_t v = w ? (((uint64_t)hv << tw) | tv) : 0ull;
                valhi[h] = (uint32_t)(v >> 32); vallo[h] = (uint32_t)v; width[h] = w;
                if (job.open_info) {                                       // (wave-uniform; null for whole blocks)
                    const uint32_t pos = Pb + 256u * h + 64u * wave + (node[h] & 63u);
                    if (w && len >= kTokenMax && pos + kTokenMax + ext == n) { job.open_info[0] = off; job.open_info[1] = pos; }
                }
                incl[h] = wave_inclusive_sum(w);
                if (lane == 63u) L.chunk_bits[4 * h + wave] = incl[h];     // (exitfn is live: no dummy stores here)
            }
            PROF_STAMP(24);
            __syncthreads();
            PROF_STAMP(25);
            // bits of the chunks before mine: a scan over the 8 chunk sums in lanes 0..7
            const uint32_t cbits = lane < 8u ? L.chunk_bits[lane & 7u] : 0u;
            const uint32_t cincl = wave_inclusive_sum(cbits);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)cincl, 7);
            const uint32_t cexcl = cincl - cbits;
            const uint32_t at0 = wg_bit_at(o);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)cexcl, (int)(4 * h + wave));
                const uint32_t at = (at0 + before + incl[h] - width[h]) & 8191u;
                // OR the token (<= 33 bits), MSB first, at bit `at`; a lane without one ORs zeros
                const uint64_t v = (((uint64_t)valhi[h] << 32) | vallo[h]) << ((64u - (at & 31u) - width[h]) & 63u);
                // zeros go to a word of the thread's own: many lanes ORing into one word serialize
                const uint32_t vh = (uint32_t)(v >> 32), vl = (uint32_t)v;
                const uint32_t d = at >> 5;
                const uint32_t dh = vh ? d : tid, dl = vl ? (d + 1u) & (kWgBitWords - 1) : tid;
                __hip_atomic_fetch_or(&L.bits[dh], vh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(&L.bits[dl], vl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            o.head += total;
            PROF_STAMP(26);

            PROF_MARK(6);
            if (chain_end != kOpen) {
                c = Pb + chain_end;                            // >= pend
            } else {
                // ---- open match at Pb + open_at: wave 0 finishes it alone (:417-431)
                __syncthreads();                               // every wave's bits of this round are in
                if (wave == 0) {
                    const uint32_t off = (0u - L.res[(Pb + open_at) & (kWgResN - 1)]) & kWindow;   // raw key: (len << 16) - offset
                    if (lane == 0) {
                        if (off <= kShortMax) bits_or(L.bits, kWgBitWords, wg_bit_at(o), (((3u << 7) | off) << 4) | 0xFu, 13);
                        else                  bits_or(L.bits, kWgBitWords, wg_bit_at(o), (((2u << 11) | off) << 4) | 0xFu, 17);
                    }
                    o.head += off <= kShortMax ? 13u : 17u;
                    c = Pb + open_at + kTokenMax;
                    // Up to 240 bytes = 16 nibbles per round: lane i compares the 4 bytes at
                    // c + 4i with those `off` before.  Nothing is inserted into the chains here;
                    // the batches run over are built with the next pool.
                    bool more = true;
                    while (more) {
                        while (o.head >= 2048u) { wg_store_quarter(o, L, lane); o.flushed += 256u; o.head -= 2048u; }
                        if (o.flushed >= o.cap) break;
                        wave_refill(L, src, n, src16, lane, loaded, c + 260);
                        const uint32_t rem = n - c;
                        const uint32_t span = rem < 240u ? rem : 240u;
                        const uint32_t q = c + 4u * lane;
                        const uint32_t aq = (q & kRingMask) >> 2, bq = ((q - off) & kRingMask) >> 2;
                        const uint32_t x = __builtin_amdgcn_alignbyte(L.ring[aq + 1], L.ring[aq], q) ^
                                           __builtin_amdgcn_alignbyte(L.ring[bq + 1], L.ring[bq], q - off);
                        uint32_t eq = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;           // equal leading bytes
                        const uint32_t mine = span > 4u * lane ? span - 4u * lane : 0u;    // bytes of the span in my word
                        eq = eq < mine ? eq : mine;
                        const uint64_t stops = __builtin_amdgcn_ballot_w64(eq < 4u);       // lanes >= 60 have mine = 0
                        const uint32_t l = uniform((uint32_t)__builtin_ctzll(stops));
                        const uint32_t m = 4u * l + (uint32_t)__builtin_amdgcn_readlane((int)eq, (int)l);   // equal bytes <= span
                        c += m;
                        const uint32_t full = m / kNibbleMax;
                        more = (m == 240u);
                        // `full` nibbles of 15, then (unless all 240 bytes matched and the run may
                        // go on) the closing nibble 0..14: up to 68 bits, 32 per lane from lane 0 on
                        const uint32_t ones = 4u * full, seq = more ? ones : ones + 4u;
                        if (lane < 3u) {
                            const uint32_t a = 32u * lane;
                            const uint32_t n1 = ones > a ? (ones - a < 32u ? ones - a : 32u) : 0u;     // ones in my piece
                            const bool closes = !more && ones >= a && ones < a + 32u;                  // the nibble is in my piece
                            const uint32_t width = n1 + (closes ? 4u : 0u);
                            uint32_t v = n1 >= 32u ? ~0u : (1u << n1) - 1u;
                            if (closes) v = (v << 4) | (m % kNibbleMax);
                            if (width) bits_or(L.bits, kWgBitWords, (wg_bit_at(o) + a) & 8191u, v, width);
                        }
                        o.head += seq;
                    }
                    // Positions the match ran over that can no longer be a candidate for anything
                    // (more than a window before c) are never inserted: chains only ever lead from
                    // built positions to older built ones, and every position searched from here
                    // on looks back at most 2047.
                    if (c > next + 2176u) next = (c - 2112u) & ~63u;
                    while (o.head >= 2048u) { wg_store_quarter(o, L, lane); o.flushed += 256u; o.head -= 2048u; }
                    if (job.open_info && lane == 0 && c == n) { job.open_info[0] = off; job.open_info[1] = Pb + open_at; }
                    if (lane == 0) {
                        L.bcast[0] = c; L.bcast[1] = loaded; L.bcast[2] = next;
                        L.bcast[3] = o.flushed; L.bcast[4] = o.head;
                    }
                }
                __syncthreads();
                c = L.bcast[0]; loaded = L.bcast[1]; next = L.bcast[2];
                o.flushed = L.bcast[3]; o.head = L.bcast[4];
                __syncthreads();
                PROF_MARK(7);
            }
        }
        PROF_COUNT(13, 1);
        // the pool just searched is parsed next, unless an open match ran past all of it
        Pb = Sb; pend = send; pending = fresh;
        if (pending && c >= pend) { pending = false; W.resp = wg_dummy(L); }
    }
    if (wave == 0) { PROF_DONE; }
    __syncthreads();                                           // the last round's bits are in

    if (wave == 0) {
        if (job.exit_pos && lane == 0) {
            *job.exit_pos = c;
            *job.nbits = 8ull * o.flushed + o.head - head0;        // without an end marker
        }
        // ---- end marker 1 1 0000000, zero pad to a byte, drain (:449-466)
        if (job.last) {
            if (lane == 0) bits_or(L.bits, kWgBitWords, wg_bit_at(o), 0x180u, 9);
            o.head = (o.head + 9u + 7u) & ~7u;
        }
        while (o.head >= 2048u) { wg_store_quarter(o, L, lane); o.flushed += 256u; o.head -= 2048u; }
        __builtin_amdgcn_wave_barrier();
        const uint32_t nbytes = (o.head + 7u) >> 3;                // a segment may end inside a byte
        if (o.ored) {
            if (32u * lane < o.head) wg_store_quarter(o, L, lane);     // the words the last bits reach into
        } else {
            for (uint32_t i = lane; i < nbytes; i += 64) {
                const uint32_t bit = ((o.flushed << 3) + 8 * i) & 8191u;
                const uint32_t v = (L.bits[bit >> 5] >> (24u - (bit & 24u))) & 0xFFu;
                if (o.flushed + i < o.limit) o.dst[o.flushed + i] = (uint8_t)v;
            }
        }
        const uint32_t total = o.flushed + nbytes;
        if (job.out_len && lane == 0) *job.out_len = total < o.cap ? total : o.cap;
    }
}

__global__ __launch_bounds__(kWgThreads)
void lzs_compress_blocks_wg_kernel(uint8_t *__restrict__ out, size_t out_stride, uint32_t out_cap,
                                   uint32_t *__restrict__ out_len,
                                   const uint8_t *__restrict__ in, size_t in_stride,
                                   const uint32_t *__restrict__ in_len, uint32_t in_len_uniform,
                                   uint32_t nblocks)
{
    __shared__ BlkLds L;
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    WgJob job;
    job.src = in + (size_t)b * in_stride;
    job.n = in_len ? in_len[b] : in_len_uniform;
    job.cend = job.n; job.c0 = 0; job.w0 = 0; job.last = true;
    job.out_len = out_len + b; job.exit_pos = nullptr; job.nbits = nullptr; job.open_info = nullptr;
    WgOut o;
    o.flushed = 0; o.head = 0;
    o.dst = out + (size_t)b * out_stride;
    o.cap = out_cap;
    o.aligned4 = ((uintptr_t)o.dst & 3u) == 0;
    o.limit = out_cap; o.ored = false;
    wg_compress_job(L, job, o);
}

// One long stream cut into segments of `seg` bytes (a multiple of 64), one workgroup each.  The
// search is a pure function of (input, position), so a segment only needs the 2047 bytes before it
// in its chains -- but where its first token starts, and at which bit its output begins, depends
// on the segment before.  Every segment writes its bits into a slot of its own, as if it began at
// bit 0; the host (lzs_host.c) enters every segment at its own start first, re-runs the segments
// whose predecessor turned out to end its last token elsewhere (`dirty`) until all entries agree,
// takes the prefix sum of the bit counts and has lzs_stitch_segments_kernel shift the slots into
// place.
__global__ __launch_bounds__(kWgThreads)
void lzs_compress_segments_kernel(uint8_t *__restrict__ slots, size_t slot_stride,
                                  const uint8_t *__restrict__ in, uint32_t n, uint32_t seg, uint32_t nseg,
                                  const uint32_t *__restrict__ entry, const uint8_t *__restrict__ dirty,
                                  uint32_t *__restrict__ exit_pos, unsigned long long *__restrict__ nbits,
                                  uint8_t *__restrict__ out, const unsigned long long *__restrict__ bit_at,
                                  uint32_t lim, uint32_t *__restrict__ open_info)
{
    __shared__ BlkLds L;
    const uint32_t k = blockIdx.x;
    if (k >= nseg) return;
    if (dirty && !dirty[k]) return;
    const uint32_t s = k * seg;
    // The last segment ends with the input -- or, for a piece of a stream that will go on
    // (lim < n), 15 bytes before it: a token is decided once 12 bytes of look-ahead are there,
    // a length nibble once 15 are (lzs-compression.c:641-647, 750-758).
    const uint32_t e = s + seg < lim ? s + seg : lim;
    WgJob job;
    job.src = in; job.n = n; job.cend = e;
    job.c0 = entry[k];
    // chains from a window before the first token: batches of 64, and one more so that HASH sees
    // the byte before the first position that matters
    const uint32_t c64 = job.c0 & ~63u;
    job.w0 = c64 > 2176u ? c64 - 2176u : 0u;
    job.last = false;
    job.out_len = nullptr; job.exit_pos = exit_pos + k; job.nbits = nbits + k;
    job.open_info = open_info ? open_info + 2 * k : nullptr;
    if (job.c0 >= e) {                                         // the segment before ran over all of this one
        if (threadIdx.x == 0) {
            exit_pos[k] = job.c0; nbits[k] = 0;
            if (open_info) { open_info[2 * k] = 0; open_info[2 * k + 1] = 0; }
        }
        return;
    }
    WgOut o;
    o.flushed = 0; o.cap = ~0u; o.aligned4 = true;
    if (!out) {
        // the usual case: the segment's bits fit its slot (they do unless a match runs on for more
        // than ~120 KB past the segment); if not, the stores stop there and the count goes on
        o.dst = slots + (size_t)k * slot_stride;               // 16-aligned by the host
        o.head = 0; o.limit = (uint32_t)slot_stride; o.ored = false;
    } else {
        // a segment that did not fit: once more, now that its bit offset is known, ORed straight
        // into the output (into the 256-byte granule its first bit falls in)
        const unsigned long long g = bit_at[k];
        o.dst = out + 256ull * (g >> 11);
        o.head = (uint32_t)(g & 2047ull); o.limit = ~0u; o.ored = true;
    }
    wg_compress_job(L, job, o);
}

// Segment k's bits (nbits[k] of them, MSB first from bit 0 of its slot) go to bit bit_at[k] of the
// stream; the last segment is followed by the end marker 1 1 0000000.  `out` is 4-aligned and
// zeroed: neighbours share their boundary words, so everything is ORed in.
__global__ __launch_bounds__(256)
void lzs_stitch_segments_kernel(uint8_t *__restrict__ out, const uint8_t *__restrict__ slots, size_t slot_stride,
                                const unsigned long long *__restrict__ bit_at,
                                const unsigned long long *__restrict__ nbits, uint32_t nseg, uint32_t end_marker)
{
    const uint32_t k = blockIdx.x;
    if (k >= nseg) return;
    const unsigned long long at = bit_at[k];
    const bool fits = nbits[k] <= 8ull * slot_stride;              // else its bits are ORed in directly
    const uint32_t nb = fits ? (uint32_t)nbits[k] : 0u;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(slots + (size_t)k * slot_stride);
    unsigned int *dst = reinterpret_cast<unsigned int *>(out) + (at >> 5);
    const uint32_t sh = (uint32_t)(at & 31ull);
    const uint32_t words = (nb + 31u) >> 5;
    for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) {
        uint32_t w = __builtin_bswap32(src[j]);                    // big-endian: stream bit 32j is the MSB
        if (j == words - 1u && (nb & 31u)) w &= ~0u << (32u - (nb & 31u));
        const uint32_t hi = w >> sh, lo = sh ? w << (32u - sh) : 0u;
        if (hi) atomicOr(dst + j, __builtin_bswap32(hi));
        if (lo) atomicOr(dst + j + 1, __builtin_bswap32(lo));
    }
    if (end_marker && k == nseg - 1u && threadIdx.x == 0) {        // end marker after the last bit
        const unsigned long long end = at + nbits[k];
        const uint32_t s2 = (uint32_t)(end & 31ull);
        const unsigned long long m = (unsigned long long)0x180u << (64u - 9u - s2);   // 9 bits, left-aligned after s2
        unsigned int *d2 = reinterpret_cast<unsigned int *>(out) + (end >> 5);
        atomicOr(d2, __builtin_bswap32((uint32_t)(m >> 32)));
        if ((uint32_t)m) atomicOr(d2 + 1, __builtin_bswap32((uint32_t)m));
    }
}

// A piece of a stream that begins inside a long match (lzs_compress_incremental: the call before
// ended with the match still running, state COMPRESS_EXTENDED of lzs-compression.c:750-776).  One
// wavefront measures how far in[c0..n) goes on repeating what lies `off` before it and writes the
// length nibbles at bit `bit0` of the zeroed, 4-aligned `out`: 1111 for every 15 bytes, then the
// closing nibble -- unless the run reaches the end of the data and more may follow (!last): then
// only the full 15s are written and the match stays open.  result: {next position, still open,
// bits written (low, high)}.
__global__ __launch_bounds__(64)
void lzs_extend_resume_kernel(uint8_t *__restrict__ out, uint32_t bit0, const uint8_t *__restrict__ in,
                              uint32_t n, uint32_t c0, uint32_t off, uint32_t last, uint32_t *__restrict__ result)
{
    const uint32_t lane = threadIdx.x;
    uint32_t c = c0;
    for (;;) {                                                     // 256 bytes a round, 4 per lane
        const uint32_t q = c + 4u * lane;
        uint32_t eq = 0;
        while (eq < 4u && q + eq < n && in[q + eq] == in[q + eq - off]) eq++;
        const uint64_t stops = __builtin_amdgcn_ballot_w64(eq < 4u);
        if (stops) {
            const uint32_t l = uniform((uint32_t)__builtin_ctzll(stops));
            c += 4u * l + (uint32_t)__builtin_amdgcn_readlane((int)eq, (int)l);
            break;
        }
        c += 256u;
    }
    const uint32_t run = c - c0;
    const bool open = c == n && !last;
    const uint32_t full = run / kNibbleMax;
    const unsigned long long ones_end = (unsigned long long)bit0 + 4ull * full;
    unsigned int *dst = reinterpret_cast<unsigned int *>(out);
    for (unsigned long long w = (bit0 >> 5) + lane; w <= (ones_end >> 5) && 4ull * full; w += 64) {
        const unsigned long long lo = w * 32ull > bit0 ? w * 32ull : bit0;
        const unsigned long long hi = (w + 1ull) * 32ull < ones_end ? (w + 1ull) * 32ull : ones_end;
        if (hi > lo) {
            const uint32_t a = (uint32_t)(lo - w * 32ull), b = (uint32_t)(hi - w * 32ull);    // bits [a, b) of the word, MSB first
            const uint32_t m = (~0u >> a) & (b == 32u ? ~0u : ~(~0u >> b));
            atomicOr(dst + w, __builtin_bswap32(m));
        }
    }
    if (!open && lane == 0) {
        const uint32_t s2 = (uint32_t)(ones_end & 31ull);
        const unsigned long long m = (unsigned long long)(run - kNibbleMax * full) << (64u - 4u - s2);
        unsigned int *d2 = dst + (ones_end >> 5);
        if ((uint32_t)(m >> 32)) atomicOr(d2, __builtin_bswap32((uint32_t)(m >> 32)));
        if ((uint32_t)m) atomicOr(d2 + 1, __builtin_bswap32((uint32_t)m));
    }
    if (lane == 0) {
        const unsigned long long bits = 4ull * full + (open ? 0ull : 4ull);
        result[0] = open ? c0 + kNibbleMax * full : c;
        result[1] = open ? 1u : 0u;
        result[2] = (uint32_t)bits; result[3] = (uint32_t)(bits >> 32);
    }
}

// ---------------------------------------------------------------------------------
// lzs_decompress() per block.  reference lzs-decompression.c:156-412
// One wave per stream: the token parse is wave-uniform, match copies are lane-parallel
// (lane i produces byte i of the copy; overlapping copies replicate with period `off`).
//   ring[4096]  the OUTPUT's sliding window, drained to HBM 1 KiB at a time;
//   stage[256]  (as 1 KiB with the ring's spare? no:) input is read straight from HBM
//               in 64-bit big-endian gulps through the scalar-friendly uniform path.
// ---------------------------------------------------------------------------------
struct __attribute__((aligned(16))) DecLds {
    uint32_t ring[kRingWords];     // output window
    uint32_t inbuf[kTile / 4];     // compressed input tile
};

__global__ __launch_bounds__(kWavesPerWG * 64)
void lzs_decompress_blocks_kernel(uint8_t *__restrict__ out, size_t out_stride, uint32_t out_cap,
                                  uint32_t *__restrict__ out_len,
                                  const uint8_t *__restrict__ in, size_t in_stride,
                                  const uint32_t *__restrict__ in_len, uint32_t in_len_uniform,
                                  uint32_t nblocks, uint32_t concat)
{
    __shared__ DecLds lds[kWavesPerWG];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv   = uniform(threadIdx.x >> 6);   // wave-uniform, and the compiler knows it
    const uint32_t b    = blockIdx.x * kWavesPerWG + wv;
    if (b >= nblocks) return;

    DecLds &L = lds[wv];
    uint8_t *ring8 = reinterpret_cast<uint8_t *>(L.ring);
    const uint8_t *src = in + (size_t)b * in_stride;
    const uint32_t n   = in_len ? in_len[b] : in_len_uniform;
    const bool src16   = ((uintptr_t)src & 15u) == 0;
    uint8_t *dst       = out + (size_t)b * out_stride;
    const bool dst16   = ((uintptr_t)dst & 15u) == 0;
    const uint32_t cap = out_cap;

    uint64_t bits = 0;        // left-aligned bit buffer
    uint32_t have = 0;        // valid bits in `bits`
    uint32_t ipos = 0;        // next input byte to feed (multiple of 4)
    uint32_t itile = 0;       // inbuf holds input [itile-1024, itile)
    uint32_t count = 0;       // bytes produced
    uint32_t flushed = 0;     // bytes stored to HBM (multiple of kTile)
    uint32_t off = 0;
    bool extended = false;

    for (;;) {
        // ---- refill (lzs-decompression.c:181-187): top up to > 32 bits while input lasts
        while (have <= 32 && ipos < n) {
            if (ipos >= itile) {
                const uint32_t p = itile + 16 * lane;
                *reinterpret_cast<uint4 *>(&L.inbuf[(p & (kTile - 1)) >> 2]) = load16(src, p, n, src16);
                itile += kTile;
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t w = uniform(__builtin_bswap32(L.inbuf[(ipos & (kTile - 1)) >> 2]));
            const uint32_t nb = n - ipos < 4 ? n - ipos : 4;     // bytes that really exist
            if (nb < 4) w &= ~0u << (8 * (4 - nb));
            bits |= (uint64_t)w << (32 - have);
            have += 8 * nb;
            ipos += 4;
        }
        if (have == 0 || count >= cap) break;                      // :189, :200

        uint32_t copy_len = 0;
        if (extended) {                                            // :370-406
            if (have < 4) break;
            const uint32_t e = (uint32_t)(bits >> 60);
            bits <<= 4; have -= 4;
            copy_len = e;
            if (e != kNibbleMax) extended = false;
        } else if ((bits >> 63) == 0 && have >= 9) {
            // a run of literals (:217-233), up to 7 at once: token i of an all-literal run starts
            // at bit 63 - 9i, so the first set type bit among those tells how long the run is
            const uint64_t types = bits & 0x8040201008040200ull;
            uint32_t k = (types ? (uint32_t)__builtin_clzll(types) : 64u) / 9u;
            k = k < have / 9u ? k : have / 9u;
            k = k < cap - count ? k : cap - count;
            if (lane < k) ring8[(count + lane) & kRingMask] = (uint8_t)(bits >> (55u - 9u * lane));
            count += k;
            bits <<= 9u * k; have -= 9u * k;
        } else if (have > 32 && (bits >> 63) != 0) {
            // a match token whose bits are all certainly there (at most 17 + 4 of more than 32):
            // same decoding as below without the per-field "enough bits left?" tests
            const uint32_t top = (uint32_t)(bits >> 43);           // 1 s ooooooo[oooo] cccc ...
            const bool is_short = (top >> 19) & 1u;
            const uint32_t o = is_short ? (top >> 12) & 0x7Fu : (top >> 8) & 0x7FFu;
            const uint32_t used = is_short ? 9u : 13u;
            if (o == 0) {
                bits <<= used; have -= used;
                if (is_short) {                                    // end marker (:255-260 / :564-576)
                    if (!concat) break;
                    const uint32_t pad = have & 7u;
                    bits <<= pad; have -= pad;
                } else {
                    off = 0;                                       // long offset 0: no copy (:280)
                }
                continue;
            }
            const uint32_t code = (is_short ? top >> 8 : top >> 4) & 0xFu;
            const uint32_t len = code < 0xC ? 2 + (code >> 2) : 5 + (code - 0xC);
            const uint32_t width = code < 0xC ? 2u : 4u;
            bits <<= used + width; have -= used + width;
            off = o;
            if (len == kTokenMax) extended = true;
            copy_len = len;
        } else {
            const uint32_t is_match = (uint32_t)(bits >> 63);
            bits <<= 1; have -= 1;
            if (!is_match) {                                       // literal :217-233
                if (have < 8) break;
                const uint32_t byte = (uint32_t)(bits >> 56);
                bits <<= 8; have -= 8;
                if (lane == 0) ring8[count & kRingMask] = (uint8_t)byte;
                count += 1;
            } else {
                if (have < 1) break;                               // :238-241
                const uint32_t is_short = (uint32_t)(bits >> 63);
                bits <<= 1; have -= 1;
                if (is_short) {                                    // :248-260
                    if (have < 7) break;
                    off = (uint32_t)(bits >> 57);
                    bits <<= 7; have -= 7;
                    if (off == 0) {                                // end marker
                        if (!concat) break;                        // one-shot rule: stop (:255-260)
                        // file rule (the incremental decoder, :564-576): drop the pad bits up
                        // to the byte boundary and go on with the next stream
                        const uint32_t pad = have & 7u;
                        bits <<= pad; have -= pad;
                        continue;
                    }
                } else {                                           // :272-279
                    if (have < 11) break;
                    off = (uint32_t)(bits >> 53);
                    bits <<= 11; have -= 11;
                }
                if (off != 0) {                                    // :280
                    const uint32_t code = (uint32_t)(bits >> 60);  // :103-120, :325-342
                    uint32_t len, width;
                    if (code < 0xC) { len = 2 + (code >> 2); width = 2; }
                    else            { len = 5 + (code - 0xC); width = 4; }
                    if (have < width) break;
                    bits <<= width; have -= width;
                    if (len == kTokenMax) extended = true;
                    copy_len = len;
                }
            }
        }

        if (copy_len) {                                            // :346-365, :381-400
            const uint32_t room = cap - count;
            const uint32_t m = copy_len < room ? copy_len : room;
            __builtin_amdgcn_wave_barrier();
            uint32_t v = 0;
            if (lane < m) {
                // overlap replicates with period `off`; m <= 15, so only short offsets wrap
                // (`off` is wave-uniform: the division is skipped for the common long offsets)
                const uint32_t k = off > 15u ? lane : lane % off;
                const uint32_t from = count + k;                   // position + off of the source
                v = from >= off ? ring8[(from - off) & kRingMask] : 0u;   // before out[0] -> 0
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < m) ring8[(count + lane) & kRingMask] = (uint8_t)v;
            count += m;
        }

        // ---- drain whole tiles of finished output
        while (count - flushed >= kTile) {
            __builtin_amdgcn_wave_barrier();
            const uint32_t p = flushed + 16 * lane;
            const uint4 v = *reinterpret_cast<const uint4 *>(&L.ring[(p & kRingMask) >> 2]);
            if (dst16) {
                *reinterpret_cast<uint4 *>(dst + p) = v;
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (uint32_t k = 0; k < 16; k++) dst[p + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            }
            flushed += kTile;
        }
        if (count >= cap) break;                                   // mid-copy stop :361-364
    }

    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = flushed + lane; i < count; i += 64) dst[i] = ring8[i & kRingMask];
    if (lane == 0) out_len[b] = count;
}

// ---------------------------------------------------------------------------------
// lzs_decompress() per block, second version (the default).  Same rules, same wave-per-stream
// shape; what changed is where the instructions go.  The first version spent 29 scalar
// instructions per output byte and saturated the CU's one scalar unit (rocprofv3: 3.1e10 SALU per
// GiB = 93 % of its issue slots) -- its compressed input went HBM -> LDS tile -> ds_read ->
// v_readfirstlane, and its conditions were combined as lane masks.  Here the compressed stream is
// read by plain word loads one word ahead of use (no LDS tile), the token decode is nested single
// compares, and the fields of a match token are extracted on the (idle) vector unit and come back
// packed through one v_readfirstlane.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kWavesPerWG * 64)
void lzs_decompress_blocks_v2_kernel(uint8_t *__restrict__ out, size_t out_stride, uint32_t out_cap,
                                     uint32_t *__restrict__ out_len,
                                     const uint8_t *__restrict__ in, size_t in_stride,
                                     const uint32_t *__restrict__ in_len, uint32_t in_len_uniform,
                                     uint32_t nblocks, uint32_t concat)
{
    __shared__ uint32_t rings[kWavesPerWG][kRingWords];           // the OUTPUT's sliding window
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv   = uniform(threadIdx.x >> 6);
    const uint32_t b    = blockIdx.x * kWavesPerWG + wv;
    if (b >= nblocks) return;

    uint32_t *ring = rings[wv];
    uint8_t *ring8 = reinterpret_cast<uint8_t *>(ring);
    const uint8_t *src = in + (size_t)b * in_stride;
    const uint32_t n   = uniform(in_len ? in_len[b] : in_len_uniform);
    uint8_t *dst       = out + (size_t)b * out_stride;
    const bool dst16   = ((uintptr_t)dst & 15u) == 0;
    const uint32_t cap = out_cap;

    // ---- the input as aligned words: word j holds stream bytes [4j - skew, 4j - skew + 4)
    const uint32_t skew = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(src - skew);
    const uint32_t nwords = n ? (skew + n + 3u) >> 2 : 0u;         // words that hold stream bytes
    uint64_t bits = 0;        // left-aligned bit buffer
    uint32_t have = 0;        // valid bits in `bits`
    uint32_t pos  = 0;        // stream bytes fed so far
    uint32_t wi   = 0;        // next word to feed
    uint32_t nextw = 0;       // that word, loaded ahead of use
    if (nwords) {
        uint32_t w = __builtin_bswap32(w32[0]) << (8u * skew);
        const uint32_t avail = 4u - skew < n ? 4u - skew : n;
        if (avail < 4u) w &= ~0u << (32u - 8u * avail);
        bits = (uint64_t)w << 32;
        have = 8u * avail;
        pos = avail;
        wi = 1;
        if (nwords > 1u) nextw = w32[1];
    }
    uint32_t count = 0;       // bytes produced
    uint32_t flushed = 0;     // bytes stored to HBM (multiple of kTile)
    uint32_t off = 0;
    uint32_t extended = 0;

    for (;;) {
        // ---- refill (lzs-decompression.c:181-187).  One word per token is enough: no token path
        // below takes more than 32 bits except a run of literals, which takes what is there.
        if (have <= 32u) {
            if (pos < n) {
                uint32_t w = __builtin_bswap32(nextw);
                const uint32_t rem = n - pos;
                const uint32_t nb = rem < 4u ? rem : 4u;           // bytes that really exist
                if (rem < 4u) w &= ~0u << (32u - 8u * rem);
                bits |= (uint64_t)w << (32u - have);
                have += 8u * nb;
                pos += nb;
                wi += 1u;
                if (wi < nwords) nextw = w32[wi];
            }
        }
        if (have == 0u) break;                                     // :189
        if (count >= cap) break;                                   // :200, and mid-copy :361-364
        const uint32_t room = cap - count;

        uint32_t copy_len = 0;
        const uint32_t top = (uint32_t)(bits >> 32);
        if (extended) {                                            // :370-406
            if (have < 4u) break;
            const uint32_t e = top >> 28;
            bits <<= 4; have -= 4u;
            copy_len = e;
            extended = e == kNibbleMax ? 1u : 0u;
        } else if ((int32_t)top >= 0) {
            // a run of literals (:217-233), up to 7 at once: token i of an all-literal run starts
            // at bit 63 - 9i, so the first set type bit among those tells how long the run is
            if (have < 9u) break;                                  // type bit, then 8 more or stop (:220-223)
            // (counted on the vector unit, like the match fields below)
            const uint32_t th = opaque(top) & 0x80402010u, tl = opaque((uint32_t)bits) & 0x08040200u;
            const uint32_t lead = th ? (uint32_t)__builtin_clz(th) : (tl ? 32u + (uint32_t)__builtin_clz(tl) : 64u);
            uint32_t kv = (lead * 57u) >> 9;                       // lead / 9 for lead <= 64
            const uint32_t fitv = (opaque(have) * 57u) >> 9;       // have / 9 for have <= 64
            kv = kv < fitv ? kv : fitv;
            kv = kv < room ? kv : room;
            const uint32_t k = uniform(kv);
            if (lane < kv) ring8[(count + lane) & kRingMask] = (uint8_t)(bits >> (55u - 9u * lane));
            count += k;
            bits <<= 9u * k; have -= 9u * k;
        } else {
            // a match token: 1 s ooooooo[oooo] cccc (:238-342).  Every field is decoded from the
            // zero-padded buffer without asking whether its bits exist; the ONE test on `need`
            // covers all the "not enough bits: stop" exits of the reference (:240,250,274,334),
            // because a token produces nothing before its last field is read, and bits can only
            // be missing when the input is exhausted (the refill above keeps more than a token's
            // worth otherwise).
            // field extraction on the vector unit (the scalar unit is the bottleneck): the values
            // are the same in every lane and come back through v_readfirstlane
            const uint32_t t = opaque(top) >> 11;
            const bool is_short_v = (t >> 19) & 1u;
            const uint32_t o_v = is_short_v ? (t >> 12) & 0x7Fu : (t >> 8) & 0x7FFu;
            const uint32_t used_v = is_short_v ? 9u : 13u;
            const uint32_t code_v = (is_short_v ? t >> 8 : t >> 4) & 0xFu;
            const uint32_t len_v = code_v < 0xCu ? 2u + (code_v >> 2) : code_v - 7u;
            const uint32_t width_v = code_v < 0xCu ? 2u : 4u;
            // packed: o (11) | used (4) << 11 | len (4) << 15 | width (3) << 19 | is_short << 22
            const uint32_t packed = uniform(o_v | (used_v << 11) | (len_v << 15) | (width_v << 19) | ((is_short_v ? 1u : 0u) << 22));
            const uint32_t o = packed & 0x7FFu, used = (packed >> 11) & 15u;
            const bool is_short = (packed >> 22) & 1u;
            if (o == 0u) {
                if (have < used) break;
                bits <<= used; have -= used;
                if (is_short) {                                    // end marker (:255-260 / :564-576)
                    if (!concat) break;                            // one-shot rule: stop
                    // file rule (the incremental decoder): drop the pad bits up to the byte
                    // boundary and go on with the next stream
                    const uint32_t pad = have & 7u;
                    bits <<= pad; have -= pad;
                } else {
                    off = 0;                                       // long offset 0: no copy (:280)
                }
                continue;
            }
            const uint32_t len = (packed >> 15) & 15u, width = (packed >> 19) & 7u;
            if (have < used + width) break;
            bits <<= used + width; have -= used + width;
            off = o;
            extended = len == kTokenMax ? 1u : 0u;
            copy_len = len;
        }

        if (copy_len) {                                            // :346-365, :381-400
            const uint32_t m = copy_len < room ? copy_len : room;
            __builtin_amdgcn_wave_barrier();
            uint32_t v = 0;
            if (lane < m) {
                // overlap replicates with period `off`; m <= 15, so only short offsets wrap
                // (`off` is wave-uniform: the division is skipped for the common long offsets)
                const uint32_t k = off > 15u ? lane : lane % off;
                const uint32_t from = count + k;                   // position + off of the source
                v = from >= off ? ring8[(from - off) & kRingMask] : 0u;   // before out[0] -> 0
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < m) ring8[(count + lane) & kRingMask] = (uint8_t)v;
            count += m;
        }

        // ---- drain whole tiles of finished output
        if (count - flushed >= kTile) {
            __builtin_amdgcn_wave_barrier();
            const uint32_t p = flushed + 16 * lane;
            const uint4 v = *reinterpret_cast<const uint4 *>(&ring[(p & kRingMask) >> 2]);
            if (dst16) {
                *reinterpret_cast<uint4 *>(dst + p) = v;
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (uint32_t k = 0; k < 16; k++) dst[p + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            }
            flushed += kTile;
        }
    }

    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = flushed + lane; i < count; i += 64) dst[i] = ring8[i & kRingMask];
    if (lane == 0) out_len[b] = count;
}

// ---------------------------------------------------------------------------------
// ONE long stream decompressed by many wavefronts (the counterpart of the segment compressor).
// The compressed stream is cut into segments of `seg` bytes (kDecSeg below), one wave each.  What a segment
// cannot know by itself is the decoder's state when its bit cursor first crosses into it: the bit
// where the first token starts, whether an extension is running and at which offset.  So:
//   SCAN    every segment walks its tokens without copying anything, entered at its first bit in
//           the normal state, and reports the state in which it leaves and how many bytes it
//           would produce; the host re-runs the segments whose predecessor left in another state
//           until all agree (walks entered at different bits fall in step after some tokens), and
//           takes the prefix sum of the byte counts;
//   DECODE  every segment decodes into the output at its offset.  A copy whose source lies before
//           the segment's own output cannot be done yet -- that part of the output is being
//           produced by another wave -- so every byte carries an ORIGIN: clean, or the output
//           position it is a copy of.  Origins are copied along with the bytes;
//   RESOLVE rounds of pointer jumping over the origins (a byte whose origin is clean takes its
//           value, otherwise it adopts its origin's origin) until none is left.
// Stop rules as in lzs_decompress_blocks_v2_kernel, one-shot form (the first end marker ends it).
// ---------------------------------------------------------------------------------
constexpr uint32_t kDecSegMax = 8192;                 // compressed bytes per segment of a long stream (a wave walks it in
                                                      // ~1 ms); short streams take smaller ones (the host chooses `seg`)
constexpr uint32_t kClean    = 0xFFFFFFFFu;           // origin: the byte is final
constexpr uint32_t kDoneBase = 0xFFFFFF00u;           // origin: resolved in round (value & 0xFF)
constexpr uint32_t kSegStop  = 1u << 30;              // state word: the stream ended in this segment
constexpr uint32_t kScanMarkWords = 132;              // per segment: 64 marks, 64 byte counts, exit state, total, pad
// state word: bits 0..7 cursor past the segment start (bits), bit 8 extension running, 9..19 offset

// In LDS an origin is how far BEFORE the segment's output it lies (1..2047: a copy can only reach
// that far, and copies of copies inherit it), 0 = clean: 16 bits per byte of the window.
struct DecSegLds {
    uint32_t ring[kRingWords];                        // the segment's own output window
    uint16_t origin[kRingWords * 4];                  // per byte of it
};

template <bool DECODE>
__device__ __forceinline__ void lzs_stream_segment(const uint8_t *__restrict__ in, uint32_t n, uint32_t seg_byte, uint32_t entry,
                                                   uint32_t *exit_out, uint32_t *count_out,
                                                   uint8_t *out, uint32_t cap, uint32_t out_start, uint32_t out_floor,
                                                   uint32_t *origin_g, uint32_t *tainted_total,
                                                   DecSegLds *Lp, uint32_t lane, uint32_t kDecSeg, bool concat,
                                                   uint32_t *marks = nullptr, bool compare = false)
{
    const uint32_t kDecEnd = 8u * kDecSeg;                        // the segment's length in bits
    // SCAN only.  A walk that is repeated from another entry falls in step with the walk before
    // after a few dozen tokens, and from there on it IS that walk: a full walk leaves behind its
    // 64 of its token starts (state word and bytes produced so far: `marks`), and a repeated one
    // stops as soon as it stands on one of them -- it leaves as the full walk left, with that
    // walk's byte count from there on.  (Otherwise every round costs a walk over the whole
    // segment.)  The marks are the first 16 steps of the walk, then every 2nd, 4th, .. 64th, eight
    // of each: two walks in step take the same steps, so the repeated one meets the next mark
    // within that many, and the marks reach a thousand steps into the segment.
    uint32_t my_mark = ~0u, my_count = 0, nmark = 0, iter = 0, next_mark = 0, mark_step = 1;
    uint32_t old_mark = ~0u, old_count = 0, old_last = 0;
    bool checking = false, merged = false;
    if (!DECODE && compare) {
        old_mark = marks[lane]; old_count = marks[64u + lane];
        const uint64_t valid = __builtin_amdgcn_ballot_w64(old_mark != ~0u);
        if (valid) {
            checking = true;
            old_last = (uint32_t)__builtin_amdgcn_readlane((int)old_mark, 63 - (int)__builtin_clzll(valid)) & 0xFFFFu;
        }
    }
    uint8_t *ring8 = DECODE ? reinterpret_cast<uint8_t *>(Lp->ring) : nullptr;
    uint16_t *origin = DECODE ? Lp->origin : nullptr;
    const uint32_t rel0 = entry & 0xFFu;
    const uint32_t byte0 = seg_byte + (rel0 >> 3);              // (the segment starts at in[seg_byte], its stream ends at in[n])
    uint32_t count = 0, flushed = 0, tainted = 0;
    uint32_t off = (entry >> 9) & 0x7FFu;
    uint32_t extended = (entry >> 8) & 1u;
    uint32_t state = kSegStop;
    if (byte0 < n) {
        const uint8_t *src = in + byte0;
        const uint32_t nn = n - byte0;
        // the input as aligned words (as in lzs_decompress_blocks_v2_kernel)
        const uint32_t skew = (uint32_t)((uintptr_t)src & 3u);
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(src - skew);
        const uint32_t nwords = (skew + nn + 3u) >> 2;
        uint64_t bits; uint32_t have, pos, wi = 1, nextw = 0;
        {
            uint32_t w = __builtin_bswap32(w32[0]) << (8u * skew);
            const uint32_t avail = 4u - skew < nn ? 4u - skew : nn;
            if (avail < 4u) w &= ~0u << (32u - 8u * avail);
            bits = (uint64_t)w << 32; have = 8u * avail; pos = avail;
            if (nwords > 1u) nextw = w32[1];
        }
        bits <<= rel0 & 7u; have -= rel0 & 7u;                   // the first token starts inside the byte
        const uint32_t base = 8u * (rel0 >> 3);
        for (;;) {
            if (have <= 32u) {
                if (pos < nn) {
                    uint32_t w = __builtin_bswap32(nextw);
                    const uint32_t rem = nn - pos;
                    const uint32_t nb = rem < 4u ? rem : 4u;
                    if (rem < 4u) w &= ~0u << (32u - 8u * rem);
                    bits |= (uint64_t)w << (32u - have);
                    have += 8u * nb; pos += nb; wi += 1u;
                    if (wi < nwords) nextw = w32[wi];
                }
            }
            const uint32_t cur = base + 8u * pos - have;         // bits past the segment start
            if (cur >= kDecEnd) {                                // the next token belongs to the next segment
                // (the offset is part of the state only while an extension runs: otherwise two walks
                // that fell in step would still look different to the host)
                state = (cur - kDecEnd) | (extended << 8) | ((extended ? off : 0u) << 9);
                break;
            }
            if (!DECODE && marks) {
                const uint32_t word = cur | (extended << 16) | ((extended ? off : 0u) << 17);
                if (checking) {
                    const uint64_t hit = __builtin_amdgcn_ballot_w64(old_mark == word);
                    if (hit) {
                        const uint32_t j = uniform((uint32_t)__builtin_ctzll(hit));
                        state = marks[128];
                        count += marks[129] - (uint32_t)__builtin_amdgcn_readlane((int)old_count, (int)j);
                        merged = true;
                        break;
                    }
                    if (cur > old_last) checking = false;
                }
                if (nmark < 64u && iter == next_mark) {
                    my_mark = lane == nmark ? word : my_mark;
                    my_count = lane == nmark ? count : my_count;
                    if (nmark >= 15u && ((nmark - 15u) & 7u) == 0u) mark_step <<= 1;
                    next_mark += mark_step;
                    nmark++;
                }
                iter++;
            }
            if (have == 0u) break;                               // input exhausted (:189)
            if (DECODE && out_start + count >= cap) break;       // output full (:200)
            uint32_t copy_len = 0;
            const uint32_t top = (uint32_t)(bits >> 32);
            if (extended) {                                      // :370-406
                if (have < 4u) break;
                // nibbles of 15 (a long match is thousands of them) go up to four at a time: 60 bytes
                const uint32_t ones = (~bits ? (uint32_t)__builtin_clzll(~bits) : 64u) >> 2;   // leading 1111 nibbles
                uint32_t run = ones < have / 4u ? ones : have / 4u;
                const uint32_t mine = (kDecEnd - cur + 3u) / 4u;                // nibbles that start in this segment
                run = run < mine ? run : mine;
                run = run < 4u ? run : 4u;
                if (run) {
                    copy_len = kNibbleMax * run;
                    bits <<= 4u * run; have -= 4u * run;
                } else {
                    copy_len = top >> 28;                        // the closing nibble, 0..14
                    bits <<= 4; have -= 4u;
                    extended = 0;
                }
            } else if ((int32_t)top >= 0) {
                // a run of literals, but only those that start inside this segment
                if (have < 9u) break;
                const uint64_t types = bits & 0x8040201008040200ull;
                uint32_t kk = (types ? (uint32_t)__builtin_clzll(types) : 64u) / 9u;
                const uint32_t fit = have / 9u;
                const uint32_t mine = (kDecEnd - cur + 8u) / 9u;
                kk = kk < fit ? kk : fit;
                kk = kk < mine ? kk : mine;
                if (DECODE) {
                    if (lane < kk) {
                        const uint32_t at = (count + lane) & kRingMask;
                        ring8[at] = (uint8_t)(bits >> (55u - 9u * lane));
                        origin[at] = 0;
                    }
                }
                count += kk;
                bits <<= 9u * kk; have -= 9u * kk;
            } else {
                // 1 s ooooooo[oooo] cccc: fields on the vector unit (see lzs_decompress_blocks_v2_kernel)
                const uint32_t t = opaque(top) >> 11;
                const bool is_short_v = (t >> 19) & 1u;
                const uint32_t o_v = is_short_v ? (t >> 12) & 0x7Fu : (t >> 8) & 0x7FFu;
                const uint32_t used_v = is_short_v ? 9u : 13u;
                const uint32_t code_v = (is_short_v ? t >> 8 : t >> 4) & 0xFu;
                const uint32_t len_v = code_v < 0xCu ? 2u + (code_v >> 2) : code_v - 7u;
                const uint32_t width_v = code_v < 0xCu ? 2u : 4u;
                const uint32_t packed = uniform(o_v | (used_v << 11) | (len_v << 15) | (width_v << 19) | ((is_short_v ? 1u : 0u) << 22));
                const uint32_t o = packed & 0x7FFu, used = (packed >> 11) & 15u;
                const bool is_short = (packed >> 22) & 1u;
                if (o == 0u) {
                    if (have < used) break;
                    bits <<= used; have -= used;
                    if (is_short) {
                        if (!concat) break;                      // end marker: the stream ends here (:255-260)
                        const uint32_t pad = have & 7u;          // file rule (:564-576): on from the next byte
                        bits <<= pad; have -= pad;
                        continue;
                    }
                    off = 0;                                     // long offset 0: no copy (:280)
                    continue;
                }
                const uint32_t len = (packed >> 15) & 15u, width = (packed >> 19) & 7u;
                if (have < used + width) break;
                bits <<= used + width; have -= used + width;
                off = o;
                extended = len == kTokenMax ? 1u : 0u;
                copy_len = len;
            }
            if (copy_len) {
                if (DECODE) {
                    __builtin_amdgcn_wave_barrier();
                    uint32_t v = 0, og = 0;
                    if (lane < copy_len) {
                        // overlap replicates with period `off` (copy_len <= 60)
                        const uint32_t kk = off >= 60u ? lane : (off ? lane % off : 0u);
                        const uint32_t from = count + kk;        // position + off of the source
                        if (from >= off) {                       // inside this segment's own output
                            const uint32_t at = (from - off) & kRingMask;
                            v = ring8[at]; og = origin[at];
                        } else if ((unsigned long long)out_start + from >= (unsigned long long)off + out_floor) {
                            og = off - from;                     // produced by another wave: that far before my output
                        }                                        // else before the stream's out[0] (out_floor): zero (:350-357)
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < copy_len) {
                        const uint32_t at = (count + lane) & kRingMask;
                        ring8[at] = (uint8_t)v; origin[at] = (uint16_t)og;
                    }
                }
                count += copy_len;
            }
            if (DECODE) {
                if (count - flushed >= kTile) {                  // a finished KiB goes out
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t j = 0; j < kTile; j += 64u) {
                        const uint32_t p = flushed + j + lane;
                        const uint32_t g = out_start + p;
                        const uint32_t og = origin[p & kRingMask];
                        if (g < cap && g >= out_start) { out[g] = ring8[p & kRingMask]; origin_g[g] = og ? out_start - og : kClean; tainted += og != 0u; }
                    }
                    flushed += kTile;
                }
            }
        }
    }
    if (DECODE) {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t p = flushed + lane; p < count; p += 64u) {
            const uint32_t g = out_start + p;
            const uint32_t og = origin[p & kRingMask];
            if (g < cap && g >= out_start) { out[g] = ring8[p & kRingMask]; origin_g[g] = og ? out_start - og : kClean; tainted += og != 0u; }
        }
        if (tainted) atomicAdd(tainted_total, tainted);
    } else {
        if (marks && !merged) {
            marks[lane] = my_mark; marks[64u + lane] = my_count;
            if (lane == 0) { marks[128] = state; marks[129] = count; }
        }
        if (lane == 0) {
            *exit_out = state;
            *count_out = count;
        }
    }
}

__global__ __launch_bounds__(256)
void lzs_scan_stream_kernel(const uint8_t *__restrict__ in, uint32_t n, uint32_t nseg,
                            const uint32_t *__restrict__ entry, const uint8_t *__restrict__ dirty,
                            uint32_t *__restrict__ exit_state, uint32_t *__restrict__ count,
                            uint8_t *__restrict__ all_ones, uint32_t *__restrict__ marks, uint32_t compare,
                            uint32_t kDecSeg, uint32_t concat,
                            const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_end)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + uniform(threadIdx.x >> 6);
    if (k >= nseg) return;
    // one stream cut into equal segments -- or many streams in one buffer (a batch of blocks), each
    // cut into segments of its own: then the tables say where segment k starts and its stream ends
    const uint32_t sbase = seg_base ? uniform(seg_base[k]) : k * kDecSeg;
    if (seg_end) n = uniform(seg_end[k]);
    if (all_ones) {
        // A whole segment of 0xFF bytes inside a running extension is nothing but nibbles of 15:
        // it leaves in the state in which it was entered.  The host uses this to carry a long
        // match across its segments without a round each.
        bool ones = (size_t)sbase + kDecSeg <= n;
        if (ones) {
            const uint8_t *p = in + sbase;
            for (uint32_t i = lane; i < kDecSeg && ones; i += 64u) ones = p[i] == 0xFFu;
        }
        const bool all = __builtin_amdgcn_ballot_w64(!ones) == 0ull;
        // 2: and so are the first three bits after it (the last nibble that starts in the segment
        // may reach that far into the next one)
        const size_t after = (size_t)sbase + kDecSeg;
        if (lane == 0) all_ones[k] = !all ? 0 : (after < n && (in[after] & 0xE0u) == 0xE0u) ? 2 : 1;
    }
    if (dirty && !dirty[k]) return;
    lzs_stream_segment<false>(in, n, sbase, uniform(entry[k]), exit_state + k, count + k,
                              nullptr, 0, 0, 0, nullptr, nullptr, nullptr, lane, kDecSeg, concat != 0u,
                              marks ? marks + (size_t)k * kScanMarkWords : nullptr, compare != 0u);
}

__global__ __launch_bounds__(256)
void lzs_decode_stream_kernel(uint8_t *__restrict__ out, uint32_t cap, uint32_t *__restrict__ origin_g,
                              uint32_t *__restrict__ tainted_total,
                              const uint8_t *__restrict__ in, uint32_t n, uint32_t nseg,
                              const uint32_t *__restrict__ entry, const uint32_t *__restrict__ out_start,
                              uint32_t kDecSeg, uint32_t concat,
                              const uint32_t *__restrict__ seg_base, const uint32_t *__restrict__ seg_end,
                              const uint32_t *__restrict__ out_floor, const uint32_t *__restrict__ out_limit)
{
    __shared__ DecSegLds lds[4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = uniform(threadIdx.x >> 6);
    const uint32_t k = blockIdx.x * 4u + wv;
    if (k >= nseg) return;
    const uint32_t e = uniform(entry[k]);
    if (e & kSegStop) return;                                 // the stream ended before this segment
    const uint32_t sbase = seg_base ? uniform(seg_base[k]) : k * kDecSeg;
    if (seg_end) n = uniform(seg_end[k]);
    if (out_limit) cap = uniform(out_limit[k]);                // where this segment's stream must stop writing
    lzs_stream_segment<true>(in, n, sbase, e, nullptr, nullptr, out, cap, uniform(out_start[k]),
                             out_floor ? uniform(out_floor[k]) : 0u,
                             origin_g, tainted_total, &lds[wv], lane, kDecSeg, concat != 0u);
}

// One round of pointer jumping over the origins (each launch only trusts what earlier launches
// finished: a byte resolved in this round is marked with the round number and becomes a source in
// the next one).  `left` counts the bytes still open after the round.
__global__ __launch_bounds__(256)
void lzs_resolve_stream_kernel(uint8_t *__restrict__ out, uint32_t *__restrict__ origin_g, uint32_t total,
                               uint32_t round, uint32_t *__restrict__ left)
{
    uint32_t open = 0;
    for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < total; p += gridDim.x * 256u) {
        const uint32_t o = origin_g[p];
        if (o >= kDoneBase) continue;                             // final, or resolved earlier
        const uint32_t oo = origin_g[o];
        if (oo == kClean || (oo >= kDoneBase && (oo & 0xFFu) < round)) {
            out[p] = out[o];
            origin_g[p] = kDoneBase | round;
        } else if (oo < kDoneBase) {
            // two jumps a round: the origin's origin may be final already, or lead further back
            const uint32_t ooo = origin_g[oo];
            if (ooo == kClean || (ooo >= kDoneBase && (ooo & 0xFFu) < round)) {
                out[p] = out[oo];
                origin_g[p] = kDoneBase | round;
            } else {
                origin_g[p] = ooo < kDoneBase ? ooo : oo;         // adopt the farthest origin that is still open
                open++;
            }
        } else {
            open++;                                               // the origin was resolved in this very round: next time
        }
    }
    if (open) atomicAdd(left, open);
}

// The same for a batch of blocks (lzs_decompress_batch of a small batch: every block cut into
// segments): origins never leave their block, so one workgroup per block jumps pointers until its
// block is done, with barriers instead of launches between the rounds.  In place: a byte whose
// origin is clean takes its value and becomes clean itself (value first, then the mark: release /
// acquire at workgroup scope), any other adopts its origin's origin -- whichever of the two a
// neighbour reads meanwhile is a valid origin of that byte.
__global__ __launch_bounds__(1024)
void lzs_resolve_blocks_kernel(uint8_t *__restrict__ out, uint32_t *__restrict__ origin_g, size_t out_stride,
                               const uint32_t *__restrict__ len)
{
    const uint32_t b = blockIdx.x;
    const uint32_t n = len[b];
    const uint32_t base = (uint32_t)(b * out_stride);
    for (;;) {
        int pending = 0;
        for (uint32_t p = base + threadIdx.x; p < base + n; p += 1024u) {
            const uint32_t o = origin_g[p];
            if (o == kClean) continue;
            const uint32_t oo = __hip_atomic_load(&origin_g[o], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (oo == kClean) {
                out[p] = out[o];
                __hip_atomic_store(&origin_g[p], kClean, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                origin_g[p] = oo;
                pending = 1;
            }
        }
        if (!__syncthreads_or(pending)) break;
    }
}

// ---------------------------------------------------------------------------------
// Compaction of fixed-stride slots into one dense string.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void lzs_scan_lengths_kernel(uint64_t *__restrict__ offsets, const uint32_t *__restrict__ len,
                             uint32_t nblocks)
{
    // single workgroup: each thread sums a contiguous chunk, then a block-wide scan of the sums
    __shared__ uint64_t partial[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblocks + 1023u) / 1024u;
    const uint32_t lo = t * per < nblocks ? t * per : nblocks;
    const uint32_t hi = lo + per < nblocks ? lo + per : nblocks;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += len[i];
    partial[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t add = t >= d ? partial[t - d] : 0;
        __syncthreads();
        partial[t] += add;
        __syncthreads();
    }
    uint64_t run = partial[t] - sum;     // exclusive prefix of this chunk
    for (uint32_t i = lo; i < hi; i++) { offsets[i] = run; run += len[i]; }
    if (t == 1023) offsets[nblocks] = partial[1023];
}

__global__ __launch_bounds__(256)
void lzs_gather_slots_kernel(uint8_t *__restrict__ dense, const uint64_t *__restrict__ offsets,
                             const uint8_t *__restrict__ slots, size_t slot_stride,
                             const uint32_t *__restrict__ len, uint32_t nblocks)
{
    const uint32_t b = blockIdx.x;
    if (b >= nblocks) return;
    const uint8_t *s = slots + (size_t)b * slot_stride;
    uint8_t *d = dense + offsets[b];
    const uint32_t n = len[b];
    // head bytes until d is 4-aligned, then dst-aligned words assembled from two aligned
    // source words, then tail bytes
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u);
    const uint32_t h = head < n ? head : n;
    if (threadIdx.x < h) d[threadIdx.x] = s[threadIdx.x];
    const uint32_t words = (n - h) >> 2;
    const uint8_t *sb = s + h;
    const uint32_t shift = (uint32_t)((uintptr_t)sb & 3u);
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(sb - shift);
    uint32_t *dw = reinterpret_cast<uint32_t *>(d + h);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
        const uint32_t lo = sw[i];
        const uint32_t hi = shift ? sw[i + 1] : 0u;
        dw[i] = __builtin_amdgcn_alignbyte(hi, lo, shift);
    }
    const uint32_t done = h + 4 * words;
    if (threadIdx.x < n - done) d[done + threadIdx.x] = s[done + threadIdx.x];
}

// ---------------------------------------------------------------------------------
// lzs_decompress_incremental(): one call's worth of decoding, resumable.
// reference lzs-decompression.c:459-743 (a 9-state machine over a 32-bit queue; here the same
// stop rules at token granularity).  One wavefront.  `st` carries what the reference keeps in
// LzsDecompressParameters_t between calls: the bits of an unfinished token, the copy in progress
// (offset, bytes left, extended), and the last <= 2047 bytes of output as history.  The call
// stops when the input runs out (INPUT_STARVED, with INPUT_FINISHED if no bit is left), when the
// next byte has no room in `out` (NO_OUTPUT_BUFFER_SPACE, also in the middle of a copy), or after
// an end marker (END_MARKER: the pad bits up to the byte boundary are dropped, :564-576, and
// history is kept for what follows).  Whole unread input bytes are handed back (in_used).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void lzs_decode_resume_kernel(lzs_dec_resume_t *__restrict__ st, const uint8_t *__restrict__ in, uint32_t n,
                              uint8_t *__restrict__ out, uint32_t cap)
{
    __shared__ uint32_t ring[kRingWords];
    uint8_t *ring8 = reinterpret_cast<uint8_t *>(ring);
    const uint32_t lane = threadIdx.x;
    const uint32_t base = uniform(st->hist_len);                  // ring position of out[0]
    for (uint32_t i = lane; i < base; i += 64) ring8[i] = st->hist[i];
    __builtin_amdgcn_wave_barrier();

    uint64_t bits = (uint64_t)uniform(st->bitq) << 32;            // left-aligned
    uint32_t have = uniform(st->qlen);
    const uint32_t carried = have;                                // bits that are not from `in`
    uint32_t off = uniform(st->off), rem = uniform(st->rem);
    bool extended = uniform(st->extended) != 0;
    uint32_t ipos = 0;                                            // next input byte to feed (multiple of 4)
    uint32_t count = base, flushed = base;
    const uint32_t limit = base + cap;
    uint32_t status = 0;
    // The compressed input, 256 bytes (a word per lane) at a time and one batch ahead of use: a
    // dependent global load per word would cost its full latency every 4 bytes.
    const uint32_t *in32 = reinterpret_cast<const uint32_t *>(in);   // staged 4-aligned by the host
    const uint32_t nwords = (n + 3u) >> 2;
    uint32_t batch = 0;
    uint32_t cur = lane < nwords ? in32[lane] : 0u;
    uint32_t nxt = 64u + lane < nwords ? in32[64u + lane] : 0u;

    for (;;) {
        while (have <= 32 && ipos < n) {
            const uint32_t wi = ipos >> 2;
            if ((wi >> 6) != batch) {
                batch = wi >> 6;
                cur = nxt;
                const uint32_t at = 64u * (batch + 1u) + lane;
                nxt = at < nwords ? in32[at] : 0u;
            }
            uint32_t w = __builtin_bswap32((uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(wi & 63u)));
            const uint32_t nb = n - ipos < 4 ? n - ipos : 4;
            if (nb < 4) w &= ~0u << (8 * (4 - nb));
            bits |= (uint64_t)w << (32 - have);
            have += 8 * nb;
            ipos += 4;
        }
        // No bit left: the reference stops here whatever it was doing (:475-478, :492-496) -- also
        // with a copy pending, which then waits for the next call that brings input.
        if (have == 0) { status |= LZS_INC_INPUT_FINISHED | LZS_INC_INPUT_STARVED; break; }
        if (rem) {                                                 // :640-704
            const uint32_t room = limit - count;
            if (room == 0) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }
            const uint32_t m = rem < room ? rem : room;
            __builtin_amdgcn_wave_barrier();
            uint32_t v = 0;
            if (lane < m) {
                const uint32_t k = off > 15u ? lane : lane % off;
                const uint32_t from = count + k;
                v = from >= off ? ring8[(from - off) & kRingMask] : 0u;   // before the history's start -> 0 (:676-683)
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < m) ring8[(count + lane) & kRingMask] = (uint8_t)v;
            count += m;
            rem -= m;
        } else {
            bool starved = false;
            if (extended) {                                        // :706-723
                if (have < 4) starved = true;
                else {
                    const uint32_t e = (uint32_t)(bits >> 60);
                    bits <<= 4; have -= 4;
                    rem = e;
                    if (e != kNibbleMax) extended = false;
                }
            } else if ((bits >> 63) == 0) {                        // literals :516-541, up to 7 at once
                if (have < 9) starved = true;
                else if (count >= limit) { status |= LZS_INC_NO_OUTPUT_SPACE; break; }
                else {
                    const uint64_t types = bits & 0x8040201008040200ull;
                    uint32_t k = (types ? (uint32_t)__builtin_clzll(types) : 64u) / 9u;
                    k = k < have / 9u ? k : have / 9u;
                    k = k < limit - count ? k : limit - count;
                    if (lane < k) ring8[(count + lane) & kRingMask] = (uint8_t)(bits >> (55u - 9u * lane));
                    count += k;
                    bits <<= 9u * k; have -= 9u * k;
                }
            } else {                                               // offset, then length or end marker
                const bool is_short = ((bits >> 62) & 1u) != 0;
                const uint32_t used = is_short ? 9u : 13u;
                if (have < used) starved = true;
                else {
                    const uint32_t o = is_short ? (uint32_t)(bits >> 55) & 0x7Fu : (uint32_t)(bits >> 51) & 0x7FFu;
                    if (o == 0) {
                        bits <<= used; have -= used;
                        if (is_short) {                            // end marker :564-576
                            const uint32_t pad = have & 7u;
                            bits <<= pad; have -= pad;
                            status |= LZS_INC_END_MARKER;
                            break;
                        }
                        off = 0;                                   // long offset 0: no copy (one-shot rule, :280)
                    } else {
                        const uint32_t code = (uint32_t)((bits << used) >> 60);
                        const uint32_t width = code < 0xCu ? 2u : 4u;
                        if (have < used + width) starved = true;
                        else {
                            const uint32_t len = code < 0xCu ? 2u + (code >> 2) : 5u + (code - 0xCu);
                            bits <<= used + width; have -= used + width;
                            off = o;
                            rem = len;
                            extended = len == kTokenMax;
                        }
                    }
                }
            }
            if (starved) { status |= LZS_INC_INPUT_STARVED; break; }     // the token's bits stay queued
        }
        if (count - flushed >= kTile) {
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = flushed + lane; i < count; i += 64) out[i - base] = ring8[i & kRingMask];
            flushed = count;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = flushed + lane; i < count; i += 64) out[i - base] = ring8[i & kRingMask];

    // whole bytes of this call's input that were not needed go back to the caller
    const uint32_t fed = ipos < n ? ipos : n;
    const uint32_t consumed = carried + 8u * fed - have;          // bits used up in this call
    const uint32_t fed_left = consumed >= carried ? have : 8u * fed;
    // (a starved call keeps the unfinished token's bits, < 17, and takes all the input, as the
    // reference does: its callers read more only when inLength is 0)
    const uint32_t back = (status & LZS_INC_INPUT_STARVED) ? 0u : fed_left >> 3;
    have -= 8u * back;
    const uint32_t hist_len = count < kWindow ? count : kWindow;
    for (uint32_t i = lane; i < hist_len; i += 64) st->hist[i] = ring8[(count - hist_len + i) & kRingMask];
    if (lane == 0) {
        st->bitq = (uint32_t)(bits >> 32) & (have ? ~0u << (32u - have) : 0u);
        st->qlen = have;
        st->off = off; st->rem = rem; st->extended = extended ? 1u : 0u;
        st->hist_len = hist_len;
        st->in_used = fed - back;
        st->out_made = count - base;
        st->status = status;
    }
}

}  // namespace

// =====================================================================================
// extern "C" shim (see lzs_hip_shim.h)
// =====================================================================================
extern "C" {

int lzs_hip_device_count(int *count) { return (int)hipGetDeviceCount(count); }

const char *lzs_hip_strerror(int e) { return hipGetErrorString((hipError_t)e); }

int lzs_hip_describe(char *buf, size_t cap)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    snprintf(buf, cap, "hip device %d: %s (%s), %d CUs, %.0f GiB, LDS/CU %zu KiB; kernels: workgroup-per-block LZS compress, wave-per-stream decompress (gfx950)",
             dev, p.name, p.gcnArchName, p.multiProcessorCount,
             (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0),
             (size_t)p.maxSharedMemoryPerMultiProcessor / 1024);
    return 0;
}

int lzs_hip_malloc(void **p, size_t bytes) { return (int)hipMalloc(p, bytes ? bytes : 1); }
int lzs_hip_free(void *p) { return (int)hipFree(p); }
int lzs_hip_stream_create(void **s) { return (int)hipStreamCreateWithFlags((hipStream_t *)s, hipStreamNonBlocking); }
int lzs_hip_stream_destroy(void *s) { return (int)hipStreamDestroy((hipStream_t)s); }
int lzs_hip_stream_sync(void *s) { return (int)hipStreamSynchronize((hipStream_t)s); }
int lzs_hip_h2d(void *d, const void *s, size_t n, void *st)
{
    return n ? (int)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st) : 0;
}
int lzs_hip_d2h(void *d, const void *s, size_t n, void *st)
{
    return n ? (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st) : 0;
}
int lzs_hip_memset(void *d, int v, size_t n, void *st)
{
    return n ? (int)hipMemsetAsync(d, v, n, (hipStream_t)st) : 0;
}

int lzs_hip_launch_compress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                            const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                            uint32_t in_len, uint32_t nblocks, void *stream)
{
    if (nblocks == 0) return 0;
    const uint32_t grid = (nblocks + kWavesPerWG - 1) / kWavesPerWG;
    // LZS_KERNEL selects a variant for A/B runs and cross-checks: "wg" (default, one
    // workgroup per block), "chain" (one wave per block), "scan" (brute force)
    static const int variant = [] {
        const char *v = getenv("LZS_KERNEL");
        return !v ? 0 : (v[0] == 's' ? 2 : (v[0] == 'c' ? 1 : 0));
    }();
    if (variant == 2)
        hipLaunchKernelGGL(lzs_compress_blocks_scan_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
    else if (variant == 1)
        hipLaunchKernelGGL(lzs_compress_blocks_kernel, dim3(nblocks), dim3(64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
    else
        hipLaunchKernelGGL(lzs_compress_blocks_wg_kernel, dim3(nblocks), dim3(kWgThreads), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
    return (int)hipGetLastError();
}

static int launch_decompress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                             const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                             uint32_t in_len, uint32_t nblocks, void *stream, uint32_t concat)
{
    if (nblocks == 0) return 0;
    const uint32_t grid = (nblocks + kWavesPerWG - 1) / kWavesPerWG;
    static const int use_v1 = [] { const char *v = getenv("LZS_DECODER"); return v && v[0] == 'v' && v[1] == '1'; }();
    if (use_v1)
        hipLaunchKernelGGL(lzs_decompress_blocks_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, concat);
    else
        hipLaunchKernelGGL(lzs_decompress_blocks_v2_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, concat);
    return (int)hipGetLastError();
}

int lzs_hip_launch_decompress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                              const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                              uint32_t in_len, uint32_t nblocks, void *stream)
{
    return launch_decompress(d_out, out_stride, out_cap, d_out_len, d_in, in_stride, d_in_len, in_len, nblocks, stream, 0);
}

int lzs_hip_launch_decompress_concat(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                                     const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                                     uint32_t in_len, uint32_t nblocks, void *stream)
{
    return launch_decompress(d_out, out_stride, out_cap, d_out_len, d_in, in_stride, d_in_len, in_len, nblocks, stream, 1);
}

int lzs_hip_launch_compress_segments(void *d_slots, size_t slot_stride, const void *d_in, uint32_t n,
                                     uint32_t seg, uint32_t nseg, const uint32_t *d_entry,
                                     const uint8_t *d_dirty, uint32_t *d_exit, uint64_t *d_nbits,
                                     void *d_out, const uint64_t *d_bit_at, uint32_t lim, uint32_t *d_open,
                                     void *stream)
{
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(lzs_compress_segments_kernel, dim3(nseg), dim3(kWgThreads), 0, (hipStream_t)stream,
                       (uint8_t *)d_slots, slot_stride, (const uint8_t *)d_in, n, seg, nseg,
                       d_entry, d_dirty, d_exit, (unsigned long long *)d_nbits,
                       (uint8_t *)d_out, (const unsigned long long *)d_bit_at, lim, d_open);
    return (int)hipGetLastError();
}

int lzs_hip_launch_extend_resume(void *d_out, uint32_t bit0, const void *d_in, uint32_t n, uint32_t c0,
                                 uint32_t off, int last, uint32_t *d_result, void *stream)
{
    hipLaunchKernelGGL(lzs_extend_resume_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, bit0, (const uint8_t *)d_in, n, c0, off, last ? 1u : 0u, d_result);
    return (int)hipGetLastError();
}

int lzs_hip_launch_stitch_segments(void *d_out, const void *d_slots, size_t slot_stride,
                                   const uint64_t *d_bit_at, const uint64_t *d_nbits, uint32_t nseg,
                                   int end_marker, void *stream)
{
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(lzs_stitch_segments_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, (const uint8_t *)d_slots, slot_stride,
                       (const unsigned long long *)d_bit_at, (const unsigned long long *)d_nbits, nseg,
                       end_marker ? 1u : 0u);
    return (int)hipGetLastError();
}

int lzs_hip_launch_scan_stream(const void *d_in, uint32_t n, uint32_t nseg, const uint32_t *d_entry,
                               const uint8_t *d_dirty, uint32_t *d_exit, uint32_t *d_count,
                               uint8_t *d_all_ones, uint32_t *d_marks, int compare, uint32_t seg, int concat,
                               const uint32_t *d_seg_base, const uint32_t *d_seg_end, void *stream)
{
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(lzs_scan_stream_kernel, dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)d_in, n, nseg, d_entry, d_dirty, d_exit, d_count, d_all_ones,
                       d_marks, compare ? 1u : 0u, seg, concat ? 1u : 0u, d_seg_base, d_seg_end);
    return (int)hipGetLastError();
}

int lzs_hip_launch_decode_stream(void *d_out, uint32_t cap, uint32_t *d_origin, uint32_t *d_tainted,
                                 const void *d_in, uint32_t n, uint32_t nseg, const uint32_t *d_entry,
                                 const uint32_t *d_out_start, uint32_t seg, int concat,
                                 const uint32_t *d_seg_base, const uint32_t *d_seg_end,
                                 const uint32_t *d_out_floor, const uint32_t *d_out_limit, void *stream)
{
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(lzs_decode_stream_kernel, dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, cap, d_origin, d_tainted, (const uint8_t *)d_in, n, nseg, d_entry, d_out_start, seg, concat ? 1u : 0u,
                       d_seg_base, d_seg_end, d_out_floor, d_out_limit);
    return (int)hipGetLastError();
}

int lzs_hip_launch_resolve_blocks(void *d_out, uint32_t *d_origin, size_t out_stride, const uint32_t *d_len,
                                  uint32_t nblocks, void *stream)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(lzs_resolve_blocks_kernel, dim3(nblocks), dim3(1024), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, d_origin, out_stride, d_len);
    return (int)hipGetLastError();
}

int lzs_hip_launch_resolve_stream(void *d_out, uint32_t *d_origin, uint32_t total, uint32_t round,
                                  uint32_t *d_left, void *stream)
{
    if (total == 0) return 0;
    uint32_t grid = (total + 255u) / 256u;
    if (grid > 65536u) grid = 65536u;
    hipLaunchKernelGGL(lzs_resolve_stream_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, d_origin, total, round, d_left);
    return (int)hipGetLastError();
}

unsigned lzs_hip_dec_segment_bytes(void) { return kDecSegMax; }

int lzs_hip_launch_compact(void *d_dense, uint64_t *d_offsets, const void *d_slots,
                           size_t slot_stride, const uint32_t *d_len, uint32_t nblocks, void *stream)
{
    hipLaunchKernelGGL(lzs_scan_lengths_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       d_offsets, d_len, nblocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || nblocks == 0) return (int)e;
    hipLaunchKernelGGL(lzs_gather_slots_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_dense, d_offsets, (const uint8_t *)d_slots, slot_stride, d_len,
                       nblocks);
    return (int)hipGetLastError();
}


int lzs_hip_launch_decode_resume(lzs_dec_resume_t *d_state, const void *d_in, uint32_t n,
                                 void *d_out, uint32_t cap, void *stream)
{
    hipLaunchKernelGGL(lzs_decode_resume_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       d_state, (const uint8_t *)d_in, n, (uint8_t *)d_out, cap);
    return (int)hipGetLastError();
}

}  // extern "C"
