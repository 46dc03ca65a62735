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
__device__ __forceinline__ uint4 load16(const uint8_t *src, uint32_t p, uint32_t n, bool aligned16)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p + 16 <= n && aligned16) {
        v = *reinterpret_cast<const uint4 *>(src + p);
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
    const uint32_t wv   = threadIdx.x >> 6;
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
// lzs_compress() per block, variant "chain" (the default).
//
// The search rule is a pure function of (input, position), so it is hoisted out of the
// serial parse and made position-parallel: the wave takes 64 consecutive positions at a
// time (lane = position) and
//   1. BUILD   inserts them into a previous-occurrence chain keyed by a hash of the two
//              bytes at the position -- only offsets whose first two bytes match can give
//              a match >= 2, so the chain is a complete candidate list (the reference's
//              hash chains, lzs-compression.c:328-361,435-443, exploit the same fact);
//   2. SEARCH  every lane walks its own chain nearest-first, comparing 12 bytes per
//              candidate, keeping the first strictly longer match, stopping at the cap or
//              when the chain leaves the 2047-byte window (lzs-compression.c:334-361);
//   3. PARSE   the wave then runs the greedy token loop over those 64 results, reading
//              each token's (length, offset) from the owning lane, and packs bits
//              (lzs-compression.c:365-431).
// Positions swallowed by a long match are built but not searched.
//
// Chain storage, per wave, in LDS:
//   head[2048]  low 16 bits of the latest position per hash (stale/aliased entries only
//               ever add byte-verified candidates at increasing distance: harmless);
//   link[2112]  per position (ring of 33 batches): distance to the previous position
//               with the same hash.
// Several lanes of one batch may share a hash; their order is resolved exactly by an
// in-wave bitonic sort of (hash, lane), not by relying on LDS write-conflict order.
// ---------------------------------------------------------------------------------
constexpr uint32_t kHashBits  = 11;
constexpr uint32_t kHeads     = 1u << kHashBits;
constexpr uint32_t kLinkSlots = 2112;            // 33 x 64 >= 2047 + 64

struct __attribute__((aligned(16))) ChainLds {
    uint32_t ring[kRingWords + 4];               // +16 B mirror of ring[0..15]: reads never wrap
    uint16_t head[kHeads];
    uint16_t link[kLinkSlots];
    uint32_t stage[kStage / 4];
};

__device__ __forceinline__ void ringm_read12(const uint32_t *ring, uint32_t q,
                                             uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
    const uint32_t a = (q & kRingMask) >> 2;     // words a..a+3 exist thanks to the mirror
    const uint32_t s = q & 3;
    const uint32_t d0 = ring[a], d1 = ring[a + 1], d2 = ring[a + 2], d3 = ring[a + 3];
    w0 = __builtin_amdgcn_alignbyte(d1, d0, s);
    w1 = __builtin_amdgcn_alignbyte(d2, d1, s);
    w2 = __builtin_amdgcn_alignbyte(d3, d2, s);
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

// Ascending bitonic sort of one value per lane across the wave.
__device__ __forceinline__ uint32_t wave_sort(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v, (int)j, 64);
            const bool up = (lane & k) == 0 || k == 64;
            const bool low = (lane & j) == 0;
            const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
            v = (low == up) ? mn : mx;
        }
    }
    return v;
}

// BUILD for the batch starting at position B (multiple of 64).  Returns this lane's 12
// bytes (t0..t2) and its chain distance (0 = no previous position with this hash).
__device__ __forceinline__ uint32_t chain_build(ChainLds &L, uint32_t B, uint32_t n, uint32_t lane,
                                                uint32_t &t0, uint32_t &t1, uint32_t &t2)
{
    const uint32_t p = B + lane;
    ringm_read12(L.ring, p, t0, t1, t2);
    const bool has2 = p + 1 < n;                              // a 2-gram starts here
    const uint32_t gram = t0 & 0xFFFFu;
    const uint32_t h = has2 ? ((gram * 40503u) >> 5) & (kHeads - 1) : kHeads + lane;
    // nearest lower lane with the same hash, and whether this lane is the last of its hash
    const uint32_t sorted = wave_sort((h << 6) | lane, lane);
    const uint32_t before = (uint32_t)__shfl_up((int)sorted, 1, 64);
    const uint32_t after  = (uint32_t)__shfl_down((int)sorted, 1, 64);
    const bool same = lane > 0 && (before >> 6) == (sorted >> 6);
    const bool last = lane == 63 || (after >> 6) != (sorted >> 6);
    const uint32_t note = (same ? (0x40u | (before & 63u)) : 0u) | (last ? 0x80u : 0u);
    // hand the note back to the lane that owns the position
    const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_permute((int)((sorted & 63u) << 2), (int)note);

    const uint32_t old = has2 ? L.head[h] : 0u;
    uint32_t dist;
    if (mine & 0x40u) dist = lane - (mine & 63u);
    else              dist = (p - old) & 0xFFFFu;
    if (!has2) dist = 0;
    const uint32_t slot = ((B >> 6) % 33u) * 64u + lane;
    L.link[slot] = (uint16_t)dist;
    __builtin_amdgcn_wave_barrier();
    if (has2 && (mine & 0x80u)) L.head[h] = (uint16_t)p;
    __builtin_amdgcn_wave_barrier();
    return dist;
}

__global__ __launch_bounds__(kWavesPerWG * 64)
void lzs_compress_blocks_kernel(uint8_t *__restrict__ out, size_t out_stride, uint32_t out_cap,
                                uint32_t *__restrict__ out_len,
                                const uint8_t *__restrict__ in, size_t in_stride,
                                const uint32_t *__restrict__ in_len, uint32_t in_len_uniform,
                                uint32_t nblocks)
{
    __shared__ ChainLds lds[kWavesPerWG];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv   = threadIdx.x >> 6;
    const uint32_t b    = blockIdx.x * kWavesPerWG + wv;
    if (b >= nblocks) return;

    ChainLds &L = lds[wv];
    const uint8_t *src = in + (size_t)b * in_stride;
    const uint32_t n   = in_len ? in_len[b] : in_len_uniform;
    const bool src16   = ((uintptr_t)src & 15u) == 0;

    Sink s;
    s.acc = 0; s.nbits = 0; s.fill = 0; s.flushed = 0;
    s.dst = out + (size_t)b * out_stride;
    s.cap = out_cap;
    s.aligned4 = ((uintptr_t)s.dst & 3u) == 0;

    // never-set heads must look far away: 0x8000 is > 2047 from every early position, and a
    // later alias is only a byte-verified extra candidate
    for (uint32_t i = lane; i < kHeads / 2; i += 64)
        reinterpret_cast<uint32_t *>(L.head)[i] = 0x80008000u;
    __builtin_amdgcn_wave_barrier();

    uint32_t c = 0;          // start of the next token
    uint32_t loaded = 0;     // ring holds [loaded-4096, loaded)
    uint32_t next = 0;       // next batch to build (multiple of 64)

    while (c < n) {
        if (s.flushed >= s.cap) break;                        // output full (:306-309)
        const uint32_t B = next;
        chain_refill(L, src, n, src16, lane, loaded, (B > c ? B : c) + 80);
        uint32_t t0, t1, t2;
        uint32_t dist = chain_build(L, B, n, lane, t0, t1, t2);
        next = B + 64;
        if (c >= next) continue;                              // batch lies inside a match

        // ---- SEARCH (lane = position): lzs-compression.c:322-363
        const uint32_t p = B + lane;
        const uint32_t lim = p < n ? (n - p < kSearchCap ? n - p : kSearchCap) : 0u;
        const uint32_t reach = p < kWindow ? p : kWindow;
        const uint32_t myslot = ((B >> 6) % 33u) * 64u + lane;
        bool walking = p >= c && lim >= 2;
        uint32_t best_len = 0, best_off = 0, cum = 0;
        for (;;) {
            cum += dist;
            walking = walking && dist != 0 && cum <= reach;
            if (!__any(walking)) break;
            if (walking) {
                uint32_t w0, w1, w2;
                ringm_read12(L.ring, p - cum, w0, w1, w2);
                const uint32_t e0 = eq_bytes(w0 ^ t0);
                const uint32_t e1 = eq_bytes(w1 ^ t1);
                const uint32_t e2 = eq_bytes(w2 ^ t2);
                uint32_t len = e0 + (e0 == 4 ? e1 + (e1 == 4 ? e2 : 0u) : 0u);
                len = len < lim ? len : lim;
                if (len > best_len) { best_len = len; best_off = cum; }   // strict: nearest wins
                if (len == lim) walking = false;                          // :341-344
                int32_t at = (int32_t)myslot - (int32_t)cum;
                if (at < 0) at += (int32_t)kLinkSlots;
                dist = L.link[at];
            }
        }
        const uint32_t found = (best_len << 11) | best_off;

        // ---- PARSE + PACK the tokens that start in this batch
        const uint32_t stop = next < n ? next : n;
        while (c < stop) {
            if (s.flushed >= s.cap) break;
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)found, (int)uniform(c - B));
            const uint32_t len = r >> 11;
            if (len < 2) {                                    // literal (:365-375)
                const uint32_t lit = (uint32_t)__builtin_amdgcn_readlane((int)t0, (int)uniform(c - B)) & 0xFFu;
                sink_put(s, L.stage, lane, lit, 9);
                c += 1;
                continue;
            }
            const uint32_t off = r & kWindow;                 // match head (:376-409)
            const uint32_t first = len < kTokenMax ? len : kTokenMax;
            if (off <= kShortMax) sink_put(s, L.stage, lane, (3u << 7) | off, 9);
            else                  sink_put(s, L.stage, lane, (2u << 11) | off, 13);
            if (first <= 4) sink_put(s, L.stage, lane, first - 2, 2);
            else            sink_put(s, L.stage, lane, 0xCu + (first - 5), 4);
            c += first;
            if (first == kTokenMax) {
                // extension at the same offset (:417-431), up to 60 bytes = 4 nibbles per round
                bool more = true;
                while (more) {
                    chain_refill(L, src, n, src16, lane, loaded, c + 64);
                    // keep the chains current while c runs ahead: build every batch that is
                    // now wholly behind c (its bytes must still be in the ring)
                    while (next + 128 <= c) {
                        uint32_t u0, u1, u2;
                        chain_build(L, next, n, lane, u0, u1, u2);
                        next += 64;
                    }
                    const uint32_t rem = n - c;
                    const uint32_t span = rem < 60u ? rem : 60u;
                    const bool differs = lane < span &&
                                         ring_byte(L.ring, c + lane) != ring_byte(L.ring, c + lane - off);
                    const uint64_t stopmask = __ballot(differs) | (1ull << span);
                    const uint32_t m = uniform((uint32_t)__builtin_ctzll(stopmask));  // equal bytes <= span
                    c += m;
                    for (uint32_t k = m / kNibbleMax; k > 0; k--) sink_put(s, L.stage, lane, kNibbleMax, 4);
                    // 60 equal bytes = four full nibbles and the match may go on; anything
                    // shorter ends it with a last nibble of 0..14 (0 when it ended on a
                    // multiple of 15 or at the end of the input)
                    more = (m == 60u);
                    if (!more) sink_put(s, L.stage, lane, m % kNibbleMax, 4);
                }
            }
        }
    }
    sink_finish(s, L.stage, lane, &out_len[b]);
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
                                  uint32_t nblocks)
{
    __shared__ DecLds lds[kWavesPerWG];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv   = threadIdx.x >> 6;
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
                    if (off == 0) break;                           // end marker
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
                const uint32_t k = lane % off;                     // overlap replicates
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
    snprintf(buf, cap, "hip device %d: %s (%s), %d CUs, %.0f GiB, LDS/CU %zu KiB; kernels: wave-per-block LZS (gfx950)",
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
    // LZS_KERNEL=scan selects the brute-force variant (A/B and cross-checking); default "chain"
    static const bool use_scan = [] {
        const char *v = getenv("LZS_KERNEL");
        return v && v[0] == 's';
    }();
    if (use_scan)
        hipLaunchKernelGGL(lzs_compress_blocks_scan_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
    else
        hipLaunchKernelGGL(lzs_compress_blocks_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
    return (int)hipGetLastError();
}

int lzs_hip_launch_decompress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                              const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                              uint32_t in_len, uint32_t nblocks, void *stream)
{
    if (nblocks == 0) return 0;
    const uint32_t grid = (nblocks + kWavesPerWG - 1) / kWavesPerWG;
    hipLaunchKernelGGL(lzs_decompress_blocks_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                       (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                       (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
    return (int)hipGetLastError();
}

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

}  // extern "C"
