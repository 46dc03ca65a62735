// lzs_workload_gen.hip -- the seeded synthetic block streams of lzs_workload.c, generated ON THE
// DEVICE (64 GiB of text-class blocks are generated in HBM at the root GPU
// and scattered; a host generator plus 64 GiB over PCIe would dominate the run).
//
// Bench/test tooling like lzs_workload.c (built into liblzs_workload_hip.so; the codec library does
// not need it).  Block b of a class depends on (seed, class, b) alone and the bytes are those of
// lzs_workload_fill() bit for bit: tests/test_gpu_workload.py compares SHA-256 of device blocks
// with tests/golden/class_digests.json and with the host generator at several first_block values.
//
// Shape: the text and low-entropy streams are sequential per block (every draw of the counter
// PRNG steers what is drawn next), so ONE LANE generates one block, 64 blocks per wavefront, bytes
// collected eight at a time in a register and stored as one 8-byte word; the Zipf table sits in
// LDS (binary search), the per-lane "recent words" ring too.  The high-entropy class is a pure
// function of (block, word index): one thread per 8 bytes, coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

extern "C" int lzs_workload_vocab(uint8_t *len, char *txt, uint32_t *cdf);   // lzs_workload.c

namespace {

constexpr uint32_t kVocab = 5000, kWordMax = 14;
enum { kText = 0, kLowEnt = 1, kRandom = 2 };

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng { uint64_t key, ctr; };
__device__ __forceinline__ Rng rng_for(uint64_t seed, uint32_t cls, uint64_t block)
{
    Rng r; r.key = mix64(seed ^ mix64(((uint64_t)cls << 56) ^ block)); r.ctr = 0; return r;
}
__device__ __forceinline__ uint64_t rng_next(Rng &r) { return mix64(r.key ^ (r.ctr++ * 0xD1342543DE82EF95ull)); }
__device__ __forceinline__ uint32_t rng_below(Rng &r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }

// Output of one lane: bytes are collected little-endian in `acc` and leave as 8-byte words
// (`wide`: the block starts 8-aligned), else byte by byte.  Like wr_c() of lzs_workload.c the
// position runs on past `len`; nothing is stored there.
struct Wr { uint8_t *dst; uint32_t len, at; uint64_t acc; bool wide; };
__device__ __forceinline__ void wr_c(Wr &w, uint32_t c)
{
    if (w.at < w.len) {
        if (w.wide) {
            w.acc |= (uint64_t)(c & 0xFFu) << (8u * (w.at & 7u));
            if ((w.at & 7u) == 7u) { *reinterpret_cast<uint64_t *>(w.dst + (w.at & ~7u)) = w.acc; w.acc = 0; }
        } else {
            w.dst[w.at] = (uint8_t)c;
        }
    }
    w.at++;
}
__device__ __forceinline__ void wr_end(Wr &w)
{
    if (!w.wide) return;
    const uint32_t tail = w.len & 7u, base = w.len & ~7u;
    for (uint32_t k = 0; k < tail; k++) w.dst[base + k] = (uint8_t)(w.acc >> (8u * k));
}
__device__ __forceinline__ void wr_s(Wr &w, const char *s) { while (*s) wr_c(w, (uint32_t)(uint8_t)*s++); }
__device__ __forceinline__ void wr_num(Wr &w, uint32_t x)
{
    uint32_t p = 1;
    while (x / p >= 10u) p *= 10u;                       // highest power of ten <= x (1 for x < 10)
    for (; p; p /= 10u) wr_c(w, '0' + (x / p) % 10u);
}

// vocabulary as the device sees it: 16 bytes per word (14 letters, byte 15 = length), cdf apart
struct Vocab { const uint4 *word; const uint32_t *cdf; };

__device__ __forceinline__ void wr_word(Wr &w, const Vocab &v, uint32_t r, bool cap)
{
    const uint4 q = v.word[r];
    const uint32_t len = q.w >> 24;
    uint64_t lo = ((uint64_t)q.y << 32) | q.x, hi = ((uint64_t)q.w << 32) | q.z;
    for (uint32_t k = 0; k < len; k++) {
        uint32_t c = (uint32_t)(lo & 0xFFu);
        lo = (lo >> 8) | (hi << 56); hi >>= 8;
        if (cap && k == 0) c -= 32u;
        wr_c(w, c);
    }
}

// Topic locality (lzs_workload.c vocab_pick): the last 64 words of the lane, in LDS as recent[slot][lane]
struct Topic { uint16_t *recent; uint32_t n; };

__device__ __forceinline__ uint32_t zipf_pick(const uint32_t *cdf, Rng &r)
{
    const uint32_t u = (uint32_t)(rng_next(r) >> 32);
    uint32_t lo = 0, hi = kVocab - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t vocab_pick(const uint32_t *cdf, Rng &r, Topic &t)
{
    uint32_t w;
    if (t.n >= 8 && rng_below(r, 100) < 27) w = t.recent[64u * rng_below(r, t.n < 64 ? t.n : 64)];
    else                                     w = zipf_pick(cdf, r);
    t.recent[64u * (t.n & 63u)] = (uint16_t)w;
    t.n++;
    return w;
}

// Where lzs_workload.c passes two PRNG draws as arguments of one call (a word pick and a
// capitalisation draw), gcc evaluates the LAST argument first; the order is spelled out here
// (and in lzs_workload.c since round 2) so that every compiler produces the same stream.
__device__ void gen_text(const Vocab &v, const uint32_t *cdf, uint16_t *recent, uint8_t *dst, uint32_t len, bool wide,
                         uint64_t seed, uint64_t block)
{
    Rng r = rng_for(seed, kText, block);
    Wr w = { dst, len, 0, 0, wide };
    bool sentence_start = true;
    Topic topic = { recent, 0 };
    while (w.at < len) {
        const uint32_t kind = rng_below(r, 1000);
        if (kind < 8) {                                     // section heading
            const uint32_t depth = 2 + rng_below(r, 2);
            wr_c(w, '\n');
            for (uint32_t k = 0; k < depth; k++) wr_c(w, '=');
            wr_c(w, ' ');
            wr_word(w, v, vocab_pick(cdf, r, topic), true);
            if (rng_below(r, 2)) { wr_c(w, ' '); wr_word(w, v, vocab_pick(cdf, r, topic), false); }
            wr_c(w, ' ');
            for (uint32_t k = 0; k < depth; k++) wr_c(w, '=');
            wr_c(w, '\n');
            sentence_start = true;
        } else if (kind < 11) {                             // XML page scaffolding
            wr_s(w, "\n  </revision>\n</page>\n<page>\n  <title>");
            wr_word(w, v, vocab_pick(cdf, r, topic), true);
            wr_s(w, "</title>\n  <id>");
            wr_num(w, 1000 + rng_below(r, 9000000));
            wr_s(w, "</id>\n  <revision>\n    <timestamp>20");
            wr_num(w, 10 + rng_below(r, 16)); wr_c(w, '-');
            wr_num(w, 10 + rng_below(r, 3));  wr_c(w, '-');
            wr_num(w, 10 + rng_below(r, 19));
            wr_s(w, "T00:00:00Z</timestamp>\n    <text xml:space=\"preserve\">");
            sentence_start = true;
        } else if (kind < 60) {                             // wiki link
            wr_s(w, "[[");
            { const bool cap = rng_below(r, 2) != 0; const uint32_t word = vocab_pick(cdf, r, topic); wr_word(w, v, word, cap); }
            if (rng_below(r, 3) == 0) { wr_c(w, ' '); wr_word(w, v, vocab_pick(cdf, r, topic), false); }
            if (rng_below(r, 4) == 0) { wr_c(w, '|'); wr_word(w, v, vocab_pick(cdf, r, topic), false); }
            wr_s(w, "]] ");
            sentence_start = false;
        } else if (kind < 75) {                             // emphasis
            const uint32_t q = 2 + rng_below(r, 2);
            for (uint32_t k = 0; k < q; k++) wr_c(w, '\'');
            wr_word(w, v, vocab_pick(cdf, r, topic), false);
            for (uint32_t k = 0; k < q; k++) wr_c(w, '\'');
            wr_c(w, ' ');
            sentence_start = false;
        } else if (kind < 95) {                             // number
            const uint32_t range = rng_below(r, 2) ? 2100u : 100000u;
            wr_num(w, rng_below(r, range));
            wr_c(w, ' ');
            sentence_start = false;
        } else {                                            // plain word + separator
            { const bool cap = sentence_start || rng_below(r, 40) == 0; const uint32_t word = vocab_pick(cdf, r, topic); wr_word(w, v, word, cap); }
            sentence_start = false;
            const uint32_t sep = rng_below(r, 100);
            if (sep < 8)       { wr_s(w, ". "); sentence_start = true; if (rng_below(r, 5) == 0) wr_c(w, '\n'); }
            else if (sep < 15) wr_s(w, ", ");
            else if (sep < 16) wr_s(w, "; ");
            else if (sep < 17) wr_s(w, " (");
            else if (sep < 18) wr_s(w, ") ");
            else               wr_c(w, ' ');
        }
    }
    wr_end(w);
}

// Alternating segments: a run of 0x00 of length U[1,4096], then a 16-byte pattern U[1,256] times.
__device__ void gen_lowent(uint8_t *dst, uint32_t len, bool wide, uint64_t seed, uint64_t block)
{
    Rng r = rng_for(seed, kLowEnt, block);
    Wr w = { dst, len, 0, 0, wide };
    while (w.at < len) {
        uint32_t run = 1 + rng_below(r, 4096);
        for (; run && w.at < len; run--) wr_c(w, 0);
        const uint64_t a = rng_next(r), b = rng_next(r);
        const uint32_t reps = 1 + rng_below(r, 256);
        for (uint32_t t = 0; t < reps * 16u && w.at < len; t++)
            wr_c(w, (uint32_t)(((t & 8u) ? b : a) >> (8u * (t & 7u))));
    }
    wr_end(w);
}

__global__ __launch_bounds__(64)
void lzs_gen_blocks_kernel(uint8_t *__restrict__ dst, uint32_t cls, uint64_t seed, uint64_t first_block,
                           uint64_t nblocks, uint32_t block_len, const uint4 *__restrict__ words,
                           const uint32_t *__restrict__ cdf_g)
{
    __shared__ uint32_t cdf[kVocab];
    __shared__ uint16_t recent[64 * 64];
    if (cls == kText) {
        for (uint32_t i = threadIdx.x; i < kVocab; i += 64) cdf[i] = cdf_g[i];
        __syncthreads();
    }
    const uint64_t b = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (b >= nblocks) return;
    uint8_t *d = dst + b * block_len;
    const bool wide = (((uintptr_t)d) & 7u) == 0;
    if (cls == kText) {
        Vocab v = { words, cdf_g };
        gen_text(v, cdf, recent + threadIdx.x, d, block_len, wide, seed, first_block + b);
    } else {
        gen_lowent(d, block_len, wide, seed, first_block + b);
    }
}

// high entropy: 8 bytes per thread, word j of block b = draw j of that block's stream
__global__ __launch_bounds__(256)
void lzs_gen_random_kernel(uint8_t *__restrict__ dst, uint64_t seed, uint64_t first_block, uint64_t nblocks, uint32_t block_len)
{
    const uint32_t per = (block_len + 7u) >> 3;
    const uint64_t total = nblocks * per;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t b = i / per;
        const uint32_t j = (uint32_t)(i - b * per);
        const uint64_t key = mix64(seed ^ mix64(((uint64_t)kRandom << 56) ^ (first_block + b)));
        const uint64_t val = mix64(key ^ ((uint64_t)j * 0xD1342543DE82EF95ull));
        uint8_t *d = dst + b * block_len + 8ull * j;
        const uint32_t left = block_len - 8u * j;
        if (left >= 8u && (((uintptr_t)d) & 7u) == 0) *reinterpret_cast<uint64_t *>(d) = val;
        else for (uint32_t k = 0; k < 8u && k < left; k++) d[k] = (uint8_t)(val >> (8u * k));
    }
}

struct DevVocab { int dev; uint4 *words; uint32_t *cdf; };
DevVocab g_vocab[16];          // one copy per device that asked

int vocab_for_device(const uint4 **words, const uint32_t **cdf)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
    DevVocab &g = g_vocab[dev];
    if (!g.words) {
        uint8_t *len = (uint8_t *)malloc(kVocab);
        char *txt = (char *)malloc(kVocab * kWordMax);
        uint32_t *c = (uint32_t *)malloc(kVocab * 4);
        uint8_t *packed = (uint8_t *)calloc(kVocab, 16);
        if (!len || !txt || !c || !packed || lzs_workload_vocab(len, txt, c) != 0) { free(len); free(txt); free(c); free(packed); return (int)hipErrorOutOfMemory; }
        for (uint32_t r = 0; r < kVocab; r++) { memcpy(packed + 16 * r, txt + kWordMax * r, len[r]); packed[16 * r + 15] = len[r]; }
        e = hipMalloc((void **)&g.words, kVocab * 16);
        if (e == hipSuccess) e = hipMalloc((void **)&g.cdf, kVocab * 4);
        if (e == hipSuccess) e = hipMemcpy(g.words, packed, kVocab * 16, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(g.cdf, c, kVocab * 4, hipMemcpyHostToDevice);
        free(len); free(txt); free(c); free(packed);
        if (e != hipSuccess) { g.words = nullptr; return (int)e; }
        g.dev = dev;
    }
    *words = g.words; *cdf = g.cdf;
    return 0;
}

}  // namespace

// Fill d_dst[nblocks * block_len] (device memory of the current device) with blocks
// first_block .. first_block + nblocks - 1 of class cls, asynchronously on `stream`.
// Returns 0, -1 for a bad class, or a hipError_t value.
extern "C" int lzs_workload_fill_device(void *d_dst, unsigned cls, uint64_t seed, uint64_t first_block,
                                        size_t nblocks, size_t block_len, void *stream)
{
    if (cls > kRandom || block_len > 0xFFFFFFF0ull) return -1;
    if (nblocks == 0 || block_len == 0) return 0;
    if (cls == kRandom) {
        const uint64_t words = (uint64_t)nblocks * ((block_len + 7) / 8);
        const uint32_t grid = (uint32_t)(words / 256 + 1 < 65536 * 4 ? words / 256 + 1 : 65536 * 4);
        hipLaunchKernelGGL(lzs_gen_random_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (uint8_t *)d_dst, seed, first_block, (uint64_t)nblocks, (uint32_t)block_len);
        return (int)hipGetLastError();
    }
    const uint4 *words = nullptr; const uint32_t *cdf = nullptr;
    if (cls == kText) { const int e = vocab_for_device(&words, &cdf); if (e) return e; }
    hipLaunchKernelGGL(lzs_gen_blocks_kernel, dim3((uint32_t)((nblocks + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       (uint8_t *)d_dst, cls, seed, first_block, (uint64_t)nblocks, (uint32_t)block_len, words, cdf);
    return (int)hipGetLastError();
}
