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
// The kernels are in kernels/*.inc, included below into one translation unit (one hipcc run,
// one anonymous namespace):
//   common.inc             constants, LDS ring helpers, the wave-level bit sink
//   compress_wg.inc        the default compressor: one 256-thread workgroup per block, pools of
//                          512 positions pipelined through HASH/CHAIN, SEARCH, EXTEND, PARSE, PACK
//                          (DESIGN.md 3.1); the same loop from the middle of a stream for segments
//                          of one long stream, their bit-level stitch, and the resume of a long
//                          match for the incremental interface (3.5, 3.7)
//   (tools/variants/compress_variants.inc: "scan" and "chain", the earlier compressors -- LZS_KERNEL=, A/B builds only)
//   decompress_blocks.inc  eight streams per wavefront (3.3); the earlier one-wavefront-per-stream
//                          decoders for A/B builds
//   decompress_stream.inc  one stream or a small batch on many wavefronts: scan (a lane per
//                          segment), decode with per-byte origins (eight segments per wavefront),
//                          origins resolved on the segments' tails by chunks, then everywhere (3.6)
//   compact_resume.inc     slot compaction (3.4); the resumable decoder of the incremental
//                          interface (3.7)
// No MFMA anywhere: this is byte search and bit packing, not a contraction.
//
// The library compiles this file TWICE (csrc/Makefile): -DLZS_TU_COMPRESS at -Os -fno-unroll-loops -- the compress kernel's
// 15 KB of code are run by five workgroups per CU at five different places, and smaller is faster there -- and
// -DLZS_TU_DECOMPRESS at -O3 -fno-unroll-loops, which the decoders' one loop prefers by 2-4 % (profiles/r05/
// abdec_s27_optimisation_levels.txt).  Each half leaves the other's kernels and launchers out; with neither macro (the probes
// of tools/probes/, which include this file) everything is in.
//
// gfx950 only.  No CUDA compatibility layer, no alternate code paths.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>

#include "lzs_hip_shim.h"

namespace {

#include "kernels/common.inc"
#ifndef LZS_TU_DECOMPRESS
#ifdef LZS_WITH_VARIANTS   // the earlier compressors ("chain", "scan") and the v1 decoder: A/B builds only
#include "compress_variants.inc"                 // tools/variants/ (the Makefile adds the path for liblzs_variants.so only)
#endif
// The compress kernel, once per variant (kernels/compress_wg.inc says what the LZS_WGV_* mean): the default, the
// order-independent CHAIN for a device that fails the LDS ordering check, and the two a block's class may ask for.
namespace wgv_text {
#define LZS_WGV_SEGMENTS 1
#define LZS_WGV_PRIO 1
#ifndef LZS_EXP_TEXT_EXIT16
#define LZS_WGV_EXIT8 1
#endif
// Round 6: SIX workgroups per CU.  A sixth workgroup is +8.9 % at equal work (profiles/r06/ab_s19) and until now cost more in buckets
// than it gave (1792 / 1024 -> 1024 / 512: -9.5 %; round 5's ab_s41: 72.6 against 74.7) -- with multipliers that spread a text's
// grams (kernels/compress_wg.inc, tools/sim/hash_sim.c) the smaller tables have FEWER collisions than the large ones had:
// 26.7 KB of LDS, SEARCH's constants as literals and the allocation held to 80 registers.  profiles/r06/ab_s31 ... ab_s34:
// 74.6 -> 76.8 (the multipliers alone, 1792 / 1024, five workgroups) -> 79.1 GB/s (1024 / 512 buckets, six).
// 1152 three-byte buckets since PARSE's exit functions are one byte a position (26 720 B; 26 848 is the most six workgroups leave
// each other): 3.52 candidates a walk visits instead of 3.74, +0.6 ... 1.0 % (profiles/r06/ab_s48).
#ifndef LZS_EXP_TEXT_HEAD3
#define LZS_EXP_TEXT_HEAD3 1152
#endif
#ifndef LZS_EXP_TEXT_HEAD2
#define LZS_EXP_TEXT_HEAD2 512
#endif
#ifndef LZS_EXP_TEXT_WAVES
#define LZS_EXP_TEXT_WAVES 6
#endif
#define LZS_WGV_HEAD3 LZS_EXP_TEXT_HEAD3
#define LZS_WGV_HEAD2 LZS_EXP_TEXT_HEAD2
#if LZS_EXP_TEXT_WAVES > 0
#define LZS_WGV_WAVES LZS_EXP_TEXT_WAVES
#endif
#ifndef LZS_EXP_TEXT_NOT_LEAN
#define LZS_WGV_LEAN 1
#endif
#ifndef LZS_HASH3_MUL         // (the best of 8000 for 1024 / 512 buckets and of 6000 for 1152 / 512; -DLZS_HASH3_MUL= / -DLZS_HASH2_MUL= override: tools/probes/ab.sh)
#define LZS_WGV_HASH3_MUL 0x897397u
#endif
#ifndef LZS_HASH2_MUL
#define LZS_WGV_HASH2_MUL 0x64EBAD33u
#endif
#ifdef LZS_EXP_TEXT_POOL      // (tools/probes/ab.sh: round 6's pool of 256, profiles/r06/ab_s3)
#define LZS_WGV_POOL LZS_EXP_TEXT_POOL
#endif
#ifdef LZS_EXP_TEXT_WG_WAVES
#define LZS_WGV_WG_WAVES LZS_EXP_TEXT_WG_WAVES
#endif
#include "kernels/compress_wg.inc"
}
namespace wgv_safe {
#define LZS_WGV_CHAIN_SAFE 1
#define LZS_WGV_SEGMENTS 1
#define LZS_WGV_PRIO 1
#include "kernels/compress_wg.inc"
}
#ifdef LZS_WITH_VARIANTS      // liblzs_variants.so only (LZS_KERNEL=wg8 | p256): round 6's two other shapes of the default kernel, kept
namespace wgv_text8 {         // under test as parameters of kernels/compress_wg.inc -- eight waves per workgroup (pools of 1024) ...
#define LZS_WGV_PRIO 1
#define LZS_WGV_WG_WAVES 8
#define LZS_WGV_WAVES 6
#define LZS_WGV_EXIT8 1
#include "kernels/compress_wg.inc"
}
namespace wgv_pool256 {       // ... and pools of 256 positions (one chunk per wave)
#define LZS_WGV_PRIO 1
#define LZS_WGV_POOL 256
#include "kernels/compress_wg.inc"
}
#endif
#ifndef LZS_ONE_VARIANT      // (tools/probes/ab.sh -DLZS_ONE_VARIANT: the default alone, for the probes that launch it directly)
// Round 6: NEITHER of the two keeps a 3-byte chain (LZS_WGV_NO3: every position walks the 2-byte chain with the full rule, which is
// exact for any block -- that chain is complete for every match of 2 and more).  Where grams do not repeat the 3-byte chain holds
// collisions only, where few grams repeat endlessly it holds what the 2-byte chain holds; without head3[] / link3[] HASH and CHAIN do
// half the work, a step has no restart on another chain, and 18.5 KB of LDS with the registers held to 64 (LZS_WGV_WAVES: no vector
// spills) are EIGHT workgroups per CU -- all a SIMD takes.  profiles/r06/ab_s10 ... ab_s15, GB/s through the library's launch:
//   high entropy 107.4 (both chains, six workgroups) -> 115 (no 3-byte chain, 1024 buckets, seven) -> 119-122 (2048 buckets; hops 0 / 1 / 2 /
//   3 / 4 / 5: 120.2 / 122.4 / 121.5 / 119.1 / 116.0 / 112.8; 4096 buckets -- five workgroups -- 110; two sub-steps 112-116) -> 125.4 (eight
//   workgroups with 1024 buckets, two hops; one 124.3, none 116.7)
//   low entropy 416 -> 433 / 436 / 437 (512 / 1024 / 2048 buckets, seven workgroups; one hop 419) -> 493 (eight workgroups, 1024 buckets; 512:
//   484; run mode's pool 64 / 256: 470 / 460, its rounds 1 / 3: 491 / 491)
// Forced on text they are 58-68 GB/s: the default keeps both chains.
namespace wgv_few {           // blocks of few distinct grams: long matches, the kernel waits -- eight workgroups per CU; no priorities (+0.2 %: noise)
#define LZS_WGV_NO3 1
#define LZS_WGV_HEAD2 1024
#define LZS_WGV_WAVES 8
#define LZS_WGV_PACK_BY_CHUNK 1   // (these blocks have few tokens to format)
#define LZS_WGV_LEAN 1
#define LZS_WGV_HOPS 0        // (few candidates, long matches: the quick-reject test itself is the cost)
#include "kernels/compress_wg.inc"
}
namespace wgv_lit {           // blocks that are nearly all literals: one full step per pass of the SEARCH loop with two hops in front, PACK chunk by chunk
#define LZS_WGV_NO3 1
#define LZS_WGV_HEAD2 1024
#define LZS_WGV_WAVES 8
#define LZS_WGV_SUBSTEPS 1
#define LZS_WGV_HOPS 2
#define LZS_WGV_PACK_BY_CHUNK 1
#define LZS_WGV_PRIO 1
#define LZS_WGV_LEAN 1
#include "kernels/compress_wg.inc"
}
#endif
using wgv_text::lzs_compress_blocks_wg_kernel;        // (what tools/probes/ launch by name)
using wgv_text::kWgThreads;
#include "kernels/compress_aux.inc"
#endif  // !LZS_TU_DECOMPRESS
#ifndef LZS_TU_COMPRESS
#include "kernels/decompress_blocks.inc"
#include "kernels/decompress_stream.inc"
#endif
#ifndef LZS_TU_DECOMPRESS
#include "kernels/compact_resume.inc"
#endif

}  // namespace

// =====================================================================================
// extern "C" shim (see lzs_hip_shim.h)
// =====================================================================================
extern "C" {
#ifndef LZS_TU_DECOMPRESS

int lzs_hip_device_count(int *count) { return (int)hipGetDeviceCount(count); }

const char *lzs_hip_strerror(int e) { return hipGetErrorString((hipError_t)e); }

int lzs_hip_chain_mode(void *stream, int *mode);

int lzs_hip_describe(char *buf, size_t cap)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    int mode = -1;
    (void)lzs_hip_chain_mode(nullptr, &mode);
    snprintf(buf, cap, "hip device %d: %s (%s), %d CUs, %.0f GiB, LDS/CU %zu KiB; kernels: workgroup-per-block LZS compress "
                       "(chain build: %s), eight-streams-per-wavefront decompress (gfx950)",
             dev, p.name, p.gcnArchName, p.multiProcessorCount,
             (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0),
             (size_t)p.maxSharedMemoryPerMultiProcessor / 1024,
             mode == 0 ? "ordered LDS exchange, verified on this device" : (mode == 1 ? "order-independent fallback" : "not checked"));
    return 0;
}

int lzs_hip_total_memory(size_t *bytes)
{
    size_t free_b = 0, total = 0;
    const hipError_t e = hipMemGetInfo(&free_b, &total);
    if (e == hipSuccess) *bytes = total;
    return (int)e;
}
int lzs_hip_malloc(void **p, size_t bytes) { return (int)hipMalloc(p, bytes ? bytes : 1); }
int lzs_hip_free(void *p) { return (int)hipFree(p); }
int lzs_hip_host_malloc(void **p, size_t bytes) { return (int)hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault); }
int lzs_hip_host_malloc_staging(void **p, size_t bytes) { return (int)hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocNonCoherent); }
int lzs_hip_host_free(void *p) { return (int)hipHostFree(p); }
void lzs_hip_clear_error(void) { (void)hipGetLastError(); }

namespace {
__global__ __launch_bounds__(256) void lzs_words_to_host_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();                         // dst is host memory that the host reads at an event
}
}  // namespace
int lzs_hip_words_to_host(uint32_t *h_dst, const uint32_t *d_src, size_t nwords, void *stream)
{
    if (nwords == 0) return 0;
    const size_t grid = (nwords + 255) / 256;
    hipLaunchKernelGGL(lzs_words_to_host_kernel, dim3((unsigned)(grid < 64 ? grid : 64)), dim3(256), 0, (hipStream_t)stream, h_dst, d_src, nwords);
    return (int)hipGetLastError();
}
int lzs_hip_stream_create(void **s) { return (int)hipStreamCreateWithFlags((hipStream_t *)s, hipStreamNonBlocking); }
int lzs_hip_stream_destroy(void *s) { return (int)hipStreamDestroy((hipStream_t)s); }
int lzs_hip_stream_sync(void *s) { return (int)hipStreamSynchronize((hipStream_t)s); }
int lzs_hip_event_create(void **e) { return (int)hipEventCreateWithFlags((hipEvent_t *)e, hipEventDisableTiming); }
int lzs_hip_event_destroy(void *e) { return (int)hipEventDestroy((hipEvent_t)e); }
int lzs_hip_event_record(void *e, void *s) { return (int)hipEventRecord((hipEvent_t)e, (hipStream_t)s); }
int lzs_hip_event_sync(void *e) { return (int)hipEventSynchronize((hipEvent_t)e); }
int lzs_hip_event_done(void *e)
{
    const hipError_t r = hipEventQuery((hipEvent_t)e);
    return r == hipSuccess ? 1 : (r == hipErrorNotReady ? 0 : -(int)r);
}
int lzs_hip_stream_wait_event(void *s, void *e) { return (int)hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0); }
int lzs_hip_h2d(void *d, const void *s, size_t n, void *st)
{
    return n ? (int)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st) : 0;
}
int lzs_hip_d2h(void *d, const void *s, size_t n, void *st)
{
    return n ? (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st) : 0;
}
int lzs_hip_d2d(void *d, const void *s, size_t n, void *st)
{
    return n ? (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st) : 0;
}
int lzs_hip_memset(void *d, int v, size_t n, void *st)
{
    return n ? (int)hipMemsetAsync(d, v, n, (hipStream_t)st) : 0;
}

// 0: CHAIN by ordered LDS exchange (verified on this device), 1: the order-independent form.
// Asked once per device: lzs_lds_order_check_kernel over 16384 conflict patterns (~0.1 ms); a device
// that fails it -- or LZS_CHAIN_FALLBACK=1 -- gets the slower form, with a note on stderr.
static int g_chain_mode[64];            // 0 = not asked yet, else mode + 1
// What the classifiers of recent launches found (compress_aux.inc): four words of pinned host memory per device, h_seen[c] =
// the id of the last launch a sampled workgroup of which met a block of class c.  Made with the LDS ordering check, once per
// device; without them (the allocation failed) every launch runs all three variants.
static uint32_t *g_hint_seen[64];
static unsigned g_hint_launch;
int lzs_hip_chain_mode(void *stream, int *mode)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) { *mode = 1; return 0; }
    int m = __atomic_load_n(&g_chain_mode[dev], __ATOMIC_ACQUIRE);
    if (m == 0) {
        const char *force = getenv("LZS_CHAIN_FALLBACK");
        if (force && force[0] && force[0] != '0') {
            m = 2;
        } else {
            // On a stream of its own: the caller's stream is neither waited for nor
            // made to wait, whichever call is the first to ask -- normally require_device(), i.e.
            // the first entry into the library on this device, not the first "asynchronous" launch.
            (void)stream;
            hipStream_t own = nullptr;
            uint32_t *d_bad = nullptr, bad = 0;
            e = hipStreamCreateWithFlags(&own, hipStreamNonBlocking);
            if (e != hipSuccess) return (int)e;
            e = hipMalloc((void **)&d_bad, sizeof(uint32_t));
            if (e != hipSuccess) { (void)hipStreamDestroy(own); return (int)e; }
            e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), own);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(lzs_lds_order_check_kernel, dim3(512), dim3(256), 0, own, d_bad, 16384u);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, own);
            if (e == hipSuccess) e = hipStreamSynchronize(own);
            (void)hipFree(d_bad);
            (void)hipStreamDestroy(own);
            if (e != hipSuccess) return (int)e;
            m = bad ? 2 : 1;
            if (!bad && !g_hint_seen[dev]) {
                uint32_t *h_seen = nullptr;
                if (hipHostMalloc((void **)&h_seen, 4 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) {
                    h_seen[0] = h_seen[1] = h_seen[2] = h_seen[3] = 0;
                    uint32_t *none = nullptr;         // (two threads asking at once: one set of words stays)
                    if (!__atomic_compare_exchange_n(&g_hint_seen[dev], &none, h_seen, false, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) (void)hipHostFree(h_seen);
                } else {
                    (void)hipGetLastError();
                }
            }
            if (bad)
                fprintf(stderr, "liblzs: device %d does not apply same-address LDS exchanges of one instruction in lane order "
                                "(%u lanes off in 16384 patterns): using the order-independent chain build (slower, same output)\n", dev, bad);
        }
        __atomic_store_n(&g_chain_mode[dev], m, __ATOMIC_RELEASE);
    }
    *mode = m - 1;
    return 0;
}

// ---- the ordering property once more, UNDER LOAD: lzs_hip_chain_mode() asks an idle device.  Beside the
// first compress launch of a process on a device the same check runs again on a stream of its own -- 65536 patterns
// this time, so that it overlaps the launch -- and whichever later launch finds it finished reads the verdict: a device
// that fails it gets the order-independent CHAIN from then on, with a loud note (the launches before may be
// off in their candidates' order: same format, possibly not the reference's bytes).
struct LoadCheck { int state; hipStream_t own; hipEvent_t done; uint32_t *d_bad; uint32_t *h_bad; };   // state: 0 not started, 1 in flight, 2 read
static LoadCheck g_load_check[64];
static pthread_mutex_t g_load_lock = PTHREAD_MUTEX_INITIALIZER;

static void load_check_step(int dev, bool start)
{
    if (dev < 0 || dev >= 64 || __atomic_load_n(&g_load_check[dev].state, __ATOMIC_ACQUIRE) == 2) return;
    pthread_mutex_lock(&g_load_lock);
    LoadCheck &c = g_load_check[dev];
    if (c.state == 0 && start) {
        c.state = 2;                                             // whatever fails below: do not try again
        if (hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&c.done, hipEventDisableTiming) == hipSuccess &&
            hipMalloc((void **)&c.d_bad, sizeof(uint32_t)) == hipSuccess &&
            hipHostMalloc((void **)&c.h_bad, sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) {
            *c.h_bad = 0;
            (void)hipMemsetAsync(c.d_bad, 0, sizeof(uint32_t), c.own);
            hipLaunchKernelGGL(lzs_lds_order_check_kernel, dim3(512), dim3(256), 0, c.own, c.d_bad, 65536u);
            if (hipGetLastError() == hipSuccess &&
                hipMemcpyAsync(c.h_bad, c.d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, c.own) == hipSuccess &&
                hipEventRecord(c.done, c.own) == hipSuccess)
                c.state = 1;
        }
    } else if (c.state == 1 && hipEventQuery(c.done) == hipSuccess) {
        if (*c.h_bad) {
            fprintf(stderr, "liblzs: device %d applied same-address LDS exchanges out of lane order UNDER LOAD (%u lanes off in 65536 "
                            "patterns beside a compress launch) although it passed the check when idle: switching to the "
                            "order-independent chain build; the launches so far may differ from the reference's bytes\n", dev, *c.h_bad);
            __atomic_store_n(&g_chain_mode[dev], 2, __ATOMIC_RELEASE);
        }
        // (the four bytes, the event and the stream stay for the life of the process: releasing device memory waits for
        // the device, and this runs inside a launch that promised to be asynchronous)
        __atomic_store_n(&c.state, 2, __ATOMIC_RELEASE);
    }
    pthread_mutex_unlock(&g_load_lock);
}

// 0: never asked / finished clean; for tests: has the check beside a launch run to its end, and what did it say
int lzs_hip_load_check_state(int dev) { return dev >= 0 && dev < 64 ? __atomic_load_n(&g_load_check[dev].state, __ATOMIC_ACQUIRE) : -1; }

// LZS_VERIFY=N: every N-th compress launch of the process is run again with CHAIN in its order-independent form
// into scratch slots and compared on the device, length and bytes of every block; a difference is an error of the
// launch (and a line on stderr).  That launch waits for its result: an audit mode, not the asynchronous contract.
static int verify_every(void)
{
    static const int n = [] { const char *v = getenv("LZS_VERIFY"); const long k = v ? strtol(v, nullptr, 10) : 0; return k > 0 ? (int)k : 0; }();
    return n;
}
static unsigned g_launch_count;

static int verify_launch(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                         const void *d_in, size_t in_stride, const uint32_t *d_in_len, uint32_t in_len, uint32_t nblocks, hipStream_t stream)
{
    const size_t stride = ((size_t)out_cap + 15u) & ~(size_t)15u;
    uint8_t *d_scratch = nullptr; uint32_t *d_len2 = nullptr, bad = 0;
    hipError_t e = hipMalloc((void **)&d_scratch, stride * nblocks + 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_len2, sizeof(uint32_t) * ((size_t)nblocks + 1));
    if (e == hipSuccess) e = hipMemsetAsync(d_len2 + nblocks, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(wgv_safe::lzs_compress_blocks_wg_kernel, dim3(nblocks), dim3(wgv_safe::kWgThreads), 0, stream, d_scratch, stride, out_cap, d_len2,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, 0u);
        hipLaunchKernelGGL(lzs_verify_slots_kernel, dim3(nblocks), dim3(256), 0, stream, (const uint8_t *)d_out, (const uint32_t *)d_out_len,
                           (const uint8_t *)d_scratch, (const uint32_t *)d_len2, out_stride, stride, nblocks, d_len2 + nblocks);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_len2 + nblocks, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_scratch); (void)hipFree(d_len2);
    if (e != hipSuccess) return (int)e;
    if (bad) {
        fprintf(stderr, "liblzs: LZS_VERIFY: %u of %u blocks of this launch differ between the ordered-exchange chain build and the "
                        "order-independent one\n", bad, nblocks);
        return (int)hipErrorAssert;
    }
    return 0;
}

int lzs_hip_launch_compress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                            const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                            uint32_t in_len, uint32_t nblocks, void *stream)
{
    if (nblocks == 0) return 0;
    int chain_mode = 0;
    { const int e = lzs_hip_chain_mode(stream, &chain_mode); if (e) return e; }
#ifdef LZS_WITH_VARIANTS
    const uint32_t grid = (nblocks + kWavesPerWG - 1) / kWavesPerWG;
    // A/B builds only: LZS_KERNEL=chain (one wave per block) | scan (brute force) select the earlier kernels
    static const int variant = [] {
        const char *v = getenv("LZS_KERNEL");
        return !v ? 0 : (v[0] == 's' ? 2 : (v[0] == 'c' ? 1 : (v[0] == 'w' ? 3 : (v[0] == 'p' ? 4 : 0))));
    }();
    if (variant == 3 || variant == 4) {        // the default kernel in its two other shapes (round 6)
        if (variant == 3)
            hipLaunchKernelGGL(wgv_text8::lzs_compress_blocks_wg_kernel, dim3(nblocks), dim3(wgv_text8::kWgThreads), 0, (hipStream_t)stream, (uint8_t *)d_out,
                               out_stride, out_cap, d_out_len, (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, 0u);
        else
            hipLaunchKernelGGL(wgv_pool256::lzs_compress_blocks_wg_kernel, dim3(nblocks), dim3(wgv_pool256::kWgThreads), 0, (hipStream_t)stream, (uint8_t *)d_out,
                               out_stride, out_cap, d_out_len, (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, 0u);
        return (int)hipGetLastError();
    }
    if (variant == 2) {
        hipLaunchKernelGGL(lzs_compress_blocks_scan_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
        return (int)hipGetLastError();
    }
    if (variant == 1) {
        hipLaunchKernelGGL(lzs_compress_blocks_kernel, dim3(nblocks), dim3(64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks);
        return (int)hipGetLastError();
    }
#endif
    // Which variant compresses a block: the order-independent CHAIN for all of them on a device that needs it; otherwise the
    // block's class decides, on the device (lzs_classify_blocks_kernel leaves a code in out_len[b]; every variant is launched
    // over the whole grid and a workgroup whose block is another's returns at once: ~10 us per empty grid of 16 384).
    // LZS_VARIANT=text|few|lit (development, read once) gives every block that one; small launches take the default alone.
    static const int forced = [] {
        const char *v = getenv("LZS_VARIANT");
        return !v ? 0 : (v[0] == 't' ? 1 : (v[0] == 'f' ? 2 : (v[0] == 'l' ? 3 : 0)));
    }();
#define LZS_LAUNCH_VARIANT(ns, serve) \
    hipLaunchKernelGGL(ns::lzs_compress_blocks_wg_kernel, dim3(nblocks), dim3(ns::kWgThreads), 0, (hipStream_t)stream, (uint8_t *)d_out, out_stride, \
                       out_cap, d_out_len, (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, serve)
    if (chain_mode != 0) LZS_LAUNCH_VARIANT(wgv_safe, 0u);
#ifndef LZS_ONE_VARIANT
    else if (forced == 2) LZS_LAUNCH_VARIANT(wgv_few, 0u);
    else if (forced == 3) LZS_LAUNCH_VARIANT(wgv_lit, 0u);
    else if (forced == 0 && nblocks >= kClassifyMinBlocks) {
        // the variants recent launches had blocks for (all three while nothing is known); a block of another class goes to the default
        // variant if that runs, else to the first that does
        int dev = 0;
        uint32_t allow = 0xEu, catch_all = 1u, *h_seen = nullptr;
        const uint32_t id = (uint32_t)__atomic_add_fetch(&g_hint_launch, 1u, __ATOMIC_RELAXED) | 0x80000000u;      // (never 0)
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && (h_seen = __atomic_load_n(&g_hint_seen[dev], __ATOMIC_ACQUIRE)) != nullptr) {
            uint32_t s[4], newest = 0;
            for (int c = 1; c <= 3; c++) { s[c] = __atomic_load_n(&h_seen[c], __ATOMIC_RELAXED); if (s[c] && (newest == 0 || (int32_t)(s[c] - newest) > 0)) newest = s[c]; }
            // (while the stream is being captured into a graph the guess would be frozen with it: every variant is launched, so that
            // a replay on another class of data keeps its variant -- ADVICE r05)
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (stream && hipStreamIsCapturing((hipStream_t)stream, &capturing) != hipSuccess) { capturing = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
            if (newest && capturing == hipStreamCaptureStatusNone) {
                allow = 0;
                for (int c = 1; c <= 3; c++) if (s[c] && newest - s[c] <= 2u) allow |= 1u << c;       // (seen by one of the last three classifiers)
                catch_all = (allow & 2u) ? 1u : ((allow & 4u) ? 2u : 3u);
            }
        }
        hipLaunchKernelGGL(lzs_classify_blocks_kernel, dim3((nblocks + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, allow, catch_all, h_seen, id);
        if (allow & 2u) LZS_LAUNCH_VARIANT(wgv_text, kClassCode | 1u);
        if (allow & 4u) LZS_LAUNCH_VARIANT(wgv_few, kClassCode | 2u);
        if (allow & 8u) LZS_LAUNCH_VARIANT(wgv_lit, kClassCode | 3u);
    }
#endif
    else LZS_LAUNCH_VARIANT(wgv_text, 0u);
#undef LZS_LAUNCH_VARIANT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (chain_mode == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) load_check_step(dev, nblocks >= 256u);     // (beside a launch that fills the device)
        const int every = verify_every();
        if (every && (__atomic_add_fetch(&g_launch_count, 1u, __ATOMIC_RELAXED) % (unsigned)every) == 0u)
            return verify_launch(d_out, out_stride, out_cap, d_out_len, d_in, in_stride, d_in_len, in_len, nblocks, (hipStream_t)stream);
    }
    return 0;
}

// The classifier alone (tests, bench.py's class mix): d_codes[b] = 1 the default variant, 2 few distinct grams, 3 nearly all literals.
int lzs_hip_classify_blocks(uint32_t *d_codes, const void *d_in, size_t in_stride, const uint32_t *d_in_len, uint32_t in_len,
                            uint32_t nblocks, void *stream)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(lzs_classify_blocks_kernel, dim3((nblocks + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, d_codes,
                       (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, 0xEu, 1u, (uint32_t *)nullptr, 0u);
    return (int)hipGetLastError();
}

#endif  // !LZS_TU_DECOMPRESS
#ifndef LZS_TU_COMPRESS
static int launch_decompress(void *d_out, size_t out_stride, uint32_t out_cap, uint32_t *d_out_len,
                             const void *d_in, size_t in_stride, const uint32_t *d_in_len,
                             uint32_t in_len, uint32_t nblocks, void *stream, uint32_t concat)
{
    if (nblocks == 0) return 0;
#ifdef LZS_WITH_VARIANTS
    // A/B builds only: LZS_DECODER=v1|v2 select the wave-per-stream decoders of round 1
    static const int older = [] { const char *v = getenv("LZS_DECODER"); return (v && v[0] == 'v') ? (v[1] == '1' ? 1 : (v[1] == '2' ? 2 : 0)) : 0; }();
    if (older) {
        const uint32_t grid = (nblocks + kWavesPerWG - 1) / kWavesPerWG;
        if (older == 1)
            hipLaunchKernelGGL(lzs_decompress_blocks_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                               (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                               (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, concat);
        else
            hipLaunchKernelGGL(lzs_decompress_blocks_v2_kernel, dim3(grid), dim3(kWavesPerWG * 64), 0,
                               (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                               (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, concat);
        return (int)hipGetLastError();
    }
#endif
    // The kernel bounds a wavefront's input loads by one 32-bit extent over its streams.  Streams so
    // far apart that eight of them do not fit (strides of hundreds of MB) go one to a wavefront.
    const unsigned long long longest = d_in_len ? 0xC0000400ull : in_len;
    const uint32_t per_wave = (unsigned long long)in_stride * (kDecGroups - 1u) + longest < 0xFFFFFF00ull ? kDecGroups : 1u;
    if (concat)
        hipLaunchKernelGGL(lzs_decompress_blocks_grp_kernel<true>, dim3((nblocks + per_wave - 1) / per_wave), dim3(64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, concat, per_wave);
    else
        hipLaunchKernelGGL(lzs_decompress_blocks_grp_kernel<false>, dim3((nblocks + per_wave - 1) / per_wave), dim3(64), 0,
                           (hipStream_t)stream, (uint8_t *)d_out, out_stride, out_cap, d_out_len,
                           (const uint8_t *)d_in, in_stride, d_in_len, in_len, nblocks, concat, per_wave);
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

#endif  // !LZS_TU_COMPRESS
#ifndef LZS_TU_DECOMPRESS
int lzs_hip_launch_compress_segments(void *d_slots, size_t slot_stride, const void *d_in, uint32_t n,
                                     uint32_t seg, uint32_t nseg, const uint32_t *d_entry,
                                     const uint8_t *d_dirty, uint32_t *d_exit, uint64_t *d_nbits,
                                     void *d_out, const uint64_t *d_bit_at, uint32_t lim, uint32_t *d_open,
                                     void *stream)
{
    if (nseg == 0) return 0;
    int chain_mode = 0;
    { const int e = lzs_hip_chain_mode(stream, &chain_mode); if (e) return e; }
    if (chain_mode != 0)
        hipLaunchKernelGGL(wgv_safe::lzs_compress_segments_kernel, dim3(nseg), dim3(wgv_safe::kWgThreads), 0, (hipStream_t)stream,
                           (uint8_t *)d_slots, slot_stride, (const uint8_t *)d_in, n, seg, nseg,
                           d_entry, d_dirty, d_exit, (unsigned long long *)d_nbits,
                           (uint8_t *)d_out, (const unsigned long long *)d_bit_at, lim, d_open);
    else
        hipLaunchKernelGGL(wgv_text::lzs_compress_segments_kernel, dim3(nseg), dim3(wgv_text::kWgThreads), 0, (hipStream_t)stream,
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

#endif  // !LZS_TU_DECOMPRESS
#ifndef LZS_TU_COMPRESS
int lzs_hip_launch_scan_stream(const void *d_in, uint32_t n, uint32_t nseg, const uint32_t *d_entry,
                               const uint8_t *d_dirty, uint32_t *d_exit, uint32_t *d_count,
                               uint8_t *d_all_ones, uint32_t *d_marks, int compare, uint32_t seg, int concat,
                               const uint32_t *d_seg_base, const uint32_t *d_seg_end, uint32_t in_extent, void *stream)
{
    if (nseg == 0) return 0;
    static const int old_scan = [] { const char *v = getenv("LZS_SCAN"); return v && v[0] == 'w'; }();   // LZS_SCAN=wave: A/B
    if (!old_scan && (d_all_ones || compare)) {
        // a lane per segment; the first round with the all-0xFF flags, the later ones against its marks
        static const int g8_scan = [] { const char *v = getenv("LZS_SCAN"); return v && v[0] == 'g'; }();   // LZS_SCAN=g8: A/B
        if (d_all_ones)
            hipLaunchKernelGGL(lzs_all_ones_kernel, dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                               (const uint8_t *)d_in, n, nseg, d_all_ones, seg, d_seg_base, d_seg_end);
#define LZS_LAUNCH_SCAN(L, C, GRID) hipLaunchKernelGGL((lzs_scan_stream_g8_kernel<L, C>), dim3(GRID), dim3(64), 0, (hipStream_t)stream, \
                               (const uint8_t *)d_in, n, nseg, d_entry, d_dirty, d_exit, d_count, d_marks, compare ? 1u : 0u, seg, \
                               concat ? 1u : 0u, d_seg_base, d_seg_end, in_extent)
        if (g8_scan) {
            if (concat) LZS_LAUNCH_SCAN(8, true, (nseg + 7) / 8); else LZS_LAUNCH_SCAN(8, false, (nseg + 7) / 8);
        } else {
            if (d_marks && !compare)     // no marks yet (one store per lane and mark otherwise: 128 scattered words a segment)
                (void)hipMemsetAsync(d_marks, 0xFF, (size_t)nseg * kScanMarkWords * sizeof(uint32_t), (hipStream_t)stream);
            if (concat) LZS_LAUNCH_SCAN(1, true, (nseg + 63) / 64); else LZS_LAUNCH_SCAN(1, false, (nseg + 63) / 64);
        }
#undef LZS_LAUNCH_SCAN
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(lzs_scan_stream_kernel, dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)d_in, n, nseg, d_entry, d_dirty, d_exit, d_count, d_all_ones,
                       d_marks, compare ? 1u : 0u, seg, concat ? 1u : 0u, d_seg_base, d_seg_end);
    return (int)hipGetLastError();
}

int lzs_hip_launch_decode_stream(void *d_out, uint32_t cap, uint32_t *d_origin, uint32_t *d_tainted,
                                 const void *d_in, uint32_t n, uint32_t in_extent, uint32_t nseg, const uint32_t *d_entry,
                                 const uint32_t *d_out_start, uint32_t seg, int concat,
                                 const uint32_t *d_seg_base, const uint32_t *d_seg_end,
                                 const uint32_t *d_out_floor, const uint32_t *d_out_limit, void *stream)
{
    if (nseg == 0) return 0;
    static const int old_decode = [] { const char *v = getenv("LZS_SEG_DECODE"); return v && v[0] == 'w'; }();   // LZS_SEG_DECODE=wave: A/B
    // streams that expand more than sixfold are runs: there one wavefront per segment wins (60 bytes a step)
    // (n == 0: a batch of blocks with segment tables -- the expansion is not known here)
    if (!old_decode && (n == 0u || (unsigned long long)cap <= 6ull * n)) {
        // between a quarter and nine tenths of its output a stream is mostly short matches (text: 0.57):
        // there a second token per trip pays (lzs_decompress_blocks_grp has the numbers)
        static const int one_token = [] { return getenv("LZS_DEC_ONE_TOKEN") != nullptr; }();
        const bool two = !one_token && n != 0u && 4ull * n > cap && 10ull * n < 9ull * cap;
        const bool wide = !one_token && n != 0u && 10ull * n >= 9ull * cap;      // mostly literals: the 96-bit buffer
#define LZS_LAUNCH_G8(T, W, C) hipLaunchKernelGGL((lzs_decode_stream_g8_kernel<T, W, C>), dim3((nseg + 7) / 8), dim3(64), 0, (hipStream_t)stream, \
                               (uint8_t *)d_out, cap, d_origin, d_tainted, (const uint8_t *)d_in, n, in_extent, nseg, d_entry, d_out_start, seg, concat ? 1u : 0u, \
                               d_seg_base, d_seg_end, d_out_floor, d_out_limit)
        if (concat) {
            if (two) LZS_LAUNCH_G8(true, false, true);
            else if (wide) LZS_LAUNCH_G8(false, true, true);
            else LZS_LAUNCH_G8(false, false, true);
        } else {
            if (two) LZS_LAUNCH_G8(true, false, false);
            else if (wide) LZS_LAUNCH_G8(false, true, false);
            else LZS_LAUNCH_G8(false, false, false);
        }
#undef LZS_LAUNCH_G8
        return (int)hipGetLastError();
    }
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
                                  uint32_t *d_left, int last, void *stream)
{
    if (total == 0) return 0;
    uint32_t grid = (total / 4u + 255u) / 256u + 1u;          // four bytes a thread
    if (grid > 65536u) grid = 65536u;
    hipLaunchKernelGGL(lzs_resolve_stream_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, d_origin, total, round, d_left, last ? 1u : 0u);
    return (int)hipGetLastError();
}

int lzs_hip_launch_resolve_tails(void *d_out, uint32_t *d_origin, uint32_t total, const uint32_t *d_seg_start,
                                 uint32_t nseg, uint32_t stride, uint32_t round, uint32_t *d_left, void *stream)
{
    if (total == 0 || nseg == 0 || stride == 0) return 0;
    hipLaunchKernelGGL(lzs_resolve_tails_kernel, dim3((nseg + stride - 1u) / stride), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, d_origin, total, d_seg_start, stride, round, d_left);
    return (int)hipGetLastError();
}

int lzs_hip_launch_resolve_chunks(void *d_out, uint32_t *d_origin, uint32_t total, const uint32_t *d_seg_start,
                                  uint32_t nseg, uint32_t per_chunk, uint32_t round, void *stream)
{
    if (total == 0 || nseg < 2 || per_chunk == 0) return 0;
    hipLaunchKernelGGL(lzs_resolve_chunks_kernel, dim3((nseg + per_chunk - 1u) / per_chunk), dim3(256), 0, (hipStream_t)stream,
                       (uint8_t *)d_out, d_origin, total, d_seg_start, nseg, per_chunk, round);
    return (int)hipGetLastError();
}

unsigned lzs_hip_dec_segment_bytes(void) { return kDecSegMax; }

#endif  // !LZS_TU_COMPRESS
#ifndef LZS_TU_DECOMPRESS
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

#endif  // !LZS_TU_DECOMPRESS
}  // extern "C"
