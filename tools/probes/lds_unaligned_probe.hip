// Does gfx950 serve byte-aligned ds_read_b64 / ds_read_b96 correctly, and what do they cost
// next to "four aligned words + three v_alignbyte" (ringm_read12 in kernels/common.inc)?
// One workgroup of 1024 threads (4 waves per SIMD), pseudo-random byte offsets in a 16 KiB table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

constexpr uint32_t kBytes = 16384, kIters = 512;

template <int MODE>
__global__ void __launch_bounds__(1024) probe(const uint8_t *src, uint32_t *out, unsigned long long *ticks, uint32_t seed)
{
    __shared__ uint32_t tab[kBytes / 4 + 8];
    for (uint32_t i = threadIdx.x; i < kBytes / 4 + 8; i += 1024) tab[i] = ((const uint32_t *)src)[i];
    __syncthreads();
    uint32_t q = (threadIdx.x * 2654435761u + seed) % (kBytes - 16), acc = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (uint32_t it = 0; it < kIters; it++) {
        uint32_t w0, w1, w2;
        if (MODE == 0) {
            const uint32_t a = q >> 2;
            const uint32_t d0 = tab[a], d1 = tab[a + 1], d2 = tab[a + 2], d3 = tab[a + 3];
            w0 = __builtin_amdgcn_alignbyte(d1, d0, q);
            w1 = __builtin_amdgcn_alignbyte(d2, d1, q);
            w2 = __builtin_amdgcn_alignbyte(d3, d2, q);
        } else if (MODE == 1) {
            uint32_t addr = (uint32_t)(uintptr_t)tab + q;
            typedef uint32_t v3 __attribute__((ext_vector_type(3)));
            v3 r;
            asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
            w0 = r.x; w1 = r.y; w2 = r.z;
        } else {
            uint32_t addr = (uint32_t)(uintptr_t)tab + q;
            unsigned long long r; uint32_t r2;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(r), "=v"(r2) : "v"(addr) : "memory");
            w0 = (uint32_t)r; w1 = (uint32_t)(r >> 32); w2 = r2;
        }
        acc = acc * 31 + (w0 ^ (w1 << 1) ^ (w2 << 2));
        q = (q + 4099u + (w0 & 3u)) % (kBytes - 16);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

static uint32_t host(const uint8_t *s, uint32_t tid, uint32_t seed)
{
    uint32_t q = (tid * 2654435761u + seed) % (kBytes - 16), acc = 0;
    for (uint32_t it = 0; it < kIters; it++) {
        uint32_t w0, w1, w2;
        memcpy(&w0, s + q, 4); memcpy(&w1, s + q + 4, 4); memcpy(&w2, s + q + 8, 4);
        acc = acc * 31 + (w0 ^ (w1 << 1) ^ (w2 << 2));
        q = (q + 4099u + (w0 & 3u)) % (kBytes - 16);
    }
    return acc;
}

int main()
{
    static uint8_t h[kBytes + 32];
    uint32_t x = 12345;
    for (uint32_t i = 0; i < kBytes + 32; i++) { x = x * 1664525u + 1013904223u; h[i] = (uint8_t)(x >> 24); }
    uint8_t *d; uint32_t *o; unsigned long long *t;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 4096); hipMalloc(&t, 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const char *names[3] = {"4 aligned words + 3 v_alignbyte", "ds_read_b96 at a byte address", "ds_read_b64 + ds_read_b32 at a byte address"};
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 2; rep++) {
            uint32_t res[1024]; unsigned long long ticks;
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(1024), 0, 0, d, o, t, 7u + rep);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(1024), 0, 0, d, o, t, 7u + rep);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(1024), 0, 0, d, o, t, 7u + rep);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed (%s)\n", names[mode], hipGetErrorString(hipGetLastError())); return 1; }
            hipMemcpy(res, o, 4096, hipMemcpyDeviceToHost); hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
            uint32_t bad = 0;
            for (uint32_t i = 0; i < 1024; i++) bad += res[i] != host(h, i, 7u + rep);
            printf("%-46s: %u wrong lanes of 1024; %.1f ticks per iteration (wave 0, 4 waves/SIMD)\n", names[mode], bad, (double)ticks / kIters);
        }
    return 0;
}
