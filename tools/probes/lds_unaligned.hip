// Development probe: do the DS loads of gfx950 take byte addresses that are not multiples of their size (the driver sets
// SH_MEM_CONFIG's alignment mode to "unaligned"), and what do they cost?  Prints what each width returns at addresses
// 0..7 + 16 * lane against the bytes that are there, and ticks per instruction for aligned and unaligned addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k_check(uint32_t *out, int shift)
{
    __shared__ uint8_t lds[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)lds + 16u * threadIdx.x + shift;    // LDS byte address
    uint32_t a; uint64_t b; uint32_t c0, c1, c2; uint32_t d0, d1, d2, d3;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u3 c; u4 d;
    asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(addr) : "memory");
    uint32_t *o = out + threadIdx.x * 10;
    o[0] = a; o[1] = (uint32_t)b; o[2] = (uint32_t)(b >> 32); o[3] = c.x; o[4] = c.y; o[5] = c.z; o[6] = d.x; o[7] = d.y; o[8] = d.z; o[9] = d.w;
}
template <int W>
__global__ void k_time(uint64_t *out, uint32_t stride, uint32_t shift)
{
    __shared__ uint8_t lds[16384 + 64];
    for (int i = threadIdx.x; i < 16384 + 64; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)lds + ((stride * (threadIdx.x & 63u)) & 16383u & ~15u) + shift;
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    uint32_t a0, a1, a2, a3; uint64_t b0, b1; u3 c0, c1; uint32_t acc = 0;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < 256; i++) {
        if (W == 4) { asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:64\n ds_read_b32 %2, %4 offset:128\n ds_read_b32 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr) : "memory"); acc += a0 + a1 + a2 + a3; }
        if (W == 8) { asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:64\n ds_read_b64 %0, %2 offset:128\n ds_read_b64 %1, %2 offset:192\n s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1) : "v"(addr) : "memory"); acc += (uint32_t)b0 + (uint32_t)b1; }
        if (W == 12) { asm volatile("ds_read_b96 %0, %2\n ds_read_b96 %1, %2 offset:64\n ds_read_b96 %0, %2 offset:128\n ds_read_b96 %1, %2 offset:192\n s_waitcnt lgkmcnt(0)" : "=v"(c0), "=v"(c1) : "v"(addr) : "memory"); acc += c0.x + c1.z; }
        if (W == 16) { asm volatile("ds_read2_b32 %0, %2 offset1:1\n ds_read2_b32 %1, %2 offset0:2 offset1:3\n ds_read2_b32 %0, %2 offset0:16 offset1:17\n ds_read2_b32 %1, %2 offset0:18 offset1:19\n s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1) : "v"(addr) : "memory"); acc += (uint32_t)b0 + (uint32_t)b1; }
        if (W == 128) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 d0, d1; asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:64\n ds_read_b128 %0, %2 offset:128\n ds_read_b128 %1, %2 offset:192\n s_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1) : "v"(addr) : "memory"); acc += d0.x + d1.w; }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (acc == 0x12345u) out[100] = 1;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t *d; hipMalloc(&d, 64 * 10 * 4); uint64_t *t; hipMalloc(&t, 128 * 8);
    uint32_t h[640];
    for (int shift = 0; shift < 4; shift++) {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d, shift);
        if (hipDeviceSynchronize() != hipSuccess) { printf("shift %d: the kernel failed\n", shift); return 1; }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int ok[4] = {1, 1, 1, 1};
        for (int l = 0; l < 64; l++) {
            uint8_t want[16]; for (int i = 0; i < 16; i++) want[i] = (uint8_t)((16 * l + shift + i) * 7 + 3);
            ok[0] &= memcmp(&h[l * 10 + 0], want, 4) == 0; ok[1] &= memcmp(&h[l * 10 + 1], want, 8) == 0;
            ok[2] &= memcmp(&h[l * 10 + 3], want, 12) == 0; ok[3] &= memcmp(&h[l * 10 + 6], want, 16) == 0;
        }
        printf("address = 16 * lane + %d: ds_read_b32 %s, b64 %s, b96 %s, b128 %s (lane 1 got %08x, the bytes there are %02x %02x %02x %02x)\n", shift,
               ok[0] ? "right" : "WRONG", ok[1] ? "right" : "WRONG", ok[2] ? "right" : "WRONG", ok[3] ? "right" : "WRONG", h[10],
               (uint8_t)((16 + shift) * 7 + 3), (uint8_t)((17 + shift) * 7 + 3), (uint8_t)((18 + shift) * 7 + 3), (uint8_t)((19 + shift) * 7 + 3));
    }
    printf("ticks per DS instruction and SIMD (4 waves per CU / 8 / 16), lane stride in bytes, address shift:\n");
    for (uint32_t stride : {16u, 80u, 208u, 1040u}) for (uint32_t shift : {0u, 4u, 8u, 12u, 1u}) {
        printf("stride %3u shift %u:", stride, shift);
        for (int W : {4, 8, 12, 16, 128}) {
            printf("  %s", W == 4 ? "b32" : W == 8 ? "b64" : W == 12 ? "b96" : W == 16 ? "2x read2_b32 (per pair)" : "b128");
            for (int wps = 1; wps <= 4; wps *= 2) {
                uint64_t hh[16]; double best = 1e30;
                for (int rep = 0; rep < 3; rep++) {
                    if (W == 4) hipLaunchKernelGGL(k_time<4>, dim3(1), dim3(256 * wps), 0, 0, t, stride, shift);
                    if (W == 8) hipLaunchKernelGGL(k_time<8>, dim3(1), dim3(256 * wps), 0, 0, t, stride, shift);
                    if (W == 12) hipLaunchKernelGGL(k_time<12>, dim3(1), dim3(256 * wps), 0, 0, t, stride, shift);
                    if (W == 16) hipLaunchKernelGGL(k_time<16>, dim3(1), dim3(256 * wps), 0, 0, t, stride, shift);
                    if (W == 128) hipLaunchKernelGGL(k_time<128>, dim3(1), dim3(256 * wps), 0, 0, t, stride, shift);
                    if (hipDeviceSynchronize() != hipSuccess) { printf(" failed\n"); return 1; }
                    hipMemcpy(hh, t, sizeof hh, hipMemcpyDeviceToHost);
                    double mx = 0; for (int w = 0; w < 4 * wps; w++) mx = hh[w] > mx ? (double)hh[w] : mx;
                    best = mx < best ? mx : best;
                }
                printf(" %6.1f", best / (256.0 * 4 * wps));
            }
        }
        printf("\n");
    }
    return 0;
}
