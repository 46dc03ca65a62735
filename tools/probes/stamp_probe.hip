#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
__global__ void probe(unsigned long long *out, uint32_t seed)
{
    uint32_t v = seed + threadIdx.x, w = seed * 3 + 1;
    unsigned long long t0, t1, a0, a1;
    // (a) builtin stamps, as lat_probe
    t0 = __builtin_readcyclecounter();
    asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w));
    t1 = __builtin_readcyclecounter();
    // (b) stamps that wait for their own return before anything else is issued
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a0) :: "memory");
    asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a1) :: "memory");
    // (c) builtin stamps around 256 instructions
    unsigned long long c0 = __builtin_readcyclecounter();
    asm volatile(REP64("v_add_u32 %0, %0, %1\n") REP64("v_add_u32 %0, %0, %1\n") REP64("v_add_u32 %0, %0, %1\n") REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w));
    unsigned long long c1 = __builtin_readcyclecounter();
    // (d) realtime (100 MHz) around 64 x 256 instructions
    unsigned long long r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    for (int i = 0; i < 64; i++)
        asm volatile(REP64("v_add_u32 %0, %0, %1\n") REP64("v_add_u32 %0, %0, %1\n") REP64("v_add_u32 %0, %0, %1\n") REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w));
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a1 - a0; out[2] = c1 - c0; out[3] = r1 - r0; }
    if (v == 0x7fffffff) out[31] = v;
}
int main()
{
    unsigned long long *d, h[32] = {0};
    hipMalloc(&d, sizeof h);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 12345u + rep);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("64 dependent v_add: builtin stamps %llu ticks; waited stamps %llu ticks; 256 instr builtin %llu ticks; 16384 instr take %llu ticks of the 100 MHz clock = %.2f ns per instruction\n",
               h[0], h[1], h[2], h[3], h[3] * 10.0 / 16384);
    }
    return 0;
}
