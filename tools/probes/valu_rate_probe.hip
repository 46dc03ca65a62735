// Development probe: VALU issue rate on gfx950 as a function of the waves resident per SIMD
// (4 cycles per wave64 instruction, as a lone wave sees, or 2, as
// MI355X_MICROARCH.md gives for a SIMD-32?).  One kernel per instruction mix, waves per SIMD set by
// the dynamic LDS size, every wave stamps s_memtime / s_memrealtime itself.  Reported per
// occupancy: SIMD time per wave-instruction in shader cycles and in ns, and what a single wave
// sees.  Answer (profiles/r02/valu_rate_probe.txt): both -- plain VOP2 integer ops (add, sub,
// and, or, xor, mov, ashr, cndmask) reach 2.2 cycles from two waves per SIMD on; a mix with
// v_ffbl / v_alignbyte / v_min3 / shifts in it stays at 4.0-4.2 whatever the occupancy
// (op_cost_probe.hip prices the opcodes one by one).  (A first version of this probe selected the
// mix with a run-time switch inside the timed loop and read 2x these numbers for every case.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>
#define REP8(x) x x x x x x x x
struct Rec { unsigned long long t0, t1, r0, r1; uint32_t hw, xcc; };
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define RSTAMP(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")

template <int MODE>
__global__ void rate(Rec *out, int iters)
{
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = ~a, g = a << 3, h = a + 99;
    const uint32_t k = blockIdx.x + 12345u;
    lds[threadIdx.x] = a;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    RSTAMP(r0); STAMP(t0);
    for (int i = 0; i < iters; i++) {
        if (MODE == 0)
            asm volatile(REP8("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k));
        else if (MODE == 1)
            asm volatile(REP8("v_xor_b32 %0, %0, %8\nv_ffbl_b32 %1, %0\nv_alignbyte_b32 %2, %2, %8, 1\nv_min3_u32 %3, %3, %8, %2\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\nv_add_u32 %5, %5, %8\nv_lshlrev_b32 %6, 1, %6\nv_and_b32 %7, %7, %8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k) : "vcc");
        else
            asm volatile(REP8(REP8("v_add_u32 %0, %0, %1\n")) : "+v"(a) : "v"(k));
    }
    STAMP(t1); RSTAMP(r1);
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1; r.hw = hw; r.xcc = xcc;
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
    if (a + b + c + d + e + f + g + h == 0x12345678u) out[0].hw = a;
}

template <int MODE>
void run(Rec *d, const char *name)
{
    const int iters = 2000;
    hipFuncSetAttribute((const void *)rate<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    {   // one wave alone on the chip
        Rec h;
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(rate<MODE>, dim3(1), dim3(64), 1024, 0, d, iters); hipDeviceSynchronize(); }
        hipMemcpy(&h, d, sizeof(Rec), hipMemcpyDeviceToHost);
        printf("%-32s one wave alone on the chip: %.2f cycles = %.2f ns per instruction\n", name, (double)(h.t1 - h.t0) / (iters * 64.0), (double)(h.r1 - h.r0) * 10.0 / (iters * 64.0));
    }
    for (int w : {1, 2, 3, 4, 5, 6, 7, 8}) {
        const size_t lds = (160 * 1024 / w) & ~255u;
        const int nblk = 256 * w;
        std::vector<Rec> h(nblk * 4);
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(rate<MODE>, dim3(nblk), dim3(256), lds, 0, d, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), d, sizeof(Rec) * nblk * 4, hipMemcpyDeviceToHost);
        std::map<uint64_t, std::vector<Rec>> by;
        for (auto &r : h) by[((uint64_t)(r.xcc & 15u) << 32) | (r.hw & 0xFF30u)].push_back(r);
        std::vector<double> cyc, ns, own; double wsum = 0;
        for (auto &kv : by) {
            unsigned long long a = ~0ull, b = 0, ra = ~0ull, rb = 0;
            for (auto &r : kv.second) { a = std::min(a, r.t0); b = std::max(b, r.t1); ra = std::min(ra, r.r0); rb = std::max(rb, r.r1); own.push_back((double)(r.t1 - r.t0) / (iters * 64.0)); }
            cyc.push_back((double)(b - a) / ((double)kv.second.size() * iters * 64));
            ns.push_back((double)(rb - ra) * 10.0 / ((double)kv.second.size() * iters * 64));
            wsum += kv.second.size();
        }
        std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end()); std::sort(own.begin(), own.end());
        printf("%-32s %d waves per SIMD (%.2f seen on %zu SIMDs): SIMD time per wave-instruction %.2f cycles = %.2f ns; a wave's own pace %.2f cycles per instruction\n",
               name, w, wsum / by.size(), by.size(), cyc[cyc.size() / 2], ns[ns.size() / 2], own[own.size() / 2]);
    }
}

int main()
{
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * 256 * 8 * 4);
    run<2>(d, "v_add dependent chain");
    run<0>(d, "v_xor, 8 independent chains");
    run<1>(d, "compress-step mix, independent");
    return 0;
}
