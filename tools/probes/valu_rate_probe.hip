// Development probe: VALU issue rate on gfx950 as a function of the waves resident per SIMD.
// is a wave64 VALU instruction 4 cycles (what a lone wave sees, lat_probe) or 2
// (MI355X_MICROARCH.md: SIMD-32, 2 passes) once several waves share the SIMD?  Every wave runs
// `iters` x 64 register-only instructions between two s_memtime stamps; per SIMD (told apart by
// HW_ID / XCC_ID) the probe reports  (last end - first start) / (waves x instructions)  =
// cycles of SIMD time per wave-instruction at that occupancy.
//   mode 0: v_xor_b32, 8 independent chains      mode 1: the compress step's mix (v_xor, v_ffbl,
//   v_alignbyte, v_min3, v_cndmask, v_add, v_lshlrev, v_and), independent
//   mode 2: one dependent chain of v_add          mode 3: mode 1 with one ds_read_b32 per 8 VALU
//   mode 4: v_cmp + v_cndmask pairs (VCC traffic) mode 5: s_add (scalar unit, one per CU)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) x x x x x x x x

struct Rec { unsigned long long t0, t1, r0, r1; uint32_t hw, xcc; };

__global__ __launch_bounds__(256) void rate(Rec *out, int iters, int mode)
{
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = ~a, g = a << 3, h = a + 99;
    const uint32_t k = blockIdx.x + 12345u;
    lds[threadIdx.x] = a;
    __syncthreads();
    uint32_t addr = (threadIdx.x * 4u) & 1023u;
    unsigned long long t0, t1;
    unsigned long long r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) {
        if (mode == 0) {
            asm volatile(REP8("v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\n"
                              "v_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k));
        } else if (mode == 1) {
            asm volatile(REP8("v_xor_b32 %0, %0, %8\nv_ffbl_b32 %1, %0\nv_alignbyte_b32 %2, %2, %8, 1\nv_min3_u32 %3, %3, %8, %2\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\nv_add_u32 %5, %5, %8\nv_lshlrev_b32 %6, 1, %6\nv_and_b32 %7, %7, %8\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k) : "vcc");
        } else if (mode == 2) {
            asm volatile(REP8(REP8("v_add_u32 %0, %0, %1\n")) : "+v"(a) : "v"(k));
        } else if (mode == 3) {
            asm volatile(REP8("v_xor_b32 %0, %0, %8\nv_ffbl_b32 %1, %0\nv_alignbyte_b32 %2, %2, %8, 1\nv_min3_u32 %3, %3, %8, %2\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\nv_add_u32 %5, %5, %8\nv_lshlrev_b32 %6, 1, %6\nds_read_b32 %7, %9\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k), "v"(addr) : "vcc", "memory");
        } else if (mode == 4) {
            asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cmp_lt_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k) : "vcc");
        } else {
            uint32_t s0 = k, s1 = k + 1, s2 = k + 2, s3 = k + 3;
            asm volatile(REP8(REP4("s_add_u32 %0, %0, 7\ns_add_u32 %1, %1, 7\n") ) : "+s"(s0), "+s"(s1) :: "scc");
            asm volatile("" :: "s"(s2), "s"(s3));
            a += s0 + s1;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1) :: "memory");
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.t0 = t0; r.t1 = t1; r.r0 = r0; r.r1 = r1; r.hw = hw; r.xcc = xcc;
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
    }
    if (a + b + c + d + e + f + g + h == 0x12345678u) out[0].hw = a;      // keep the chains alive
}

int main()
{
    const int iters = 2000;
    const char *names[] = {"v_xor x8 independent", "compress-step mix, independent", "v_add dependent chain",
                           "mix + 1 ds_read_b32 per 8", "v_cmp+v_cndmask pairs", "s_add (scalar unit)"};
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * 256 * 8 * 4);
    hipFuncSetAttribute((const void *)rate, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-34s %5s %6s %10s %12s %12s %10s %10s\n", "mode", "CUs", "w/SIMD", "SIMDs seen", "cyc/instr", "waves/SIMD", "clock GHz", "ns/instr");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ncu : {1, 32, 256})
    for (int mode = 0; mode < 6; mode++) {
        for (int w = 1; w <= 8; w++) {
            if (ncu != 256 && w != 1 && w != 2 && w != 5 && w != 8) continue;
            // blocks of 256 threads = one wave per SIMD; dynamic LDS sized so that exactly w blocks fit a CU
            const size_t lds = (160 * 1024 / w) & ~255u;
            const int nblk = ncu * w;
            std::vector<Rec> h(nblk * 4);
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(rate, dim3(nblk), dim3(256), lds, 0, d, iters, mode);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, sizeof(Rec) * nblk * 4, hipMemcpyDeviceToHost);
            std::map<uint64_t, std::vector<Rec>> by;
            for (auto &r : h) {
                // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 ; XCC_ID 3:0
                const uint64_t key = ((uint64_t)(r.xcc & 15u) << 32) | (r.hw & 0xFF30u);
                by[key].push_back(r);
            }
            std::vector<double> cpi, ghz; double wsum = 0;
            for (auto &r : h) ghz.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * 0.1);
            std::sort(ghz.begin(), ghz.end());
            for (auto &kv : by) {
                unsigned long long a = ~0ull, b = 0;
                for (auto &r : kv.second) { a = std::min(a, r.t0); b = std::max(b, r.t1); }
                cpi.push_back((double)(b - a) / ((double)kv.second.size() * iters * 64));
                wsum += kv.second.size();
            }
            std::sort(cpi.begin(), cpi.end());
            const double c = cpi[cpi.size() / 2], f = ghz[ghz.size() / 2];
            printf("%-34s %5d %6d %10zu %12.3f %12.2f %10.3f %10.3f\n", names[mode], ncu, w, by.size(), c, wsum / by.size(), f, c / f);
        }
    }
    return 0;
}
