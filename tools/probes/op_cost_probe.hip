// Development probe: SIMD time per wave-instruction for single opcodes and short patterns on
// gfx950, one kernel per pattern (no run-time switch in the timed loop), at 1, 2, 4, 5, 6, 8 waves
// per SIMD with all 256 CUs busy.  Each pattern is 8 instructions on 8 independent chains
// (%0..%7), %8 is loop-invariant, repeated 8 x per loop trip.
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

#define PATTERNS(X) \
 X(0,  8, "v_xor_b32 (VOP2)", "v_xor_b32 %0, %0, %8\nv_xor_b32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_xor_b32 %3, %3, %8\nv_xor_b32 %4, %4, %8\nv_xor_b32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_xor_b32 %7, %7, %8\n") \
 X(1,  8, "v_ffbl_b32", "v_ffbl_b32 %0, %0\nv_ffbl_b32 %1, %1\nv_ffbl_b32 %2, %2\nv_ffbl_b32 %3, %3\nv_ffbl_b32 %4, %4\nv_ffbl_b32 %5, %5\nv_ffbl_b32 %6, %6\nv_ffbl_b32 %7, %7\n") \
 X(2,  8, "v_alignbyte_b32 (vgpr shift)", "v_alignbyte_b32 %0, %0, %8, %1\nv_alignbyte_b32 %1, %1, %8, %2\nv_alignbyte_b32 %2, %2, %8, %3\nv_alignbyte_b32 %3, %3, %8, %4\nv_alignbyte_b32 %4, %4, %8, %5\nv_alignbyte_b32 %5, %5, %8, %6\nv_alignbyte_b32 %6, %6, %8, %7\nv_alignbyte_b32 %7, %7, %8, %0\n") \
 X(3,  8, "v_min3_u32", "v_min3_u32 %0, %0, %8, %1\nv_min3_u32 %1, %1, %8, %2\nv_min3_u32 %2, %2, %8, %3\nv_min3_u32 %3, %3, %8, %4\nv_min3_u32 %4, %4, %8, %5\nv_min3_u32 %5, %5, %8, %6\nv_min3_u32 %6, %6, %8, %7\nv_min3_u32 %7, %7, %8, %0\n") \
 X(4,  8, "v_min_u32 (VOP2)", "v_min_u32 %0, %0, %8\nv_min_u32 %1, %1, %8\nv_min_u32 %2, %2, %8\nv_min_u32 %3, %3, %8\nv_min_u32 %4, %4, %8\nv_min_u32 %5, %5, %8\nv_min_u32 %6, %6, %8\nv_min_u32 %7, %7, %8\n") \
 X(5,  8, "v_max_i32 (VOP2)", "v_max_i32 %0, %0, %8\nv_max_i32 %1, %1, %8\nv_max_i32 %2, %2, %8\nv_max_i32 %3, %3, %8\nv_max_i32 %4, %4, %8\nv_max_i32 %5, %5, %8\nv_max_i32 %6, %6, %8\nv_max_i32 %7, %7, %8\n") \
 X(6,  8, "v_cndmask_b32 x8, same vcc", "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n") \
 X(7,  8, "v_cndmask_b32 x8, sgpr pair mask", "v_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\nv_cndmask_b32 %4, %4, %8, s[20:21]\nv_cndmask_b32 %5, %5, %8, s[20:21]\nv_cndmask_b32 %6, %6, %8, s[20:21]\nv_cndmask_b32 %7, %7, %8, s[20:21]\n") \
 X(8,  8, "v_cmp_lt_u32 -> vcc x8", "v_cmp_lt_u32 vcc, %0, %8\nv_cmp_lt_u32 vcc, %1, %8\nv_cmp_lt_u32 vcc, %2, %8\nv_cmp_lt_u32 vcc, %3, %8\nv_cmp_lt_u32 vcc, %4, %8\nv_cmp_lt_u32 vcc, %5, %8\nv_cmp_lt_u32 vcc, %6, %8\nv_cmp_lt_u32 vcc, %7, %8\n") \
 X(9,  8, "v_cmp vcc + v_cndmask pairs", "v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\nv_cmp_lt_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n") \
 X(10, 8, "v_cmp vcc + 3 v_cndmask", "v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cmp_lt_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n") \
 X(11, 8, "v_cndmask vcc alternating with v_xor", "v_cndmask_b32 %0, %0, %8, vcc\nv_xor_b32 %1, %1, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_xor_b32 %3, %3, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_xor_b32 %5, %5, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_xor_b32 %7, %7, %8\n") \
 X(12, 8, "v_bfi_b32", "v_bfi_b32 %0, %0, %8, %1\nv_bfi_b32 %1, %1, %8, %2\nv_bfi_b32 %2, %2, %8, %3\nv_bfi_b32 %3, %3, %8, %4\nv_bfi_b32 %4, %4, %8, %5\nv_bfi_b32 %5, %5, %8, %6\nv_bfi_b32 %6, %6, %8, %7\nv_bfi_b32 %7, %7, %8, %0\n") \
 X(13, 8, "v_lshlrev_b32 imm", "v_lshlrev_b32 %0, 3, %0\nv_lshlrev_b32 %1, 3, %1\nv_lshlrev_b32 %2, 3, %2\nv_lshlrev_b32 %3, 3, %3\nv_lshlrev_b32 %4, 3, %4\nv_lshlrev_b32 %5, 3, %5\nv_lshlrev_b32 %6, 3, %6\nv_lshlrev_b32 %7, 3, %7\n") \
 X(14, 8, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %8\nv_lshl_add_u32 %1, %1, 1, %8\nv_lshl_add_u32 %2, %2, 1, %8\nv_lshl_add_u32 %3, %3, 1, %8\nv_lshl_add_u32 %4, %4, 1, %8\nv_lshl_add_u32 %5, %5, 1, %8\nv_lshl_add_u32 %6, %6, 1, %8\nv_lshl_add_u32 %7, %7, 1, %8\n") \
 X(15, 8, "v_add_u32 / v_sub_u32 / v_and / v_or mix", "v_add_u32 %0, %0, %8\nv_sub_u32 %1, %1, %8\nv_and_b32 %2, %2, %8\nv_or_b32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_sub_u32 %5, %5, %8\nv_and_b32 %6, %6, %8\nv_or_b32 %7, %7, %8\n") \
 X(16, 8, "v_ashrrev_i32", "v_ashrrev_i32 %0, 31, %0\nv_ashrrev_i32 %1, 31, %1\nv_ashrrev_i32 %2, 31, %2\nv_ashrrev_i32 %3, 31, %3\nv_ashrrev_i32 %4, 31, %4\nv_ashrrev_i32 %5, 31, %5\nv_ashrrev_i32 %6, 31, %6\nv_ashrrev_i32 %7, 31, %7\n") \
 X(17, 8, "v_mov_b32", "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n") \
 X(18, 8, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %8\nv_mul_u32_u24 %1, %1, %8\nv_mul_u32_u24 %2, %2, %8\nv_mul_u32_u24 %3, %3, %8\nv_mul_u32_u24 %4, %4, %8\nv_mul_u32_u24 %5, %5, %8\nv_mul_u32_u24 %6, %6, %8\nv_mul_u32_u24 %7, %7, %8\n") \
 X(19, 8, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8\n") \
 X(20, 8, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 11\nv_bfe_u32 %1, %1, 3, 11\nv_bfe_u32 %2, %2, 3, 11\nv_bfe_u32 %3, %3, 3, 11\nv_bfe_u32 %4, %4, 3, 11\nv_bfe_u32 %5, %5, 3, 11\nv_bfe_u32 %6, %6, 3, 11\nv_bfe_u32 %7, %7, 3, 11\n") \
 X(21, 8, "v_perm_b32", "v_perm_b32 %0, %0, %8, %1\nv_perm_b32 %1, %1, %8, %2\nv_perm_b32 %2, %2, %8, %3\nv_perm_b32 %3, %3, %8, %4\nv_perm_b32 %4, %4, %8, %5\nv_perm_b32 %5, %5, %8, %6\nv_perm_b32 %6, %6, %8, %7\nv_perm_b32 %7, %7, %8, %0\n") \
 X(22, 8, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0\nv_readfirstlane_b32 s21, %1\nv_readfirstlane_b32 s22, %2\nv_readfirstlane_b32 s23, %3\nv_readfirstlane_b32 s24, %4\nv_readfirstlane_b32 s25, %5\nv_readfirstlane_b32 s26, %6\nv_readfirstlane_b32 s27, %7\n") \
 X(23, 8, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n") \
 X(24, 8, "s_add_u32 (scalar unit)", "s_add_u32 s20, s20, 7\ns_add_u32 s21, s21, 7\ns_add_u32 s22, s22, 7\ns_add_u32 s23, s23, 7\ns_add_u32 s24, s24, 7\ns_add_u32 s25, s25, 7\ns_add_u32 s26, s26, 7\ns_add_u32 s27, s27, 7\n") \
 X(25, 16, "v_xor + s_add pairs (per instruction)", "v_xor_b32 %0, %0, %8\ns_add_u32 s20, s20, 7\nv_xor_b32 %1, %1, %8\ns_add_u32 s21, s21, 7\nv_xor_b32 %2, %2, %8\ns_add_u32 s22, s22, 7\nv_xor_b32 %3, %3, %8\ns_add_u32 s23, s23, 7\nv_xor_b32 %4, %4, %8\ns_add_u32 s24, s24, 7\nv_xor_b32 %5, %5, %8\ns_add_u32 s25, s25, 7\nv_xor_b32 %6, %6, %8\ns_add_u32 s26, s26, 7\nv_xor_b32 %7, %7, %8\ns_add_u32 s27, s27, 7\n") \
 X(26, 8, "v_and_or_b32 / v_or3 (VOP3 logic)", "v_and_or_b32 %0, %0, %8, %1\nv_or3_b32 %1, %1, %8, %2\nv_and_or_b32 %2, %2, %8, %3\nv_or3_b32 %3, %3, %8, %4\nv_and_or_b32 %4, %4, %8, %5\nv_or3_b32 %5, %5, %8, %6\nv_and_or_b32 %6, %6, %8, %7\nv_or3_b32 %7, %7, %8, %0\n") \
 X(27, 8, "v_xor_b32_e64 (VOP3 encoding)", "v_xor_b32_e64 %0, %0, %8\nv_xor_b32_e64 %1, %1, %8\nv_xor_b32_e64 %2, %2, %8\nv_xor_b32_e64 %3, %3, %8\nv_xor_b32_e64 %4, %4, %8\nv_xor_b32_e64 %5, %5, %8\nv_xor_b32_e64 %6, %6, %8\nv_xor_b32_e64 %7, %7, %8\n") \
 X(28, 8, "v_add3_u32", "v_add3_u32 %0, %0, %8, %1\nv_add3_u32 %1, %1, %8, %2\nv_add3_u32 %2, %2, %8, %3\nv_add3_u32 %3, %3, %8, %4\nv_add3_u32 %4, %4, %8, %5\nv_add3_u32 %5, %5, %8, %6\nv_add3_u32 %6, %6, %8, %7\nv_add3_u32 %7, %7, %8, %0\n") \
 X(29, 8, "v_cmp -> sgpr pair x8", "v_cmp_lt_u32 s[20:21], %0, %8\nv_cmp_lt_u32 s[22:23], %1, %8\nv_cmp_lt_u32 s[24:25], %2, %8\nv_cmp_lt_u32 s[26:27], %3, %8\nv_cmp_lt_u32 s[20:21], %4, %8\nv_cmp_lt_u32 s[22:23], %5, %8\nv_cmp_lt_u32 s[24:25], %6, %8\nv_cmp_lt_u32 s[26:27], %7, %8\n")

template <int OP>
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
#define X(id, n, name, text) if (OP == id) asm volatile(REP8(text) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k) \
                                                       : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        PATTERNS(X)
#undef X
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

template <int OP>
void run(Rec *d, const char *name, int per)
{
    const int iters = 1000;
    hipFuncSetAttribute((const void *)rate<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-40s", name);
    for (int w : {1, 2, 4, 5, 6, 8}) {
        const size_t lds = (160 * 1024 / w) & ~255u;
        const int nblk = 256 * w;
        std::vector<Rec> h(nblk * 4);
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(rate<OP>, dim3(nblk), dim3(256), lds, 0, d, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), d, sizeof(Rec) * nblk * 4, hipMemcpyDeviceToHost);
        std::map<uint64_t, std::vector<Rec>> by;
        for (auto &r : h) by[((uint64_t)(r.xcc & 15u) << 32) | (r.hw & 0xFF30u)].push_back(r);
        std::vector<double> cyc;
        for (auto &kv : by) {
            unsigned long long a = ~0ull, b = 0;
            for (auto &r : kv.second) { a = std::min(a, r.t0); b = std::max(b, r.t1); }
            cyc.push_back((double)(b - a) / ((double)kv.second.size() * iters * 8 * per));
        }
        std::sort(cyc.begin(), cyc.end());
        printf(" %6.2f", cyc[cyc.size() / 2]);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * 256 * 8 * 4);
    printf("SIMD cycles per wave-instruction by waves per SIMD (all 256 CUs busy)\n%-40s %6s %6s %6s %6s %6s %6s\n", "pattern", "w=1", "w=2", "w=4", "w=5", "w=6", "w=8");
#define X(id, n, name, text) run<id>(d, name, n);
    PATTERNS(X)
#undef X
    return 0;
}
