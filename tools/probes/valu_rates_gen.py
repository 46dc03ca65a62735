#!/usr/bin/env python3
"""Writes valu_rates.hip: what a vector instruction costs to issue on gfx950, by instruction, for 1 / 2 / 4 wavefronts per
SIMD, as a stream of independent instructions (eight destinations in turn) and as a dependent chain (one register).
{d} = the rotating 32-bit register, {q} = the rotating 64-bit pair, {b} {c} = loop-invariant vector operands,
{s} = a scalar operand.  One workgroup on one CU; s_memtime around 256 x 64 instructions."""
import sys
T = [  # name, template, instructions per template
 ("v_add_u32",        "v_add_u32 {d}, {b}, {d}", 1),
 ("v_add_u32 sgpr",   "v_add_u32 {d}, {s}, {d}", 1),
 ("v_add_u32 lit",    "v_add_u32 {d}, 0x12345, {d}", 1),
 ("v_add_u32_e64",    "v_add_u32_e64 {d}, {d}, {b}", 1),
 ("v_sub_u32",        "v_sub_u32 {d}, {b}, {d}", 1),
 ("v_and_b32",        "v_and_b32 {d}, {b}, {d}", 1),
 ("v_or_b32",         "v_or_b32 {d}, {b}, {d}", 1),
 ("v_xor_b32",        "v_xor_b32 {d}, {b}, {d}", 1),
 ("v_mov_b32",        "v_mov_b32 {d}, {b}", 1),
 ("v_min_u32",        "v_min_u32 {d}, {b}, {d}", 1),
 ("v_max_i32",        "v_max_i32 {d}, {b}, {d}", 1),
 ("v_lshlrev_b32",    "v_lshlrev_b32 {d}, {b}, {d}", 1),
 ("v_lshlrev_b32 imm","v_lshlrev_b32 {d}, 3, {d}", 1),
 ("v_lshrrev_b32",    "v_lshrrev_b32 {d}, {b}, {d}", 1),
 ("v_lshrrev_b32 imm","v_lshrrev_b32 {d}, 3, {d}", 1),
 ("v_ashrrev_i32 imm","v_ashrrev_i32 {d}, 3, {d}", 1),
 ("v_lshlrev_b64",    "v_lshlrev_b64 {q}, {b}, {q}", 1),
 ("v_lshrrev_b64",    "v_lshrrev_b64 {q}, {b}, {q}", 1),
 ("v_lshl_add_u64",   "v_lshl_add_u64 {q}, {q}, 0, {q}", 1),
 ("v_cndmask vcc",    "v_cndmask_b32 {d}, {b}, {d}, vcc", 1),
 ("v_cndmask e64 vcc","v_cndmask_b32_e64 {d}, {b}, {d}, vcc", 1),
 ("v_cndmask sgpr",   "v_cndmask_b32 {d}, {b}, {d}, s[20:21]", 1),
 ("v_cndmask imm",    "v_cndmask_b32 {d}, 0, {d}, vcc", 1),
 ("v_cmp vcc",        "v_cmp_lt_u32 vcc, {b}, {d}", 1),
 ("v_cmp sgpr",       "v_cmp_lt_u32 s[20:21], {b}, {d}", 1),
 ("v_cmp imm vcc",    "v_cmp_eq_u32 vcc, 0, {d}", 1),
 ("cmp+cnd vcc",      "v_cmp_lt_u32 vcc, {b}, {d}\n v_cndmask_b32 {d}, {c}, {d}, vcc", 2),
 ("cmp+cnd sgpr",     "v_cmp_lt_u32 s[20:21], {b}, {d}\n v_cndmask_b32 {d}, {c}, {d}, s[20:21]", 2),
 ("cmp,s_and,cnd",    "v_cmp_lt_u32 s[20:21], {b}, {d}\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_cndmask_b32 {d}, {c}, {d}, s[20:21]", 3),
 ("v_mul_u32_u24",    "v_mul_u32_u24 {d}, {b}, {d}", 1),
 ("v_mad_u32_u24",    "v_mad_u32_u24 {d}, {b}, {d}, {c}", 1),
 ("v_mul_lo_u32",     "v_mul_lo_u32 {d}, {b}, {d}", 1),
 ("v_mul_hi_u32",     "v_mul_hi_u32 {d}, {b}, {d}", 1),
 ("v_bfe_u32",        "v_bfe_u32 {d}, {d}, {b}, 5", 1),
 ("v_bfe_u32 imm",    "v_bfe_u32 {d}, {d}, 3, 5", 1),
 ("v_perm_b32",       "v_perm_b32 {d}, {d}, {b}, {c}", 1),
 ("v_ffbh_u32",       "v_ffbh_u32 {d}, {d}", 1),
 ("v_bfrev_b32",      "v_bfrev_b32 {d}, {d}", 1),
 ("v_min3_u32",       "v_min3_u32 {d}, {d}, {b}, {c}", 1),
 ("v_bitop3_b32",     "v_bitop3_b32 {d}, {d}, {b}, {c} bitop3:0x42", 1),
 ("v_lshl_add_u32",   "v_lshl_add_u32 {d}, {d}, 3, {b}", 1),
 ("v_lshl_or_b32",    "v_lshl_or_b32 {d}, {d}, 3, {b}", 1),
 ("v_add3_u32",       "v_add3_u32 {d}, {d}, {b}, {c}", 1),
 ("v_and_or_b32",     "v_and_or_b32 {d}, {d}, {b}, {c}", 1),
 ("v_or3_b32",        "v_or3_b32 {d}, {d}, {b}, {c}", 1),
 ("v_xad_u32",        "v_xad_u32 {d}, {d}, {b}, {c}", 1),
 ("v_alignbit_b32",   "v_alignbit_b32 {d}, {d}, {b}, {c}", 1),
 ("v_alignbit imm",   "v_alignbit_b32 {d}, {d}, {b}, 8", 1),
 ("v_addc_co_u32",    "v_addc_co_u32 {d}, vcc, 0, {d}, vcc", 1),
 ("v_add_co_u32",     "v_add_co_u32 {d}, vcc, {b}, {d}", 1),
 ("v_mov_b32 dpp",    "v_mov_b32_dpp {d}, {d} row_shr:1 row_mask:0xf bank_mask:0xf", 1),
 ("v_add_u32 dpp",    "v_add_u32_dpp {d}, {b}, {d} row_shr:1 row_mask:0xf bank_mask:0xf", 1),
 ("v_and_b32 sdwa",   "v_and_b32_sdwa {d}, {b}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", 1),
 ("v_mov_b32 sdwa",   "v_mov_b32_sdwa {d}, {d} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", 1),
 ("v_readlane_b32",   "v_readlane_b32 s24, {d}, 3", 1),
 ("v_readfirstlane",  "v_readfirstlane_b32 s24, {d}", 1),
 ("v_mbcnt_lo",       "v_mbcnt_lo_u32_b32 {d}, {s}, {d}", 1),
 ("v_cvt_f32_u32",    "v_cvt_f32_u32 {d}, {d}", 1),
 ("v_pk_add_u16",     "v_pk_add_u16 {d}, {d}, {b}", 1),
 ("v_sad_u8",         "v_sad_u8 {d}, {d}, {b}, {c}", 1),
 ("v_mad_u64_u32",    "v_mad_u64_u32 {q}, s[20:21], {b}, {c}, {q}", 1),
 ("ds_bpermute+wait", "ds_bpermute_b32 {d}, {b}, {d}\n s_waitcnt lgkmcnt(0)", 1),
 ("ds_read_u8+wait",  "ds_read_u8 {d}, {a}\n s_waitcnt lgkmcnt(0)", 1),
 ("ds_read_b32 (8 then wait)", "ds_read_b32 {d}, {a}", 1),
 ("ds_write_b8",      "ds_write_b8 {a}, {d}", 1),
 ("s_and_b64",        "s_and_b64 s[20:21], s[20:21], s[22:23]", 1),
 ("s_nop 0",          "s_nop 0", 1),

 ("cmp vcc, 4 cnd e32",       "v_cmp_lt_u32 vcc, {b}, {d}\n v_cndmask_b32 {d}, {c}, {d}, vcc\n v_cndmask_b32 {d}, {b}, {d}, vcc\n v_cndmask_b32 {d}, {c}, {d}, vcc\n v_cndmask_b32 {d}, {b}, {d}, vcc", 5),
 ("cmp vcc, 3 add, cnd e32",  "v_cmp_lt_u32 vcc, {b}, {d}\n v_add_u32 {d}, {b}, {d}\n v_add_u32 {d}, {c}, {d}\n v_add_u32 {d}, {b}, {d}\n v_cndmask_b32 {d}, {c}, {d}, vcc", 5),
 ("cmp vcc, 7 add, cnd e32",  "v_cmp_lt_u32 vcc, {b}, {d}\n v_add_u32 {d}, {b}, {d}\n v_add_u32 {d}, {c}, {d}\n v_add_u32 {d}, {b}, {d}\n v_add_u32 {d}, {b}, {d}\n v_add_u32 {d}, {c}, {d}\n v_add_u32 {d}, {b}, {d}\n v_add_u32 {d}, {b}, {d}\n v_cndmask_b32 {d}, {c}, {d}, vcc", 9),
 ("s_mov vcc, cnd e32",       "s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 {d}, {c}, {d}, vcc", 2),
 ("s_and vcc, 2 add, cnd e32","s_and_b64 vcc, s[20:21], s[22:23]\n v_add_u32 {d}, {b}, {d}\n v_add_u32 {d}, {c}, {d}\n v_cndmask_b32 {d}, {c}, {d}, vcc", 4),
 ("s_and vcc, cnd e64",       "s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32_e64 {d}, {c}, {d}, vcc", 2),
 ("cmp e64 s, cnd e64 s",     "v_cmp_lt_u32 s[20:21], {b}, {d}\n v_cndmask_b32 {d}, {c}, {d}, s[20:21]", 2),
 ("cnd e32, other dst",       "v_cndmask_b32 {d}, {b}, {c}, vcc", 1),
 ("cnd e64 0,1",              "v_cndmask_b32_e64 {d}, 0, 1, s[20:21]", 1),
 ("v_cmp e32 sgpr src",       "v_cmp_lt_u32 vcc, {s}, {d}", 1),
 ("v_add_u32 inline 9",       "v_add_u32 {d}, 9, {d}", 1),
 ("v_and_b32 lit",            "v_and_b32 {d}, 0x3fffffff, {d}", 1),
 ("v_and_b32 sgpr",           "v_and_b32 {d}, {s}, {d}", 1),
 ("v_mov_b32 sgpr",           "v_mov_b32 {d}, {s}", 1),
 ("v_mov_b32 inline",         "v_mov_b32 {d}, 7", 1),
 ("v_lshrrev_b32 sgpr amt",   "v_lshrrev_b32 {d}, {s}, {d}", 1),
 ("v_subrev_u32",             "v_subrev_u32 {d}, {b}, {d}", 1),
 ("v_not_b32",                "v_not_b32 {d}, {d}", 1),
 ("v_xnor_b32",               "v_xnor_b32 {d}, {b}, {d}", 1),
 ("v_bfi_b32",                "v_bfi_b32 {d}, {b}, {c}, {d}", 1),
 ("v_med3_u32",               "v_med3_u32 {d}, {d}, {b}, {c}", 1),
 ("v_max3_u32",               "v_max3_u32 {d}, {d}, {b}, {c}", 1),
 ("v_min_u32 e64",            "v_min_u32_e64 {d}, {d}, {b}", 1),
 ("v_min_i32",                "v_min_i32 {d}, {b}, {d}", 1),
 ("v_max_u32",                "v_max_u32 {d}, {b}, {d}", 1),
 ("v_min_u16",                "v_min_u16 {d}, {b}, {d}", 1),
 ("v_min_f32",                "v_min_f32 {d}, {b}, {d}", 1),
 ("v_add_f32",                "v_add_f32 {d}, {b}, {d}", 1),
 ("v_mul_f32",                "v_mul_f32 {d}, {b}, {d}", 1),
 ("v_fma_f32",                "v_fma_f32 {d}, {d}, {b}, {c}", 1),
 ("v_fmac_f32",               "v_fmac_f32 {d}, {b}, {c}", 1),
 ("v_mul_i32_i24",            "v_mul_i32_i24 {d}, {b}, {d}", 1),
 ("v_add_u16",                "v_add_u16 {d}, {b}, {d}", 1),
 ("v_lshlrev_b16",            "v_lshlrev_b16 {d}, {b}, {d}", 1),
 ("v_lshrrev_b16",            "v_lshrrev_b16 {d}, {b}, {d}", 1),
 ("v_writelane_b32",          "v_writelane_b32 {d}, s24, 3", 1),
 ("v_cmpx (exec)",            "v_cmpx_ne_u32 exec, {b}, {c}", 1),
 ("s_waitcnt lgkmcnt(0)",     "s_waitcnt lgkmcnt(0)", 1),
 ("s_cbranch not taken",      "s_cbranch_scc1 1f\n1:", 1),
 ("s_cmp+s_cselect",          "s_cmp_lg_u64 s[20:21], 0\n s_cselect_b64 s[22:23], -1, 0", 2),
 ("v_pk_min_u16",             "v_pk_min_u16 {d}, {d}, {b}", 1),
 ("v_pk_lshlrev_b16",         "v_pk_lshlrev_b16 {d}, {b}, {d}", 1),
 ("v_add_lshl_u32",           "v_add_lshl_u32 {d}, {d}, {b}, 3", 1),
 ("v_mad_u32_u24 (as shl)",   "v_mad_u32_u24 {d}, {d}, 8, {b}", 1),
 ("v_cvt_u32_f32",            "v_cvt_u32_f32 {d}, {d}", 1),
 ("v_ffbl_b32",               "v_ffbl_b32 {d}, {d}", 1),
 ("v_bcnt_u32_b32",           "v_bcnt_u32_b32 {d}, {d}, {b}", 1),
 ("v_lshlrev_b32 by add",     "v_add_u32 {d}, {d}, {d}", 1),
]
out = []
out.append('// GENERATED by valu_rates_gen.py -- development probe (see that file).\n#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n')
OPS = ': "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b), "v"(c), "s"(seed), "v"(addr) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "memory"'
def inst(t, i):
    return t.format(d="%%%d" % i, q="%%%d" % (8 + i % 4), b="%12", c="%13", s="%14", a="%15")
for k, (name, t, n) in enumerate(T):
    ind = "\\n ".join(inst(t, i).replace("\n", "\\n") for i in range(8))
    if "8 then wait" in name: ind += "\\n s_waitcnt lgkmcnt(0)"
    dep = "\\n ".join(inst(t, 0).replace("\n", "\\n") for i in range(8))
    if "8 then wait" in name: dep = "\\n ".join((inst(t, 0) + "\\n s_waitcnt lgkmcnt(0)") for i in range(8))
    out.append(f'''__global__ void k{k}(uint64_t *out, int dep, uint32_t seed)
{{
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x & 1023u] = seed;
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 + 9u, a5 = a0 ^ 77u, a6 = a0 + 1u, a7 = a0 * 11u;
    uint32_t b = seed | 3u, c = (seed >> 1) | 5u, addr = (threadIdx.x * 4u) & 4095u;
    uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = a2 | ((uint64_t)a3 << 32), q2 = a4 | ((uint64_t)a5 << 32), q3 = a6 | ((uint64_t)a7 << 32);
    uint64_t t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (!dep) {{
        for (int i = 0; i < 256; i++) {{
            asm volatile("{ind}\\n {ind}\\n {ind}\\n {ind}\\n {ind}\\n {ind}\\n {ind}\\n {ind}" {OPS});
        }}
    }} else {{
        for (int i = 0; i < 256; i++) {{
            asm volatile("{dep}\\n {dep}\\n {dep}\\n {dep}\\n {dep}\\n {dep}\\n {dep}\\n {dep}" {OPS});
        }}
    }}
    asm volatile("s_memtime %0\\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(q0 + q1 + q2 + q3) + lds[(threadIdx.x + 1) & 1023u] == 0x12345u) out[100] = 1;
}}
''')
out.append('struct Entry { const char *name; void (*fn)(uint64_t *, int, uint32_t); int n; };\nstatic Entry es[] = {\n')
for k, (name, t, n) in enumerate(T):
    out.append(f'    {{ "{name}", k{k}, {n} }},\n')
out.append('''};
int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint64_t *d; hipMalloc(&d, 128 * 8);
    printf("ticks of s_memtime per instruction (a stream of one wavefront alone: 4 shader cycles for the plain ones)\\n");
    printf("%-26s | independent, per instruction and SIMD, 1 / 2 / 4 waves per SIMD | dependent chain, per instruction of a wave, 1 / 2 / 4\\n", "instruction");
    for (auto &e : es) {
        if (argc > 1 && !strstr(e.name, argv[1])) continue;
        printf("%-26s |", e.name);
        for (int dep = 0; dep < 2; dep++) {
            for (int wps = 1; wps <= 4; wps *= 2) {
                const int threads = 256 * wps;
                uint64_t h[16];
                double best = 1e30;
                for (int rep = 0; rep < 3; rep++) {
                    hipLaunchKernelGGL(e.fn, dim3(1), dim3(threads), 0, 0, d, dep, 12345u + rep);
                    if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\\n"); return 1; }
                    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
                    double mx = 0; for (int w = 0; w < threads / 64; w++) mx = h[w] > mx ? (double)h[w] : mx;
                    best = mx < best ? mx : best;
                }
                const double n = 256.0 * 64 * e.n;                   // instructions per wave
                printf(" %7.3f", dep ? best / n : best / (n * wps));
            }
            printf(" |");
        }
        printf("\\n");
    }
    return 0;
}
''')
open(sys.argv[1] if len(sys.argv) > 1 else "valu_rates.hip", "w").write("".join(out).replace("#include <cstdint>\n", "#include <cstdint>\n#include <cstring>\n"))
