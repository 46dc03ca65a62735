// Development probe: SIMD time per wave-instruction for single opcodes on gfx950, at 1, 2, 4, 5, 6
// and 8 waves per SIMD (all 256 CUs busy).  Each wave runs iters x 64 copies of ONE opcode on 8
// independent register chains between s_memtime stamps; reported: ns and shader cycles of SIMD
// time per wave-instruction = (last end - first start on the SIMD) / (waves x instructions).
// The compress kernel is bound by VALU issue (profiles/r02/README.md): this is the price list its
// instruction selection is made from.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define REP8(x) x x x x x x x x
struct Rec { unsigned long long t0, t1, r0, r1; uint32_t hw, xcc; };

// OP(fmt): fmt uses %0..%7 as the 8 chains (read-write), %8 a loop-invariant VGPR, %9 an LDS address VGPR
#define OPS(X) \
    X(0,  "v_xor_b32 %0, %0, %8",                  "v_xor_b32 %1, %1, %8", "v_xor_b32 %2, %2, %8", "v_xor_b32 %3, %3, %8", "v_xor_b32 %4, %4, %8", "v_xor_b32 %5, %5, %8", "v_xor_b32 %6, %6, %8", "v_xor_b32 %7, %7, %8") \
    X(1,  "v_add_u32 %0, %0, %8",                  "v_add_u32 %1, %1, %8", "v_add_u32 %2, %2, %8", "v_add_u32 %3, %3, %8", "v_add_u32 %4, %4, %8", "v_add_u32 %5, %5, %8", "v_add_u32 %6, %6, %8", "v_add_u32 %7, %7, %8") \
    X(2,  "v_mov_b32 %0, %8",                      "v_mov_b32 %1, %8", "v_mov_b32 %2, %8", "v_mov_b32 %3, %8", "v_mov_b32 %4, %8", "v_mov_b32 %5, %8", "v_mov_b32 %6, %8", "v_mov_b32 %7, %8") \
    X(3,  "v_lshlrev_b32 %0, 3, %0",               "v_lshlrev_b32 %1, 3, %1", "v_lshlrev_b32 %2, 3, %2", "v_lshlrev_b32 %3, 3, %3", "v_lshlrev_b32 %4, 3, %4", "v_lshlrev_b32 %5, 3, %5", "v_lshlrev_b32 %6, 3, %6", "v_lshlrev_b32 %7, 3, %7") \
    X(4,  "v_ffbl_b32 %0, %0",                     "v_ffbl_b32 %1, %1", "v_ffbl_b32 %2, %2", "v_ffbl_b32 %3, %3", "v_ffbl_b32 %4, %4", "v_ffbl_b32 %5, %5", "v_ffbl_b32 %6, %6", "v_ffbl_b32 %7, %7") \
    X(5,  "v_alignbyte_b32 %0, %0, %8, 1",         "v_alignbyte_b32 %1, %1, %8, 1", "v_alignbyte_b32 %2, %2, %8, 1", "v_alignbyte_b32 %3, %3, %8, 1", "v_alignbyte_b32 %4, %4, %8, 1", "v_alignbyte_b32 %5, %5, %8, 1", "v_alignbyte_b32 %6, %6, %8, 1", "v_alignbyte_b32 %7, %7, %8, 1") \
    X(6,  "v_alignbyte_b32 %0, %0, %8, %1",        "v_alignbyte_b32 %1, %1, %8, %2", "v_alignbyte_b32 %2, %2, %8, %3", "v_alignbyte_b32 %3, %3, %8, %4", "v_alignbyte_b32 %4, %4, %8, %5", "v_alignbyte_b32 %5, %5, %8, %6", "v_alignbyte_b32 %6, %6, %8, %7", "v_alignbyte_b32 %7, %7, %8, %0") \
    X(7,  "v_min3_u32 %0, %0, %8, %1",             "v_min3_u32 %1, %1, %8, %2", "v_min3_u32 %2, %2, %8, %3", "v_min3_u32 %3, %3, %8, %4", "v_min3_u32 %4, %4, %8, %5", "v_min3_u32 %5, %5, %8, %6", "v_min3_u32 %6, %6, %8, %7", "v_min3_u32 %7, %7, %8, %0") \
    X(8,  "v_min_u32 %0, %0, %8",                  "v_min_u32 %1, %1, %8", "v_min_u32 %2, %2, %8", "v_min_u32 %3, %3, %8", "v_min_u32 %4, %4, %8", "v_min_u32 %5, %5, %8", "v_min_u32 %6, %6, %8", "v_min_u32 %7, %7, %8") \
    X(9,  "v_max_i32 %0, %0, %8",                  "v_max_i32 %1, %1, %8", "v_max_i32 %2, %2, %8", "v_max_i32 %3, %3, %8", "v_max_i32 %4, %4, %8", "v_max_i32 %5, %5, %8", "v_max_i32 %6, %6, %8", "v_max_i32 %7, %7, %8") \
    X(10, "v_cndmask_b32 %0, %0, %8, vcc",         "v_cndmask_b32 %1, %1, %8, vcc", "v_cndmask_b32 %2, %2, %8, vcc", "v_cndmask_b32 %3, %3, %8, vcc", "v_cndmask_b32 %4, %4, %8, vcc", "v_cndmask_b32 %5, %5, %8, vcc", "v_cndmask_b32 %6, %6, %8, vcc", "v_cndmask_b32 %7, %7, %8, vcc") \
    X(11, "v_cmp_lt_u32 vcc, %0, %8",              "v_cmp_lt_u32 vcc, %1, %8", "v_cmp_lt_u32 vcc, %2, %8", "v_cmp_lt_u32 vcc, %3, %8", "v_cmp_lt_u32 vcc, %4, %8", "v_cmp_lt_u32 vcc, %5, %8", "v_cmp_lt_u32 vcc, %6, %8", "v_cmp_lt_u32 vcc, %7, %8") \
    X(12, "v_cmp_lt_u32 s[20:21], %0, %8",         "v_cmp_lt_u32 s[22:23], %1, %8", "v_cmp_lt_u32 s[24:25], %2, %8", "v_cmp_lt_u32 s[26:27], %3, %8", "v_cmp_lt_u32 s[20:21], %4, %8", "v_cmp_lt_u32 s[22:23], %5, %8", "v_cmp_lt_u32 s[24:25], %6, %8", "v_cmp_lt_u32 s[26:27], %7, %8") \
    X(13, "v_bfe_u32 %0, %0, 3, 11",               "v_bfe_u32 %1, %1, 3, 11", "v_bfe_u32 %2, %2, 3, 11", "v_bfe_u32 %3, %3, 3, 11", "v_bfe_u32 %4, %4, 3, 11", "v_bfe_u32 %5, %5, 3, 11", "v_bfe_u32 %6, %6, 3, 11", "v_bfe_u32 %7, %7, 3, 11") \
    X(14, "v_perm_b32 %0, %0, %8, %1",             "v_perm_b32 %1, %1, %8, %2", "v_perm_b32 %2, %2, %8, %3", "v_perm_b32 %3, %3, %8, %4", "v_perm_b32 %4, %4, %8, %5", "v_perm_b32 %5, %5, %8, %6", "v_perm_b32 %6, %6, %8, %7", "v_perm_b32 %7, %7, %8, %0") \
    X(15, "v_lshl_or_b32 %0, %0, 3, %8",           "v_lshl_or_b32 %1, %1, 3, %8", "v_lshl_or_b32 %2, %2, 3, %8", "v_lshl_or_b32 %3, %3, 3, %8", "v_lshl_or_b32 %4, %4, 3, %8", "v_lshl_or_b32 %5, %5, 3, %8", "v_lshl_or_b32 %6, %6, 3, %8", "v_lshl_or_b32 %7, %7, 3, %8") \
    X(16, "v_add3_u32 %0, %0, %8, %1",             "v_add3_u32 %1, %1, %8, %2", "v_add3_u32 %2, %2, %8, %3", "v_add3_u32 %3, %3, %8, %4", "v_add3_u32 %4, %4, %8, %5", "v_add3_u32 %5, %5, %8, %6", "v_add3_u32 %6, %6, %8, %7", "v_add3_u32 %7, %7, %8, %0") \
    X(17, "v_mul_u32_u24 %0, %0, %8",              "v_mul_u32_u24 %1, %1, %8", "v_mul_u32_u24 %2, %2, %8", "v_mul_u32_u24 %3, %3, %8", "v_mul_u32_u24 %4, %4, %8", "v_mul_u32_u24 %5, %5, %8", "v_mul_u32_u24 %6, %6, %8", "v_mul_u32_u24 %7, %7, %8") \
    X(18, "v_mul_lo_u32 %0, %0, %8",               "v_mul_lo_u32 %1, %1, %8", "v_mul_lo_u32 %2, %2, %8", "v_mul_lo_u32 %3, %3, %8", "v_mul_lo_u32 %4, %4, %8", "v_mul_lo_u32 %5, %5, %8", "v_mul_lo_u32 %6, %6, %8", "v_mul_lo_u32 %7, %7, %8") \
    X(19, "v_mul_hi_u32 %0, %0, %8",               "v_mul_hi_u32 %1, %1, %8", "v_mul_hi_u32 %2, %2, %8", "v_mul_hi_u32 %3, %3, %8", "v_mul_hi_u32 %4, %4, %8", "v_mul_hi_u32 %5, %5, %8", "v_mul_hi_u32 %6, %6, %8", "v_mul_hi_u32 %7, %7, %8") \
    X(20, "v_mad_u32_u24 %0, %0, %8, %1",          "v_mad_u32_u24 %1, %1, %8, %2", "v_mad_u32_u24 %2, %2, %8, %3", "v_mad_u32_u24 %3, %3, %8, %4", "v_mad_u32_u24 %4, %4, %8, %5", "v_mad_u32_u24 %5, %5, %8, %6", "v_mad_u32_u24 %6, %6, %8, %7", "v_mad_u32_u24 %7, %7, %8, %0") \
    X(21, "v_and_or_b32 %0, %0, %8, %1",           "v_and_or_b32 %1, %1, %8, %2", "v_and_or_b32 %2, %2, %8, %3", "v_and_or_b32 %3, %3, %8, %4", "v_and_or_b32 %4, %4, %8, %5", "v_and_or_b32 %5, %5, %8, %6", "v_and_or_b32 %6, %6, %8, %7", "v_and_or_b32 %7, %7, %8, %0") \
    X(22, "v_xor_b32_e64 %0, %0, %8",              "v_xor_b32_e64 %1, %1, %8", "v_xor_b32_e64 %2, %2, %8", "v_xor_b32_e64 %3, %3, %8", "v_xor_b32_e64 %4, %4, %8", "v_xor_b32_e64 %5, %5, %8", "v_xor_b32_e64 %6, %6, %8", "v_xor_b32_e64 %7, %7, %8") \
    X(23, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(24, "v_readfirstlane_b32 s20, %0",           "v_readfirstlane_b32 s21, %1", "v_readfirstlane_b32 s22, %2", "v_readfirstlane_b32 s23, %3", "v_readfirstlane_b32 s24, %4", "v_readfirstlane_b32 s25, %5", "v_readfirstlane_b32 s26, %6", "v_readfirstlane_b32 s27, %7") \
    X(25, "v_lshlrev_b64 v[100:101], 3, v[100:101]", "v_lshlrev_b64 v[102:103], 5, v[102:103]", "v_lshlrev_b64 v[104:105], 3, v[104:105]", "v_lshlrev_b64 v[106:107], 5, v[106:107]", "v_lshlrev_b64 v[100:101], 3, v[100:101]", "v_lshlrev_b64 v[102:103], 5, v[102:103]", "v_lshlrev_b64 v[104:105], 3, v[104:105]", "v_lshlrev_b64 v[106:107], 5, v[106:107]") \
    X(26, "v_sad_u8 %0, %0, %8, %1",               "v_sad_u8 %1, %1, %8, %2", "v_sad_u8 %2, %2, %8, %3", "v_sad_u8 %3, %3, %8, %4", "v_sad_u8 %4, %4, %8, %5", "v_sad_u8 %5, %5, %8, %6", "v_sad_u8 %6, %6, %8, %7", "v_sad_u8 %7, %7, %8, %0") \
    X(27, "v_bfi_b32 %0, %0, %8, %1",              "v_bfi_b32 %1, %1, %8, %2", "v_bfi_b32 %2, %2, %8, %3", "v_bfi_b32 %3, %3, %8, %4", "v_bfi_b32 %4, %4, %8, %5", "v_bfi_b32 %5, %5, %8, %6", "v_bfi_b32 %6, %6, %8, %7", "v_bfi_b32 %7, %7, %8, %0") \
    X(28, "v_xor_b32 %0, %1, %2",                  "v_xor_b32 %1, %2, %3", "v_xor_b32 %2, %3, %4", "v_xor_b32 %3, %4, %5", "v_xor_b32 %4, %5, %6", "v_xor_b32 %5, %6, %7", "v_xor_b32 %6, %7, %0", "v_xor_b32 %7, %0, %1") \
    X(29, "ds_read_b32 %0, %9",                    "ds_read_b32 %1, %9", "ds_read_b32 %2, %9", "ds_read_b32 %3, %9", "ds_read_b32 %4, %9", "ds_read_b32 %5, %9", "ds_read_b32 %6, %9", "ds_read_b32 %7, %9\ns_waitcnt lgkmcnt(0)") \
    X(30, "ds_read_b64 v[100:101], %9", "ds_read_b64 v[102:103], %9", "ds_read_b64 v[104:105], %9", "ds_read_b64 v[106:107], %9", "ds_read_b64 v[100:101], %9", "ds_read_b64 v[102:103], %9", "ds_read_b64 v[104:105], %9", "ds_read_b64 v[106:107], %9\ns_waitcnt lgkmcnt(0)") \
    X(31, "ds_read_b128 v[100:103], %9", "ds_read_b128 v[104:107], %9", "ds_read_b128 v[100:103], %9", "ds_read_b128 v[104:107], %9", "ds_read_b128 v[100:103], %9", "ds_read_b128 v[104:107], %9", "ds_read_b128 v[100:103], %9", "ds_read_b128 v[104:107], %9\ns_waitcnt lgkmcnt(0)") \
    X(32, "ds_read2_b32 v[100:101], %9 offset1:1", "ds_read2_b32 v[102:103], %9 offset1:1", "ds_read2_b32 v[104:105], %9 offset1:1", "ds_read2_b32 v[106:107], %9 offset1:1", "ds_read2_b32 v[100:101], %9 offset1:1", "ds_read2_b32 v[102:103], %9 offset1:1", "ds_read2_b32 v[104:105], %9 offset1:1", "ds_read2_b32 v[106:107], %9 offset1:1\ns_waitcnt lgkmcnt(0)") \
    X(33, "ds_write_b32 %9, %0",                   "ds_write_b32 %9, %1", "ds_write_b32 %9, %2", "ds_write_b32 %9, %3", "ds_write_b32 %9, %4", "ds_write_b32 %9, %5", "ds_write_b32 %9, %6", "ds_write_b32 %9, %7\ns_waitcnt lgkmcnt(0)") \
    X(34, "ds_bpermute_b32 %0, %9, %0",            "ds_bpermute_b32 %1, %9, %1", "ds_bpermute_b32 %2, %9, %2", "ds_bpermute_b32 %3, %9, %3", "ds_bpermute_b32 %4, %9, %4", "ds_bpermute_b32 %5, %9, %5", "ds_bpermute_b32 %6, %9, %6", "ds_bpermute_b32 %7, %9, %7\ns_waitcnt lgkmcnt(0)") \
    X(35, "ds_read_u8 %0, %9",                     "ds_read_u8 %1, %9", "ds_read_u8 %2, %9", "ds_read_u8 %3, %9", "ds_read_u8 %4, %9", "ds_read_u8 %5, %9", "ds_read_u8 %6, %9", "ds_read_u8 %7, %9\ns_waitcnt lgkmcnt(0)") \
    X(36, "ds_read_u16 %0, %9",                    "ds_read_u16 %1, %9", "ds_read_u16 %2, %9", "ds_read_u16 %3, %9", "ds_read_u16 %4, %9", "ds_read_u16 %5, %9", "ds_read_u16 %6, %9", "ds_read_u16 %7, %9\ns_waitcnt lgkmcnt(0)") \
    X(37, "s_add_u32 s20, s20, 7",                 "s_add_u32 s21, s21, 7", "s_add_u32 s22, s22, 7", "s_add_u32 s23, s23, 7", "s_add_u32 s24, s24, 7", "s_add_u32 s25, s25, 7", "s_add_u32 s26, s26, 7", "s_add_u32 s27, s27, 7") \
    X(38, "v_xor_b32 %0, %0, %8\ns_add_u32 s20, s20, 7", "v_xor_b32 %1, %1, %8\ns_add_u32 s21, s21, 7", "v_xor_b32 %2, %2, %8\ns_add_u32 s22, s22, 7", "v_xor_b32 %3, %3, %8\ns_add_u32 s23, s23, 7", "v_xor_b32 %4, %4, %8\ns_add_u32 s24, s24, 7", "v_xor_b32 %5, %5, %8\ns_add_u32 s25, s25, 7", "v_xor_b32 %6, %6, %8\ns_add_u32 s26, s26, 7", "v_xor_b32 %7, %7, %8\ns_add_u32 s27, s27, 7") \
    X(39, "v_xor_b32 %0, %0, %8\nds_read_b32 %1, %9", "v_xor_b32 %2, %2, %8\nds_read_b32 %3, %9", "v_xor_b32 %4, %4, %8\nds_read_b32 %5, %9", "v_xor_b32 %6, %6, %8\nds_read_b32 %7, %9\ns_waitcnt lgkmcnt(0)", "v_xor_b32 %0, %0, %8\nds_read_b32 %1, %9", "v_xor_b32 %2, %2, %8\nds_read_b32 %3, %9", "v_xor_b32 %4, %4, %8\nds_read_b32 %5, %9", "v_xor_b32 %6, %6, %8\nds_read_b32 %7, %9\ns_waitcnt lgkmcnt(0)") \
    X(40, "v_sub_u32 %0, %0, %8",                  "v_subrev_u32 %1, %1, %8", "v_or_b32 %2, %2, %8", "v_and_b32 %3, %3, %8", "v_lshrrev_b32 %4, 1, %4", "v_ashrrev_i32 %5, 1, %5", "v_max_u32 %6, %6, %8", "v_min_i32 %7, %7, %8") \
    X(41, "v_pk_add_u16 %0, %0, %8",               "v_pk_add_u16 %1, %1, %8", "v_pk_add_u16 %2, %2, %8", "v_pk_add_u16 %3, %3, %8", "v_pk_add_u16 %4, %4, %8", "v_pk_add_u16 %5, %5, %8", "v_pk_add_u16 %6, %6, %8", "v_pk_add_u16 %7, %7, %8") \
    X(42, "v_cmp_eq_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc", "v_cmp_eq_u32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc", "v_cmp_eq_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc", "v_cmp_eq_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc", "v_cmp_eq_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc", "v_cmp_eq_u32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc", "v_cmp_eq_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc", "v_cmp_eq_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc") \
    X(43, "v_cmp_eq_u32 s[20:21], %0, %8\nv_cndmask_b32 %1, %1, %8, s[24:25]", "v_cmp_eq_u32 s[22:23], %2, %8\nv_cndmask_b32 %3, %3, %8, s[26:27]", "v_cmp_eq_u32 s[24:25], %4, %8\nv_cndmask_b32 %5, %5, %8, s[20:21]", "v_cmp_eq_u32 s[26:27], %6, %8\nv_cndmask_b32 %7, %7, %8, s[22:23]", "v_cmp_eq_u32 s[20:21], %0, %8\nv_cndmask_b32 %1, %1, %8, s[24:25]", "v_cmp_eq_u32 s[22:23], %2, %8\nv_cndmask_b32 %3, %3, %8, s[26:27]", "v_cmp_eq_u32 s[24:25], %4, %8\nv_cndmask_b32 %5, %5, %8, s[20:21]", "v_cmp_eq_u32 s[26:27], %6, %8\nv_cndmask_b32 %7, %7, %8, s[22:23]")

#define NOPS 44
static const int kPerBody[NOPS] = {8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,16,16,8,8,16,16};

__global__ __launch_bounds__(256) void rate(Rec *out, int iters, int op)
{
    extern __shared__ uint32_t lds[];
    // 64-bit pairs for the b64 forms: chains live in even-aligned pairs where needed
    uint64_t p0 = threadIdx.x, p1 = threadIdx.x * 3 + 1, p2 = threadIdx.x ^ 0x55, p3 = threadIdx.x + 7;
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5, f = ~a, g = a << 3, h = a + 99;
    const uint32_t k = blockIdx.x + 12345u;
    for (uint32_t i = threadIdx.x; i < 2048; i += 256) lds[i] = i;
    __syncthreads();
    uint32_t addr = (threadIdx.x * 4u) & 1023u;          // conflict-free, one dword per lane
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) {
        switch (op) {
#define X(id, i0, i1, i2, i3, i4, i5, i6, i7) \
        case id: asm volatile(REP8(i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\n") \
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k), "v"(addr) \
                              : "vcc", "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107"); break;
#define Y(id, i0, i1, i2, i3, i4, i5, i6, i7) \
        case id: asm volatile(REP8(i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\n") \
                              : "+v"(p0), "+v"(a), "+v"(p1), "+v"(b), "+v"(p2), "+v"(c), "+v"(p3), "+v"(d) : "v"(k), "v"(addr) \
                              : "vcc", "memory", "scc"); break;
        OPS(X)
#undef X
        default: break;
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
    if (a + b + c + d + e + f + g + h + (uint32_t)(p0 + p1 + p2 + p3) == 0x12345678u) out[0].hw = a;
}

static const char *kNames[NOPS] = {
    "v_xor_b32 (VOP2)", "v_add_u32", "v_mov_b32", "v_lshlrev_b32 imm", "v_ffbl_b32", "v_alignbyte imm", "v_alignbyte vgpr", "v_min3_u32",
    "v_min_u32", "v_max_i32", "v_cndmask vcc", "v_cmp -> vcc", "v_cmp -> sgpr pair", "v_bfe_u32", "v_perm_b32", "v_lshl_or_b32",
    "v_add3_u32", "v_mul_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_and_or_b32", "v_xor_b32 (VOP3)", "v_mov_b32 dpp row_shr",
    "v_readfirstlane", "v_lshlrev_b64", "v_sad_u8", "v_bfi_b32", "v_xor 2 distinct srcs", "ds_read_b32", "ds_read_b64", "ds_read_b128",
    "ds_read2_b32", "ds_write_b32", "ds_bpermute_b32", "ds_read_u8", "ds_read_u16", "s_add_u32", "v_xor + s_add (per pair)", "v_xor + ds_read_b32 (per pair)",
    "VOP2 int mix (sub/or/and/shr/max)", "v_pk_add_u16", "v_cmp vcc + v_cndmask (per pair)", "v_cmp sgpr + v_cndmask (per pair)"};

int main(int argc, char **argv)
{
    const int iters = 1000;
    Rec *d;
    hipMalloc(&d, sizeof(Rec) * 256 * 8 * 4);
    hipFuncSetAttribute((const void *)rate, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int ws[] = {1, 2, 4, 5, 6, 8};
    printf("SIMD cycles per wave-instruction (ns in brackets) by waves per SIMD; all 256 CUs busy\n%-36s", "opcode");
    for (int w : ws) printf("      w=%d      ", w);
    printf("\n");
    for (int op = 0; op < NOPS; op++) {
        printf("%-36s", kNames[op]);
        for (int w : ws) {
            const size_t lds = (160 * 1024 / w) & ~255u;
            const int nblk = 256 * w;
            std::vector<Rec> h(nblk * 4);
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(rate, dim3(nblk), dim3(256), lds, 0, d, iters, op);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, sizeof(Rec) * nblk * 4, hipMemcpyDeviceToHost);
            std::map<uint64_t, std::vector<Rec>> by;
            for (auto &r : h) by[((uint64_t)(r.xcc & 15u) << 32) | (r.hw & 0xFF30u)].push_back(r);
            std::vector<double> cpi, ghz;
            for (auto &r : h) ghz.push_back((double)(r.t1 - r.t0) / (double)(r.r1 - r.r0) * 0.1);
            std::sort(ghz.begin(), ghz.end());
            for (auto &kv : by) {
                unsigned long long a = ~0ull, b = 0;
                for (auto &r : kv.second) { a = std::min(a, r.t0); b = std::max(b, r.t1); }
                cpi.push_back((double)(b - a) / ((double)kv.second.size() * iters * 8 * kPerBody[op]));
            }
            std::sort(cpi.begin(), cpi.end());
            const double cyc = cpi[cpi.size() / 2], f = ghz[ghz.size() / 2];
            printf(" %6.2f (%5.2f)", cyc, cyc / f);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
