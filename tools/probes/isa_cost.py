#!/usr/bin/env python3
"""Development aid: prices the instructions of a piece of gfx950 assembly (stdin, or a file and a line range) with the issue
costs measured by valu_rates (profiles/r05/valu_rates.txt; shader cycles per instruction and SIMD with two or more
wavefronts per SIMD) and prints the sum by class.  usage: isa_cost.py file.s [first_line last_line]"""
import re, sys, collections
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_add_u16", "v_sub_u16", "v_min_u16", "v_max_u16", "v_lshlrev_b16", "v_lshrrev_b16", "v_add_f32", "v_mul_f32",
        "v_fmac_f32", "v_mov_b64"}
def cost(line):
    m = re.match(r"\s*([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|buffer_[a-z0-9_]+|global_[a-z0-9_]+)\s*(.*)", line)
    if not m: return None
    op, rest = m.group(1), m.group(2)
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("s_"):
        if op.startswith("s_waitcnt") or op.startswith("s_nop"): return ("scalar/wait", 0.0)
        return ("scalar", 0.0)       # (issued beside the vector instructions of other waves; not priced here)
    if op.startswith("ds_"): return ("lds", 4.0)
    if op.startswith("buffer_") or op.startswith("global_"): return ("vmem", 4.0)
    sgpr_src = re.search(r"(?<![a-z])s\d+|s\[\d+:\d+\]|vcc|exec", ",".join(rest.split(",")[1:])) is not None
    lit = re.search(r"0x[0-9a-f]{3,}", rest) is not None or re.search(r"(?<![a-z\[:])\b(6[5-9]|[7-9]\d|\d{3,})\b", ",".join(rest.split(",")[1:])) is not None
    if base in FULL and not op.endswith(("_e64", "_sdwa", "_dpp")):
        if sgpr_src: return ("simple op with a scalar source (4)", 4.0)
        if lit: return ("simple op with a literal (2.5)", 2.5)
        return ("simple op (2)", 2.0)
    if base == "v_bitop3_b32" or base == "v_fma_f32": return ("bitop3 (2.5)", 2.5)
    if base.startswith("v_cmp"): return ("compare (4)", 4.0)
    if base == "v_cndmask_b32": return ("select (4)", 4.0)
    if base in ("v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64", "v_mad_u64_u32"): return ("64-bit (4)", 4.0)
    return ("other 4-cycle: " + base, 4.0)
lines = open(sys.argv[1]).read().split("\n") if len(sys.argv) > 1 else sys.stdin.read().split("\n")
if len(sys.argv) > 3: lines = lines[int(sys.argv[2]) - 1:int(sys.argv[3])]
by = collections.Counter(); n = collections.Counter()
for l in lines:
    c = cost(l)
    if c: by[c[0]] += c[1]; n[c[0]] += 1
tot = sum(by.values()); cnt = sum(v for k, v in n.items() if not k.startswith("scalar"))
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"{n[k]:5d} x {k:45s} {v:8.1f} cycles")
print(f"{cnt} vector/memory instructions, {tot:.0f} cycles priced; {sum(v for k, v in n.items() if k.startswith('scalar'))} scalar")
