#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc -S listing with their VALU / SALU / LDS / memory instruction counts
(development aid).  usage: tools/isa_blocks.py file.s KERNEL_SUBSTRING [min_valu]"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
blocks, cur, inside = [], None, False
for i, ln in enumerate(open(path), 1):
    s = ln.strip()
    if not inside:
        if re.match(r'^[_A-Za-z0-9]+:', ln) and kern in ln:
            inside = True
            cur = ['entry', i, 0, 0, 0, 0, []]; blocks.append(cur)
        continue
    if s.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        cur = [m.group(1), i, 0, 0, 0, 0, []]; blocks.append(cur); continue
    if not s or s.startswith(('.', ';')): continue
    op = s.split()[0]
    if op.startswith('v_'): cur[2] += 1
    elif op.startswith('s_'): cur[3] += 1
    elif op.startswith('ds_'): cur[4] += 1
    else: cur[5] += 1
    if op.startswith('s_cbranch') or op == 's_branch': cur[6].append(s.split()[-1])
tot = [0, 0, 0, 0]
for b in blocks:
    for k in range(4): tot[k] += b[2 + k]
    if b[2] >= minv:
        print(f"{b[0]:12} line {b[1]:6} valu {b[2]:4} salu {b[3]:4} lds {b[4]:3} mem {b[5]:2} -> {' '.join(b[6])}")
print("total valu %d salu %d lds %d mem %d" % tuple(tot))
