#!/usr/bin/env python3
"""Static instruction counts of one kernel by source line (development aid).

usage: tools/isa_lines.py file.s KERNEL_SUBSTRING [lo-hi:label ...]
`file.s` comes from `hipcc ... -gline-tables-only -S --cuda-device-only lzs_kernels.hip`.  Prints, per source
line of kernels/compress_wg.inc and kernels/common.inc, how many VALU / SALU / LDS / VMEM instructions the
compiler emitted for it inside the named kernel, and sums over the given line ranges of compress_wg.inc.
"""
import re, sys, collections

def main():
    path, kern = sys.argv[1], sys.argv[2]
    ranges = []
    for a in sys.argv[3:]:
        r, label = a.split(":")
        lo, hi = r.split("-")
        ranges.append((int(lo), int(hi), label))
    files = {}
    cur = None
    inside = False
    per = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
    for ln in open(path):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if re.match(r'^[_A-Za-z0-9]+:', ln) and kern in ln and not ln.startswith('.'):
            inside = True
            continue
        if not inside:
            continue
        if s.startswith('.Lfunc_end'):
            break
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(('.', ';')) or s.endswith(':'):
            continue
        op = s.split()[0]
        if op.startswith('v_'):
            k = 0
        elif op.startswith('s_'):
            k = 1
        elif op.startswith('ds_'):
            k = 2
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            k = 3
        else:
            k = 4
        per[cur][k] += 1
    tot = [0] * 5
    for k in sorted(per, key=lambda x: (str(x[0]), x[1]) if x else ("", 0)):
        v = per[k]
        for i in range(5):
            tot[i] += v[i]
        print(f"{str(k[0]) if k else '?':>22}:{k[1] if k else 0:<5} valu {v[0]:4} salu {v[1]:4} lds {v[2]:3} vmem {v[3]:3}")
    print("total", tot)
    for lo, hi, label in ranges:
        t = [0] * 5
        for k, v in per.items():
            if k and k[0] == 'compress_wg.inc' and lo <= k[1] <= hi:
                for i in range(5):
                    t[i] += v[i]
        print(f"{label:>20} [{lo}-{hi}]: valu {t[0]} salu {t[1]} lds {t[2]} vmem {t[3]}")

main()
