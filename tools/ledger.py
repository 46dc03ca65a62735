#!/usr/bin/env python3
"""Instruction ledger of one kernel: static ISA counts per basic block x trip counts -> instructions per input byte by phase,
to be reconciled with SQ_INSTS_VALU / SALU / LDS (profiles/r06/ledger_text.txt).

usage: tools/ledger.py file.s KERNEL_SUBSTRING [trips.json]
  file.s      from tools/isa.sh (hipcc -S -gline-tables-only of lzs_kernels.hip)
  trips.json  {"phases": [[first_line, last_line, "phase", "trip expression"], ...], "ranges": [...], "vars": {...}, "per_pool_bytes": 512}
              a block belongs to the phase whose line range of kernels/compress_wg.inc holds most of its instructions' .loc
              lines (helpers -- lines 131-220 and common.inc -- vote with the block's other lines); the trip expression is
              evaluated over "vars" (measured counts per wave and pool: tools/probes/prof_compress) and says how often a
              wave runs the block per pool.  Without trips.json the blocks are listed with their line spans.
"""
import collections
import json
import re
import sys

HELPER = (131, 220)


def parse(path, kern):
    files, blocks, cur, inside, loc = {}, [], None, False, None
    for i, ln in enumerate(open(path), 1):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if re.match(r'^[_A-Za-z0-9]+:', ln) and kern in ln:
                inside = True
                cur = dict(label="entry", at=i, n=[0, 0, 0, 0, 0], lines=collections.Counter(), note="", to=[])
                blocks.append(cur)
            continue
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", s)
        if m:
            cur = dict(label=m.group(1), at=i, n=[0, 0, 0, 0, 0], lines=collections.Counter(), note=(m.group(2) or "").strip("; "), to=[])
            blocks.append(cur)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3
        if op in ("s_waitcnt", "s_nop", "s_barrier", "s_branch", "s_setprio", "s_endpgm", "s_sleep") or op.startswith("s_cbranch"):
            k = 4                                          # issue slots that are no scalar ALU work
        cur["n"][k] += 1
        if loc:
            cur["lines"][loc] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["to"].append(s.split()[-1])
            if op.startswith("s_cbranch"):                 # what follows a conditional branch is a block of its own
                cur = dict(label=cur["label"] + "+", at=i + 1, n=[0, 0, 0, 0, 0], lines=collections.Counter(), note="(falls through)", to=[])
                blocks.append(cur)
    return blocks


def own_lines(b):
    """the block's .loc lines in compress_wg.inc outside the helpers"""
    c = collections.Counter()
    for (f, l), n in b["lines"].items():
        if f.endswith("compress_wg.inc") and not (HELPER[0] <= l <= HELPER[1]):
            c[l] += n
    return c


def main():
    blocks = parse(sys.argv[1], sys.argv[2])
    cfg = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
    if not cfg:
        for b in blocks:
            c = own_lines(b)
            span = f"{min(c)}-{max(c)}" if c else "-"
            top = ",".join(str(l) for l, _ in c.most_common(3))
            print(f"{b['label']:10} @{b['at']:5} valu {b['n'][0]:4} salu {b['n'][1]:4} lds {b['n'][2]:3} vmem {b['n'][3]:2}  lines {span:11} top {top:16} {b['note'][:40]} -> {' '.join(b['to'])}")
        t = [sum(b["n"][k] for b in blocks) for k in range(4)]
        print("static total: valu %d salu %d lds %d vmem %d" % tuple(t))
        return
    env = dict(cfg["vars"])
    phases = cfg.get("phases", [])
    agg = collections.OrderedDict()
    rows = []
    prev = None
    for b in blocks:
        c = own_lines(b)
        votes = collections.Counter()
        for l, n in c.items():
            for lo, hi, name, _ in phases:
                if lo <= l <= hi:
                    votes[name] += n
                    break
        name = votes.most_common(1)[0][0] if votes else prev          # a block of helper lines only: as the block before it
        expr = next((e for lo, hi, nm, e in phases if nm == name), "0")
        # "ranges": [[first listing line, last listing line, phase, trips], ...] -- blocks by where they stand in THIS listing
        # (the first range that holds the block's first line wins; what the ledger of one compile is made of)
        for lo, hi, nm, e in cfg.get("ranges", []):
            if lo <= b["at"] <= hi:
                name, expr = nm, e
                break
        prev = name
        trips = float(eval(expr, {}, env)) if name else 0.0
        rows.append((b, name, trips))
        a = agg.setdefault(name, [0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0.0])
        a[7] += trips * b["n"][4]
        for k in range(4):
            a[k] += trips * b["n"][k]
        a[4] += b["n"][0]; a[5] += b["n"][1]; a[6] += b["n"][2]
    per = cfg.get("per_pool_bytes", 512) / 4.0            # input bytes per wave and pool
    print(f"# blocks: label, listing line, static VALU/SALU/LDS/VMEM, phase, trips per wave and pool")
    for b, name, trips in rows:
        if sum(b["n"]):
            print(f"  {b['label']:10} @{b['at']:5}  {b['n'][0]:4} {b['n'][1]:4} {b['n'][2]:3} {b['n'][3]:2}  {name or '?':22} x {trips:6.3f}")
    print(f"# per phase: static VALU / SALU / LDS of its blocks; dynamic per wave and pool; per input byte (a wave's share of a pool: {per:.0f} bytes)")
    tot = [0.0] * 4
    for name, a in agg.items():
        for k in range(4):
            tot[k] += a[k]
        print(f"  {name or '?':22} static {a[4]:4d} {a[5]:4d} {a[6]:3d} | per wave and pool valu {a[0]:7.1f} salu {a[1]:7.1f} lds {a[2]:6.1f}  wait/branch {a[7]:6.1f} | per byte valu {a[0] / per:6.3f} salu {a[1] / per:6.3f} lds {a[2] / per:6.3f}")
    print(f"  {'TOTAL':22}                      | per wave and pool valu {tot[0]:7.1f} salu {tot[1]:7.1f} lds {tot[2]:6.1f} vmem {tot[3]:5.2f} | per byte valu {tot[0] / per:6.3f} salu {tot[1] / per:6.3f} lds {tot[2] / per:6.3f}")
    if "measured" in cfg:
        m = cfg["measured"]
        print(f"# measured (rocprofv3 --pmc, {m.get('source', '')}): SQ_INSTS_VALU {m['valu']:.3f}, SQ_INSTS_SALU {m['salu']:.3f}, SQ_INSTS_LDS {m['lds']:.3f} per input byte")
        print(f"# ledger / measured: valu {tot[0] / per / m['valu']:.3f}, salu {tot[1] / per / m['salu']:.3f}, lds {tot[2] / per / m['lds']:.3f}")


main()
