#!/usr/bin/env python3
"""Turns the text output of tools/gpu_pmc.sh (one file per class: <prefix>_text.txt, <prefix>_lowent.txt,
<prefix>_random.txt -- SQ passes, FETCH_SIZE + GRBM_GUI_ACTIVE, WRITE_SIZE, each its own rocprofv3 --pmc run
with the kernel trace only) into

  profiles/pmc_limiter.json  what binds lzs_compress_blocks_wg_kernel per class: vector instructions per input
                             byte, vector-issue and LDS busy fractions, bank-conflict share, waiting share
  profiles/pmc_traffic.json  HBM bytes per launch (FETCH_SIZE doubled per MI355X_MICROARCH.md for gfx950)

both stamped with the identity of the kernel sources in THIS tree (bench.py reports them only while it matches).
usage: tools/pmc_limiter.py gpurun_out/pmc4a   [outdir = profiles]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_identity  # noqa: E402

INPUT_BYTES = 16384 * 65536
SIMDS, CUS, XCDS = 1024, 256, 8
HOW = ("rocprofv3 --kernel-trace --pmc, one pass per counter set, bench.py --steps 1 --workload <class> (16384 x 64 KiB per launch), "
       "per-dispatch means. kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs. valu_issue_busy = SQ_ACTIVE_INST_VALU x 4 cycles / "
       "(kernel_cycles x 1024 SIMDs): an upper bound, plain integer VOP2 opcodes measured at 2.2-2.3 cycles (profiles/r02/op_cost_probe.txt). "
       "lds_busy = SQ_LDS_IDX_ACTIVE / (kernel_cycles x 256 CUs). waves_waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.")
NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace only); counters are in KB; "
        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests of wide streaming reads at 64 B)")
# what the counters are the counters OF: tools/gpu_pmc.sh forces the class's own variant of the kernel on every block
# (LZS_VARIANT), so a launch of the library is ONE dispatch of that variant -- the counters are that dispatch's.  A caller's launch
# also runs lzs_classify_blocks_kernel (two 16-byte loads a lane over a block's first 2 KiB: ~32 MiB read per GiB, 3 % on top of
# FETCH) and the other variants' grids, whose workgroups return after one scalar load; bench.py's events time all of them.
VARIANT = {"text": "text", "lowent": "few", "random": "lit"}
SCOPE = ("one dispatch of lzs_compress_blocks_wg_kernel, the class's own variant forced on every block (LZS_VARIANT); not in the "
         "counters: lzs_classify_blocks_kernel (reads the first 2 KiB of every block) and the other variants' empty grids, which a "
         "caller's launch also runs and bench.py's HIP events include")


def read(path):
    vals = {}
    for ln in open(path):
        m = re.match(r"\s*([A-Z_]+): per-dispatch mean ([0-9.e+\-]+) over (\d+) dispatches", ln)
        if m:
            vals[m.group(1)] = float(m.group(2))
    return vals


def main(prefix, outdir):
    ident = kernel_identity()
    lim = {"kernel_source_sha256": ident, "how": HOW}
    tra = {"kernel_source_sha256": ident, "git_head": "(see the commit that adds this file)"}
    for cls in ("text", "lowent", "random"):
        v = read(f"{prefix}_{cls}.txt")
        cyc = v["GRBM_GUI_ACTIVE"] / XCDS
        valu_busy = v["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * SIMDS)
        lds_busy = v["SQ_LDS_IDX_ACTIVE"] / (cyc * CUS)
        lim[cls] = {
            "kernel": "lzs_compress_blocks_wg_kernel", "launch": "16384 blocks x 64 KiB", "variant_forced": VARIANT[cls], "scope": SCOPE,
            "valu_insts_per_input_byte": v["SQ_INSTS_VALU"] / INPUT_BYTES,
            "salu_insts_per_input_byte": v["SQ_INSTS_SALU"] / INPUT_BYTES,
            "lds_insts_per_input_byte": v["SQ_INSTS_LDS"] / INPUT_BYTES,
            "kernel_cycles": cyc,
            "valu_issue_busy": valu_busy, "lds_busy": lds_busy,
            "lds_bank_conflict_share_of_lds_cycles": v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"],
            "waves_waiting": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
            "bound_by": ("vector issue and the one LDS per CU together" if lds_busy > 0.5 and valu_busy > 0.7 else
                         "latency (waves waiting)" if v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"] > 0.55 and valu_busy < 0.7 else "vector issue"),
            "counters": {k: v[k] for k in sorted(v)},
        }
        tra[cls] = {"kernel": "lzs_compress_blocks_wg_kernel", "launch": "16384 blocks x 64 KiB", "variant_forced": VARIANT[cls], "scope": SCOPE,
                    "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                    "fetch_bytes_corrected": int(2 * v["FETCH_SIZE"] * 1024), "write_bytes": int(v["WRITE_SIZE"] * 1024), "note": NOTE}
    json.dump(lim, open(os.path.join(outdir, "pmc_limiter.json"), "w"), indent=1)
    json.dump(tra, open(os.path.join(outdir, "pmc_traffic.json"), "w"), indent=1)
    for cls in ("text", "lowent", "random"):
        r = lim[cls]
        print(f"{cls:7} valu/byte {r['valu_insts_per_input_byte']:.2f}  valu busy {r['valu_issue_busy']:.2f}  lds busy {r['lds_busy']:.2f}  "
              f"conflicts {r['lds_bank_conflict_share_of_lds_cycles']:.2f}  waiting {r['waves_waiting']:.2f}  -> {r['bound_by']}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles"))
