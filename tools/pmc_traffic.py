#!/usr/bin/env python3
"""Turns the counter CSVs of tools/gpu_traffic.sh into pmc_traffic.json (HBM bytes per launch of the
compress kernel, per class), stamped with the identity of the kernel sources they belong to."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_identity  # noqa: E402

NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace only); counters are in KB; "
        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests of wide streaming reads at 64 B)")


def mean_counter(directory, counter):
    vals = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "lzs_compress_blocks_wg_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for the compress kernel under {directory}")
    return sum(vals) / len(vals), len(vals)


def main(src, dst):
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    out = {"kernel_source_sha256": kernel_identity(), "git_head": head or "(no git on the GPU box: see the commit that adds this file)"}
    for cls in ("text", "lowent", "random"):
        fetch, n1 = mean_counter(os.path.join(src, f"{cls}_FETCH_SIZE"), "FETCH_SIZE")
        write, n2 = mean_counter(os.path.join(src, f"{cls}_WRITE_SIZE"), "WRITE_SIZE")
        out[cls] = {"kernel": "lzs_compress_blocks_wg_kernel", "launch": "16384 blocks x 64 KiB",
                    "variant_forced": {"text": "text", "lowent": "few", "random": "lit"}[cls],
                    "scope": "one dispatch of the class's own variant (LZS_VARIANT forced); the classifier and the other variants' empty grids of a caller's launch are not in the counters",
                    "dispatches_averaged": [n1, n2], "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
                    "fetch_bytes_corrected": int(2 * fetch * 1024), "write_bytes": int(write * 1024), "note": NOTE}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
