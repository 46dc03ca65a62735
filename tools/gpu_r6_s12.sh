#!/bin/bash
# round 6, session 12: one-byte exit functions (the tree) against the commit before them, through the library's launcher, on the three
# classes; then what depends on the kernel sources and was measured before them: chain-safe rates, block-size sweep, 20 minutes of fuzz
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for c in 2 1 0; do bash tools/gpu_ab_cls.sh s49 $c lhead ltree; done
cat $OUT/ab_s49.txt
timeout 600 python bench.py --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_text_start.json
for wl in text lowent random; do
  LZS_CHAIN_FALLBACK=1 timeout 300 python bench.py --steps 10 --warmup 2 --workload $wl --no-config5 --no-other-classes --no-cpu-baseline --no-single-stream 2>/dev/null | tail -1 > $OUT/bench_chain_safe_$wl.json
done
python - <<'PY' > $OUT/chain_safe_variant.txt
import json, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06")
print("# LZS_CHAIN_FALLBACK=1: CHAIN in its order-independent form (wgv_safe), what a device that fails lzs_lds_order_check_kernel gets;")
print("# bench.py --steps 10 --warmup 2 --workload <class>, 1 GiB of 64 KiB blocks, bit-exact check of the launch included in bench.py")
for wl in ("text", "lowent", "random"):
    try:
        d = json.loads(open(os.path.join(out, f"bench_chain_safe_{wl}.json")).read())
        print(f"{wl:8} {d['value']:8.2f} GB/s  ms_per_step {d['ms_per_step']:.3f}  roofline.frac {d['roofline']['frac']:.5f}")
    except Exception as e:
        print(wl, "failed:", e)
d = json.loads(open(os.path.join(out, "bench_text_start.json")).read())
print(f"(default CHAIN, text, same box: {d['value']:.2f} GB/s)")
PY
cat $OUT/chain_safe_variant.txt
timeout 1500 python tests/dev/blocksize_sweep.py 1024 > $OUT/blocksize_sweep.txt 2>$OUT/blocksize_sweep.err
cat $OUT/blocksize_sweep.txt
timeout 1400 python tests/dev/fuzz_all.py 1200 15001 > $OUT/fuzz_1200s_seed15001.txt 2>&1
tail -3 $OUT/fuzz_1200s_seed15001.txt
