#!/bin/bash
# Round-5 GPU session 1 (~2 GPU-minutes): the library built from the product sources + tools/r5_patches/*.patch against the product
# library, entry point by entry point (bit identity + timing) -- the evidence for landing the patches.  Build the patched library first,
# here in the container:  python tools/r5_patches/build_patched.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s1; mkdir -p $O
( time timeout 240 python tools/r5_patches/probe_patched.py --rounds 5 ) > $O/patched_vs_product.jsonl 2> $O/patched_vs_product.err
# the optional long-K loop on top (python tools/r5_patches/build_patched.py --only 0001,0003,0005,0006 --out libelastic_hip_patched_with_0006.so)
if [ -f tools/r5_patches/build/libelastic_hip_patched_with_0006.so ]; then
  ( timeout 240 python tools/r5_patches/probe_patched.py --rounds 5 --lib libelastic_hip_patched_with_0006.so ) > $O/patched_with_0006_vs_product.jsonl 2>> $O/patched_vs_product.err
fi
python - <<'PY'
import json
import os
for name in ("patched_vs_product", "patched_with_0006_vs_product"):
    path = f"gpurun_out/r5s1/{name}.jsonl"
    if not os.path.isfile(path):
        continue
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    bad = [r["case"] for r in rows if not r["bit_identical"]]
    print(name, ":", len(rows), "cases,", len(bad), "not bit-identical", bad[:5])
    for r in rows:
        print(f"{r['case'][:70]:70s} {r['product_tflops']:8.1f} -> {r['patched_tflops']:8.1f}  x{r['speedup']:.3f}")
PY
tail -3 $O/patched_vs_product.err
