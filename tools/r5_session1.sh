#!/bin/bash
# Round-5 GPU session 1 (~2 GPU-minutes): the library built from the product sources + tools/r5_patches/*.patch against the product
# library, entry point by entry point (bit identity + timing) -- the evidence for landing the patches.  Build the patched library first,
# here in the container:  python tools/r5_patches/build_patched.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s1; mkdir -p $O
( time timeout 240 python tools/r5_patches/probe_patched.py --rounds 5 ) > $O/patched_vs_product.jsonl 2> $O/patched_vs_product.err
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5s1/patched_vs_product.jsonl") if l.startswith("{")]
bad = [r["case"] for r in rows if not r["bit_identical"]]
print(len(rows), "cases,", len(bad), "not bit-identical", bad[:5])
for r in rows:
    print(f"{r['case'][:70]:70s} {r['product_tflops']:8.1f} -> {r['patched_tflops']:8.1f}  x{r['speedup']:.3f}")
PY
tail -3 $O/patched_vs_product.err
