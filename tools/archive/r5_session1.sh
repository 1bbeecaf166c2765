#!/bin/bash
# Round-5 GPU session 1 (~17 GPU-minutes): patches 0001-0005 are IN the product tree (landed before this session); this is the
# validation that decides whether they stay:
#   1. the new product library against the round-4 product library (tools/r5_patches/build/libelastic_hip_r4_product.so), entry point by
#      entry point: bit identity + timing
#   2. the complete `pytest -m gpu -x -q`
#   3. smoke()
#   4. in-situ A/B: the hipGraph-replayed SDXL forward at batch 20 / 6, old library vs new, one process, interleaved replays
#   5. bench.py --steps 4 --warmup 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s1; mkdir -p $O
( time timeout 300 python tools/r5_patches/probe_patched.py --rounds 5 ) > $O/new_vs_r4_product.jsonl 2> $O/new_vs_r4_product.err
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5s1/new_vs_r4_product.jsonl") if l.startswith("{")]
skipped = [r for r in rows if r.get("skipped")]
rows = [r for r in rows if not r.get("skipped")]
bad = [r["case"] for r in rows if not r["bit_identical"]]
print("probe:", len(rows), "cases,", len(bad), "not bit-identical", bad[:8], "| skipped:", [(r["case"], r["rc_base"], r["rc_new"]) for r in skipped])
for r in rows:
    print(f"{r['case'][:72]:72s} {r['product_tflops']:8.1f} -> {r['patched_tflops']:8.1f}  x{r['speedup']:.3f}")
PY
tail -3 $O/new_vs_r4_product.err
( time timeout 1150 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
tail -6 $O/pytest_gpu_full.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1
tail -3 $O/smoke.log
( time timeout 400 python tools/fwd_ab.py --libs tools/r5_patches/build/libelastic_hip_r4_product.so,product --batches 20,6 ) > $O/fwd_ab.jsonl 2> $O/fwd_ab.err
cat $O/fwd_ab.jsonl; tail -2 $O/fwd_ab.err
( time timeout 600 python bench.py --gpus 1 --steps 4 --warmup 2 ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5s1/bench.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("bench", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"))
print(d.get("parity_16bit_rel_l2", {}).get("gate_vs_reference_gpu_arithmetic"), d["extras"], d["graphs"])
print({k: (v.get("tflops") or v.get("gbps"), v.get("s_per_image")) for k, v in d.get("unet_kernels", {}).items()} if isinstance(d.get("unet_kernels"), dict) else d.get("unet_kernels"))
PY
tail -2 $O/bench.err
