#!/bin/bash
# Round-5 GPU session 3 (~8 GPU-minutes): the fp32 VAE's ResnetBlock convolutions on split fp16 operands (models.VAE_SPLIT_CONV):
#   1. tests/test_vae_split.py (kernels vs fp64, block and whole-VAE vs the library path) + the tests around the VAE
#   2. tools/r5_vae_ab.py: pad-strip encode / 1024 x 2048 decode / 8 decode tiles, library vs split, with per-kernel times
#   3. the headline bench (2 images) and cfg4 (1 image): strips and decode phases
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s3; mkdir -p $O
# 0. the library after the long-K loop for convolutions and the fp32-output epilogue: still bit-identical to the round-4 library on every
#    16-bit entry point, and the cumulative in-situ forward gain
( time timeout 300 python tools/r5_patches/probe_patched.py --rounds 3 ) > $O/new_vs_r4_product.jsonl 2> $O/new_vs_r4_product.err
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5s3/new_vs_r4_product.jsonl") if l.startswith("{")]
rows = [r for r in rows if not r.get("skipped")]
bad = [r["case"] for r in rows if not r["bit_identical"]]
print("probe:", len(rows), "cases,", len(bad), "not bit-identical", bad[:8])
for r in rows:
    if "conv" in r["case"]:
        print(f"{r['case'][:72]:72s} {r['product_tflops']:8.1f} -> {r['patched_tflops']:8.1f}  x{r['speedup']:.3f}")
PY
tail -2 $O/new_vs_r4_product.err
( time timeout 300 python tools/fwd_ab.py --libs tools/r5_patches/build/libelastic_hip_r4_product.so,product --batches 20,6 ) > $O/fwd_ab.jsonl 2> $O/fwd_ab.err
cat $O/fwd_ab.jsonl; tail -2 $O/fwd_ab.err
( time timeout 600 python -m pytest tests/test_vae_split.py tests/test_models_and_text.py tests/test_abi.py -m gpu -x -q ) > $O/pytest_vae.log 2>&1
tail -15 $O/pytest_vae.log
( time timeout 400 python tools/r5_vae_ab.py ) > $O/vae_ab.jsonl 2> $O/vae_ab.err
cat $O/vae_ab.jsonl; tail -3 $O/vae_ab.err
( time timeout 400 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --fp32-leg off ) > $O/bench_headline.json 2> $O/bench_headline.err
( time timeout 500 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --workload sdxl_2048x2048_tiled ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'PY'
import json
for f in ("bench_headline", "bench_cfg4"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r5s3/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["extras"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -q -n 3 $O/bench_headline.err $O/bench_cfg4.err
