#!/bin/bash
# Round-5 GPU session 2 (~12 GPU-minutes):
#   1. optional patch 0006 (long-K loop: 4 barrier intervals per K tile for K >= 1280) built on the landed product sources, against the
#      product library: bit identity + timing per entry point -> land it or drop it
#   2. the per-rank forward table of DESIGN 7 with the current library (tools/r5_batch_table.py)
#   3. the GPU tests the CPU-side changes of this round touch: autocast-shaped comparator + 1.25x gate (test_real_arch_parity), the
#      ADVICE r4 fixes (ResnetBlock gating, GEGLU grid policy: test_models_and_text, test_unet_kernels; k|v inside the graph for fresh side
#      inputs: test_interleaved_images..., the 8-rank rehearsal)
#   4. bench.py with the live fp32 leg (2 timed images, no CPU baseline)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s2; mkdir -p $O
( time timeout 300 python tools/r5_patches/probe_patched.py --rounds 5 --base product --lib libelastic_hip_with_0006.so ) > $O/with_0006_vs_product.jsonl 2> $O/with_0006_vs_product.err
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5s2/with_0006_vs_product.jsonl") if l.startswith("{")]
rows = [r for r in rows if not r.get("skipped")]
bad = [r["case"] for r in rows if not r["bit_identical"]]
print("0006 probe:", len(rows), "cases,", len(bad), "not bit-identical", bad[:8])
for r in rows:
    if "flash" not in r["case"]:
        print(f"{r['case'][:72]:72s} {r['product_tflops']:8.1f} -> {r['patched_tflops']:8.1f}  x{r['speedup']:.3f}")
PY
tail -3 $O/with_0006_vs_product.err
( time timeout 500 python tools/r5_batch_table.py ) > $O/batch_table.jsonl 2> $O/batch_table.err
python - <<'PY'
import json
for l in open("gpurun_out/r5s2/batch_table.jsonl"):
    if l.startswith("{"):
        r = json.loads(l)
        print(r["rows"], r["replay_ms"], r["ms_per_row"], r["first_eager_forward_s"], r["library_share_of_contraction_flops"], r["library_calls"])
PY
tail -3 $O/batch_table.err
( time timeout 900 python -m pytest tests/test_real_arch_parity.py tests/test_models_and_text.py tests/test_unet_kernels.py "tests/test_hip_parity.py::test_interleaved_images_equal_running_each_alone" "tests/test_multiproc_gpu.py::test_bench_multi_rank_rehearsal" -m gpu -x -q ) > $O/pytest_subset.log 2>&1
tail -6 $O/pytest_subset.log
cp gpurun_out/parity_real_arch.json $O/parity_real_arch.json 2>/dev/null
( time timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --fp32-leg on ) > $O/bench_fp32_leg.json 2> $O/bench_fp32_leg.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5s2/bench_fp32_leg.json") if l.startswith("{")][-1])
print("bench", d["value"], d["ms_per_step"], d["phase_ms_last_image"])
print(json.dumps(d["tolerance"].get("fp32_unet_same_workload")))
PY
tail -3 $O/bench_fp32_leg.err
