#!/bin/bash
# Round-6 GPU session 3 (~25 GPU-minutes): the HIP layers of the fp32-residual-stream mode (ed_layernorm_s32 / ed_add_layernorm_s32 /
# ed_groupnorm_nhwc_s32), its accuracy and cost on cfg2 and on the headline workload, attention back on the round-5 loop.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s3; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_unet_kernels.py tests/test_models_and_text.py tests/test_real_arch_parity.py -m gpu -x -q ) > $O/pytest_gpu_subset.log 2>&1; tail -5 $O/pytest_gpu_subset.log
cp gpurun_out/long_schedule_parity.json $O/ 2>/dev/null; cat $O/long_schedule_parity.json 2>/dev/null | tr -d '\n ' | cut -c1-1200; echo
timeout 200 python tools/r6_ops_ab.py --prev tools/ab/libelastic_hip_r5.so --rounds 5 --only attn > $O/ops_ab_attn.jsonl 2> $O/ops_ab.err; cat $O/ops_ab_attn.jsonl | cut -c1-200
for mode in "" "--residual-fp32"; do
  tag=plain; [ -n "$mode" ] && tag=stream32
  ( time timeout 600 python bench.py --gpus 1 --workload sd15_512x1024 --steps 3 --warmup 1 --fp32-leg on --fp32-leg-seeds 2 --no-cpu-baseline --no-extras $mode ) > $O/bench_cfg2_$tag.json 2> $O/bench_cfg2_$tag.err
  python - "$O/bench_cfg2_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    leg = d["tolerance"].get("fp32_unet_same_workload", {})
    print(d["config"]["workload"], d["config"].get("precision_mode"), d["value"], d["ms_per_step"], leg.get("fp16_latent_vs_fp32_latent_rel_l2_by_seed"), d["tolerance"].get("meets_1e-3"), d["graphs"])
    print({k: (v.get("tflops") or v.get("gbs"), v.get("ms_per_image")) for k, v in d.get("unet_kernels", {}).items()})
except Exception as e:
    print("no line", sys.argv[1], e)
PY
  tail -2 $O/bench_cfg2_$tag.err
done
for mode in "--residual-fp32" ""; do
  tag=plain; [ -n "$mode" ] && tag=stream32
  ( time timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --fp32-leg on --no-cpu-baseline --no-extras $mode ) > $O/bench_cfg3_$tag.json 2> $O/bench_cfg3_$tag.err
  python - "$O/bench_cfg3_$tag.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    leg = d["tolerance"].get("fp32_unet_same_workload", {})
    r = d.get("roofline") or {}
    print("cfg3", d["config"].get("precision_mode"), d["value"], d["ms_per_step"], leg.get("fp16_latent_vs_fp32_latent_rel_l2_by_seed"), d["tolerance"].get("meets_1e-3"), r.get("kernel"), r.get("frac"), d["roofline_e2e"]["frac"])
    print({k: (v.get("tflops") or v.get("gbs"), v.get("ms_per_image")) for k, v in d.get("unet_kernels", {}).items()})
except Exception as e:
    print("no cfg3 line", e)
PY
  tail -2 $O/bench_cfg3_$tag.err
done
du -sh $O
