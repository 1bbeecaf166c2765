#!/bin/bash
# Round-5 FINAL GPU session, second edition (~22 GPU-minutes): the tree changed after the first one (the VAE decoder's upsampler convolutions
# on the split-operand path, multi-seed fp32 leg), so the driver's round-end commands run again on the final tree: the complete
# `pytest -m gpu -x -q`, smoke(), bench.py -- then the other BASELINE configurations with the round-5 kernels (cfg5 ControlNet, cfg2 SD 1.5)
# and the 8-rank full-size cold rehearsal of the driver's multi-GPU command (N ranks sharing the one GPU over gloo).
# No product change after this run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5final2; mkdir -p $O
( time timeout 1150 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
tail -5 $O/pytest_gpu_full.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1
grep smoke $O/smoke.log | tail -3
( time timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 ) > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5final2/bench_final.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("final", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"), r.get("traffic"))
print(d.get("parity_16bit_rel_l2", {}).get("gate_vs_reference_gpu_arithmetic"), d["extras"], d["graphs"])
print(json.dumps(d["tolerance"].get("fp32_unet_same_workload")))
PY
tail -2 $O/bench_final.err
( time timeout 400 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --workload sdxl_1024x2048_controlnet ) > $O/bench_cfg5.json 2> $O/bench_cfg5.err
( time timeout 300 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --workload sd15_512x1024 ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
( time ED_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29568 bench.py --gpus 8 --steps 4 --warmup 1 --no-kernel-timing --no-extras ) > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err
python - <<'PY'
import json
for f in ("bench_cfg5", "bench_cfg2", "bench_8rank_gloo"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r5final2/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["phase_ms_last_image"], d.get("rccl"), d["graphs"], d.get("rows_computed_over_rows_total_rank0"), d["finite_output"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -q -n 2 $O/bench_cfg5.err $O/bench_cfg2.err $O/bench_8rank_gloo.err
