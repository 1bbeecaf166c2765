#!/bin/bash
# Round-4 FINAL GPU session (~17 GPU-minutes): what the driver runs at round end, on the final tree -- the complete
# `pytest -m gpu -x -q`, smoke(), and `bench.py --gpus 1 --steps 20 --warmup 5`.  No product change after this run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4final; mkdir -p $O
( time timeout 1150 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
tail -6 $O/pytest_gpu_full.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1
tail -4 $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4final/bench_final.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("final", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("traffic"))
print(d.get("cpu_baseline", {}).get("value"), d.get("parity_16bit_rel_l2", {}).get("gate_1p5x_reference_pattern"), d["extras"])
PY
tail -3 $O/bench_final.err
# the driver's multi-GPU commands on the final tree, N ranks sharing this one GPU over gloo (full-size fp16 model, cold)
for n in 2 8; do
  ( time ED_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29560+n)) bench.py --gpus $n --steps $((n/2 > 1 ? n/2 : 2)) --warmup 1 --no-kernel-timing --no-extras ) > $O/bench_${n}rank_gloo.json 2> $O/bench_${n}rank_gloo.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r4final/bench_${n}rank_gloo.json") if l.startswith("{")][-1])
    print("${n} ranks", d["value"], d["rccl"], d["graphs"], d["rows_computed_over_rows_total_rank0"], d["finite_output"])
except Exception as e:
    print("${n} ranks failed", e)
PY
  tail -3 $O/bench_${n}rank_gloo.err | cut -c1-200
done
du -sh $O
