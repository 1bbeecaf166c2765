#!/bin/bash
# Round-4 FINAL GPU session, second edition (~11 GPU-minutes): the GEMM main loop changed after the first one (early start, bias
# loads not waited for before staging: bit-identical results, +2-4 % on the short-K shapes), so the driver's round-end commands
# run again on the final tree: the complete `pytest -m gpu -x -q`, smoke(), bench.py.  No product change after this run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4final2; mkdir -p $O
( time timeout 1150 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
tail -5 $O/pytest_gpu_full.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1
tail -3 $O/smoke.log
( time timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 ) > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4final2/bench_final.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("final", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"))
print(d.get("parity_16bit_rel_l2", {}).get("gate_1p5x_reference_pattern"), d["extras"], d["graphs"])
PY
tail -2 $O/bench_final.err
