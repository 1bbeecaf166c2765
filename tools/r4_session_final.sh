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
