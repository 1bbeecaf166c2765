#!/bin/bash
# Round-6 GPU session 23 (~8 GPU-minutes): the live fp32 leg on three seeds of the headline workload (final tree, two images in flight)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s23; mkdir -p $O
( time timeout 1200 python bench.py --gpus 1 --steps 4 --warmup 2 --fp32-leg on --fp32-leg-seeds 3 --no-cpu-baseline --no-extras ) > $O/bench_fp32_3seeds.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6s23/bench_fp32_3seeds.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], json.dumps(d["tolerance"].get("fp32_unet_same_workload"))[:400], d["tolerance"].get("meets_1e-3"))
PY
tail -2 $O/bench.err
