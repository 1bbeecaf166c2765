#!/bin/bash
# Round-4 GPU session 9 (~4 GPU-minutes): the headline workload with two images in flight on the one GPU (batches 40 / 12),
# and the bf16 line with the round-4 kernels.  Extras for DESIGN.md; the headline stays one image at a time.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s9; mkdir -p $O
( time timeout 400 python bench.py --in-flight 2 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_sdxl_inflight2.json 2> $O/bench_sdxl_inflight2.err
( time timeout 300 python bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_sdxl_bf16.json 2> $O/bench_sdxl_bf16.err
python - <<'PY'
import json
for n in ("sdxl_inflight2", "sdxl_bf16"):
    d = json.loads([l for l in open(f"gpurun_out/r4s9/bench_{n}.json") if l.startswith("{")][-1])
    print(n, d["value"], d["ms_per_step"], d["latency_s_per_image"], d["graphs"], d["dtype"])
PY
