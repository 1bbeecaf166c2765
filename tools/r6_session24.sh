#!/bin/bash
# Round-6 GPU session 24 (~7 GPU-minutes): two vs three images in flight on the final tree, same box, 12 timed images each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s24; mkdir -p $O
for m in 2 3 2 3; do
  timeout 600 python bench.py --in-flight $m --steps 12 --warmup $m --no-extras --fp32-leg off --no-kernel-timing --no-cpu-baseline > $O/bench_inflight${m}_$RANDOM.json 2> $O/err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6s24/bench_inflight*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(d["config"]["images_in_flight"], d["value"], d["ms_per_step"], d.get("latency_s_per_image"))
PY
