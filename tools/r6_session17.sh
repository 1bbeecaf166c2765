#!/bin/bash
# Round-6 GPU session 17 (~20 GPU-minutes): the other benchmarked configurations on the tree with the fusions and the two-images-in-flight
# default, each with the live fp32 leg (row T): cfg2 SD 1.5 512x1024 (fp32 residual stream by default), cfg5 SDXL + ControlNet, cfg4 2048x2048 tiled.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s17; mkdir -p $O
for wl in sd15_512x1024 sdxl_1024x2048_controlnet sdxl_2048x2048_tiled; do
  ( time timeout 1500 python bench.py --gpus 1 --workload $wl --steps 4 --warmup 2 --fp32-leg on --no-cpu-baseline ) > $O/bench_$wl.json 2> $O/bench_$wl.err
  python - "$O/bench_$wl.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print(d["config"]["workload"], d["config"].get("precision_mode"), d["config"]["images_in_flight"], d["value"], d["ms_per_step"], d.get("latency_s_per_image"), d["extras"], r.get("kernel"), r.get("frac"))
    print(json.dumps(d["tolerance"].get("fp32_unet_same_workload"))[:400], d["tolerance"].get("meets_1e-3"), d["graphs"])
except Exception as e:
    print("no line", sys.argv[1], e)
PY
  tail -2 $O/bench_$wl.err
done
