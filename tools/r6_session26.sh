#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s26; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "batch_split or conv3x3_nhwc or tile_height or wrappers" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( time timeout 600 python bench.py --gpus 1 --steps 4 --warmup 2 --fp32-leg off --no-extras --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6s26/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["graphs"], d["finite_output"], {k: (v.get("tflops"), v["launches_per_image"]) for k, v in d["unet_kernels"].items() if "conv" in k})
PY
tail -2 $O/bench.err
