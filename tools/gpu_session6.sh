#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -25
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r1e.json 2> gpurun_out/bench_r1e.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_r1e.json'))
print({k:d[k] for k in ('value','images_per_min','ms_per_step','phase_ms_last_image','host_ms_last_image','roofline','roofline_e2e')})
PY
grep -v amdgpu.ids gpurun_out/bench_r1e.err | tail -5
timeout 300 python tools/run_configs.py cfg5 2 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/run_configs.py cfg4 2 2>&1 | grep -v amdgpu.ids | tail -4
tar czf gpurun_out/miopen_cache.tgz miopen_cache
