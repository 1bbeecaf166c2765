#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r1d.json 2> gpurun_out/bench_r1d.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_r1d.json'))
print({k:d[k] for k in ('value','images_per_min','ms_per_step','phase_ms_last_image','host_ms_last_image','roofline_e2e','cpu_baseline')})
print({k:(v['us_per_launch'],v['gbs']) for k,v in d['glue_kernels'].items()})
PY
grep -v amdgpu.ids gpurun_out/bench_r1d.err | tail -5
tar czf gpurun_out/miopen_cache.tgz miopen_cache
