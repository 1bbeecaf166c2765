#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 300 python tools/probe_unet.py sdxl 10,3 find ) 2>&1 | grep -v amdgpu.ids | tail -5
ED_MIOPEN_FIND=1 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('find-mode run', d['value'], d['phase_ms_last_image'])"
tar czf gpurun_out/miopen_cache.tgz miopen_cache
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r1g.json 2> gpurun_out/bench_r1g.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_r1g.json'))
print({k:d[k] for k in ('value','images_per_min','ms_per_step','phase_ms_last_image','extras','roofline_e2e','cpu_baseline')})
PY
grep -v amdgpu.ids gpurun_out/bench_r1g.err | tail -3
tar czf gpurun_out/miopen_cache.tgz miopen_cache; du -sh miopen_cache
