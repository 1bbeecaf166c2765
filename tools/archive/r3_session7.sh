#!/bin/bash
# Round-3 GPU session 7: cfg4 (SDXL 2048x2048, 16 views, 64-tile fp32 decode) end to end in fp16 -- its batch-32 / batch-18
# convolutions start from derived find-db records (tools/miopen_nhwc_from_nchw.py derive-find) instead of a find on first use.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
( time timeout 330 python bench.py --workload sdxl_2048x2048_tiled --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err; grep -v "amdgpu.ids" $O/bench_cfg4.err | tail -3
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/s7/bench_cfg4.json') if l.startswith('{')][-1])
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'dtype', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e')})
except Exception as e:
    print('bench parse failed', e)
PY
tar czf $O/miopen_cache.tgz miopen_cache
