#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp -r miopen_cache /tmp/miopen_cache_before
( time timeout 480 python tools/probe_unet.py sdxl 20,6 find ) 2>&1 | grep -v amdgpu.ids | tail -6
python tools/probe_unet.py sdxl 20,6 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_r1f.json 2> gpurun_out/bench_r1f.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_r1f.json'))
print({k:d[k] for k in ('value','images_per_min','ms_per_step','phase_ms_last_image','host_ms_last_image','extras','roofline_e2e')})
PY
grep -v amdgpu.ids gpurun_out/bench_r1f.err | tail -5
mkdir -p gpurun_out/prof_final
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o bench50 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_final/run.log 2>&1)
python tools/analyze_trace.py gpurun_out/prof_final/bench50_kernel_trace.csv > gpurun_out/prof_final/trace_summary.txt 2>&1; head -12 gpurun_out/prof_final/trace_summary.txt
rm -f gpurun_out/prof_final/bench50_kernel_trace.csv
tar czf gpurun_out/miopen_cache.tgz miopen_cache; du -sh miopen_cache
