#!/bin/bash
# Round-3 GPU session 6 (final numbers): the attention tests again (variant 5 with three K buffers), then the headline bench
# with the committed defaults (fp16, attention variant 4) and the rocprofv3 kernel stats of the same command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
( time timeout 200 python -m pytest -q -m gpu --timeout 150 -p no:cacheprovider tests/test_unet_kernels.py -k "flash" ) > $O/pytest_flash.log 2>&1
tail -3 $O/pytest_flash.log; grep -E "^FAILED|^ERROR" $O/pytest_flash.log | cut -c1-120 | head
( time timeout 400 python bench.py ) > $O/bench_fp16_full.json 2> $O/bench_fp16_full.err; grep -v "amdgpu.ids" $O/bench_fp16_full.err | tail -2
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/s6/bench_fp16_full.json') if l.startswith('{')][-1])
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'dtype', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e', 'extras')})
    print(d['roofline'])
    print({k: (v['mean_us'], v['ms_per_image'], v['tflops'], v['gbs']) for k, v in d['unet_kernels'].items()})
    print(d.get('parity_16bit_rel_l2'))
except Exception as e:
    print('bench parse failed', e)
PY
mkdir -p $O/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof/run.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -8 $O/trace_summary.txt
find $O/prof -name "*kernel_trace.csv" -delete
head -8 $O/bench_kernel_stats.csv | cut -c1-150
tar czf $O/miopen_cache.tgz miopen_cache
du -sh $O
