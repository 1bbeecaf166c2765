#!/bin/bash
# Round-3 GPU session 5 (final): lazy-maximum attention variant (tests + probe, chosen per measurement), the final headline
# bench line (fp16 default, full), the bf16 line on the same box, rocprofv3 kernel stats of the bench command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
# per-rank batches of 2/4/8-way row sharding in fp16: a clean single-process MIOpen find (the records session 4 produced with
# eight processes sharing this GPU were discarded), so the driver's multi-GPU runs start without a find
for dt in fp16; do ( ED_CL=1 ED_DTYPE=$dt timeout 420 python tools/probe_unet.py sdxl 10,3 ) 2>&1 | grep "B=" ; done | tee $O/probe_unet_b10_b3_fp16.log
tar czf $O/miopen_cache.tgz miopen_cache
( time timeout 300 python -m pytest -q -m gpu --timeout 250 -p no:cacheprovider tests/test_unet_kernels.py tests/test_models_and_text.py -k "flash or snapshot" ) > $O/pytest_flash.log 2>&1
tail -3 $O/pytest_flash.log; grep -E "^FAILED|^ERROR" $O/pytest_flash.log | head
( time timeout 200 python tools/r3_probe.py attn ) > $O/probe_attn.log 2>&1; grep "^{" $O/probe_attn.log | cut -c1-700
VARIANT=$(python - <<'PY'
import json
ok = 'failed' not in open('gpurun_out/s5/pytest_flash.log').read().split('\n')[-2] and ' passed' in open('gpurun_out/s5/pytest_flash.log').read()
rows = [json.loads(l) for l in open('gpurun_out/s5/probe_attn.log') if l.startswith('{')]
big = [r for r in rows if r['Nk'] >= 1024 and r['B'] >= 6]
faster = all(r.get('v5_us', 1e9) < 0.99 * r['v4_us'] for r in big) and bool(big)
print(5 if (ok and faster) else 4)
PY
)
echo "attention variant for the final runs: $VARIANT" | tee $O/variant.txt
export ED_FLASH_VARIANT=$VARIANT
( time timeout 700 python bench.py ) > $O/bench_fp16_full.json 2> $O/bench_fp16_full.err; grep -v "amdgpu.ids" $O/bench_fp16_full.err | tail -2
( time timeout 300 python bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_bf16.json 2> $O/bench_bf16.err; grep -v "amdgpu.ids" $O/bench_bf16.err | tail -2
python - <<'PY'
import json
for name in ('bench_fp16_full', 'bench_bf16'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/s5/{name}.json') if l.startswith('{')][-1])
        print(name, {k: d.get(k) for k in ('value', 'ms_per_step', 'dtype', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e', 'extras')})
        print(d['roofline'])
        print({k: (v['mean_us'], v['ms_per_image'], v['tflops'], v['gbs']) for k, v in d['unet_kernels'].items()})
        print(d.get('cpu_baseline'))
    except Exception as e:
        print('bench parse failed', name, e)
PY
mkdir -p $O/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof/run.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -8 $O/trace_summary.txt
find $O/prof -name "*kernel_trace.csv" -delete
head -12 $O/bench_kernel_stats.csv | cut -c1-150
tar czf $O/miopen_cache.tgz miopen_cache
du -sh $O
