#!/bin/bash
# Round-2 GPU session 6: the full -m gpu suite + smoke with the final defaults (channels-last, fused glue), the 1-GPU bench,
# rocprofv3 kernel stats of the bench command, PMC passes (UNet kernels + glue), the other BASELINE configs.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "FAILED|Error" $O/pytest.log | head -20
( time timeout 600 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; grep -E "smoke|Error|error" $O/smoke.log | tail -5
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/s6/bench.json') if l.startswith('{')][-1])
    for k in ('value', 'images_per_min', 'ms_per_step', 'graphs', 'layouts', 'extras', 'phase_ms_last_image', 'roofline', 'roofline_e2e', 'cpu_baseline', 'parity_bf16_rel_l2'):
        print(k, d.get(k))
    print({k: (v['mean_us'], v['ms_per_image'], v['tflops'], v['gbs']) for k, v in d['unet_kernels'].items()})
    print({k: (v['us_per_launch'], v['gbs'], v['launches_in_timed_region'], v['in_situ_us']) for k, v in d['glue_kernels'].items()})
except Exception as e:
    print('bench parse failed', e)
PY
mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof/run.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -8 $O/trace_summary.txt
find $O/prof -name "*kernel_trace.csv" -delete
head -16 $O/bench_kernel_stats.csv | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  d=$O/pmc_unet_$(echo $c | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -o unet -- python $GRAFT_REPO_ROOT/tools/pmc_unet.py > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  find $d -name "*kernel_trace.csv" -delete
done
python tools/pmc_summarise.py $O/r2_unet_pmc.json $O/pmc_unet_FETCH_SIZE $O/pmc_unet_WRITE_SIZE $O/pmc_unet_SQ_VALU_MFMA_BUSY_CYCLES > $O/pmc_unet_summary.txt 2>&1; tail -3 $O/pmc_unet_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc_glue_$c; mkdir -p $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -o glue -- python $GRAFT_REPO_ROOT/tools/pmc_glue.py > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  find $d -name "*kernel_trace.csv" -delete
done
python tools/pmc_summarise.py $O/r2_glue_pmc.json $O/pmc_glue_FETCH_SIZE $O/pmc_glue_WRITE_SIZE > $O/pmc_glue_summary.txt 2>&1; tail -3 $O/pmc_glue_summary.txt
( time timeout 400 python bench.py --workload sdxl_2048x2048_tiled --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -2 $O/bench_cfg4.err; cut -c1-700 $O/bench_cfg4.json
( time timeout 300 python bench.py --workload sd15_512x1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -2 $O/bench_cfg2.err; cut -c1-700 $O/bench_cfg2.json
( time MIOPEN_FIND_MODE=FAST timeout 400 python tools/run_configs.py cfg5 3 ) > $O/cfg5.log 2>&1; grep -E "^cfg5|Error" $O/cfg5.log
du -sh $O
