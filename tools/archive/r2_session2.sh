#!/bin/bash
# Round-2 GPU session 2: changed tests (interleave, sharder, rehearsals), the 1-GPU bench, rocprofv3 kernel stats of the
# bench command, PMC passes over the UNet kernels.  Results in gpurun_out/s2/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
( time timeout 900 python -m pytest -q -m gpu --timeout 600 -p no:cacheprovider -s tests/test_multiproc_gpu.py "tests/test_hip_parity.py::test_interleaved_images_equal_running_each_alone" tests/test_real_arch_parity.py::test_fused_kernels_are_inside_the_bf16_loop "tests/test_unet_kernels.py::test_groupnorm" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "FAILED|Error" $O/pytest.log | head -20
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/s2/bench.json') if l.startswith('{')][-1])
    for k in ('value', 'images_per_min', 'ms_per_step', 'graphs', 'layouts', 'extras', 'phase_ms_last_image', 'host_ms_last_image', 'roofline', 'roofline_e2e', 'cpu_baseline', 'parity_bf16_rel_l2'):
        print(k, d.get(k))
    print({k: (v['mean_us'], v['ms_per_image'], v['tflops'], v['gbs']) for k, v in d['unet_kernels'].items()})
except Exception as e:
    print('bench parse failed', e)
PY
mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof/run.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -14 $O/trace_summary.txt
find $O/prof -name "*kernel_trace.csv" -delete
head -25 $O/bench_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -o unet -- python $GRAFT_REPO_ROOT/tools/pmc_unet.py > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  tail -1 $d/run.log
  find $d -name "*kernel_trace.csv" -delete
done
python tools/pmc_summarise.py $O/r2_unet_pmc.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | tail -60
du -sh $O
