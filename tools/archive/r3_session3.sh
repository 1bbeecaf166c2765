#!/bin/bash
# Round-3 GPU session 3: after the GroupNorm dispatch fix -- kernel tests + RCCL test, fp16 vs bf16 forward time at the
# per-rank batch sizes, the headline bench (fp16 default, full line) and bf16, rocprofv3 kernel stats, PMC passes, precision.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
( time timeout 500 python -m pytest -q -m gpu --timeout 400 -p no:cacheprovider tests/test_unet_kernels.py tests/test_multiproc_gpu.py::test_rccl_world_size_one_exchange_path tests/test_real_arch_parity.py::test_fused_kernels_are_inside_the_bf16_loop tests/test_real_arch_parity.py::test_full_width_sdxl_forward_16bit_error_is_the_dtypes ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "^FAILED|Error" $O/pytest.log | head -20
( time timeout 300 python tools/r3_probe.py gn ) > $O/probe_gn.log 2>&1; grep "^{" $O/probe_gn.log
for dt in fp16 bf16; do ( ED_CL=1 ED_DTYPE=$dt timeout 200 python tools/probe_unet.py sdxl 20,10,6,3 ) 2>&1 | grep "B=" ; done | tee $O/probe_unet_dtypes.log
( time timeout 600 python bench.py ) > $O/bench_fp16_full.json 2> $O/bench_fp16_full.err; grep -v "amdgpu.ids" $O/bench_fp16_full.err | tail -2
( time timeout 300 python bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_bf16.json 2> $O/bench_bf16.err; grep -v "amdgpu.ids" $O/bench_bf16.err | tail -2
python - <<'PY'
import json
for name in ('bench_fp16_full', 'bench_bf16'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/s3/{name}.json') if l.startswith('{')][-1])
        print(name, {k: d.get(k) for k in ('value', 'ms_per_step', 'dtype', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e', 'extras', 'tolerance', 'parity_16bit_rel_l2')})
        print(d['roofline'])
        print({k: (v['mean_us'], v['ms_per_image'], v['tflops'], v['gbs']) for k, v in d['unet_kernels'].items()})
        print(d.get('cpu_baseline'))
    except Exception as e:
        print('bench parse failed', name, e)
PY
mkdir -p $O/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof/run.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -14 $O/trace_summary.txt
find $O/prof -name "*kernel_trace.csv" -delete
head -22 $O/bench_kernel_stats.csv | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  d=$O/pmc_$(echo $c | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -o unet -- python $GRAFT_REPO_ROOT/tools/pmc_unet.py > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  tail -1 $d/run.log
  find $d -name "*kernel_trace.csv" -delete
done
python tools/pmc_summarise.py $O/r3_unet_pmc.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | tail -70
( time timeout 300 python tools/r3_precision.py full ) > $O/precision_full.log 2>&1; grep "^{" $O/precision_full.log | cut -c1-1200
cp gpurun_out/r3_precision.json $O/r3_precision_full.json 2>/dev/null
du -sh $O
