#!/bin/bash
# Round-3 GPU session 2: new kernels (attention v2 variants, 2-launch GroupNorm, channels-last bias add, softmax rows / VAE
# attention) -- tests, probe timings, PMC passes over the attention variants, MIOpen find for the fp32 VAE shapes, bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
( time timeout 400 python -m pytest -q -m gpu --timeout 300 -p no:cacheprovider tests/test_multiproc_gpu.py::test_rccl_world_size_one_exchange_path tests/test_hip_parity.py::test_verbose_init_low_matches_oracle tests/test_hip_parity.py::test_verbose_image_log tests/test_cli_gpu.py ) > $O/pytest_new.log 2>&1
tail -4 $O/pytest_new.log; grep -E "FAILED|Error" $O/pytest_new.log | head -20
( time timeout 700 python -m pytest -q -m gpu --timeout 600 -p no:cacheprovider tests/test_unet_kernels.py ) > $O/pytest_kernels.log 2>&1
tail -4 $O/pytest_kernels.log; grep -E "FAILED|Error" $O/pytest_kernels.log | head -20
( time timeout 300 python tools/r3_probe.py attn gn vae ) > $O/probe.log 2>&1; grep "^{" $O/probe.log; tail -2 $O/probe.log | grep -v "^{"
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$O/pmc_$(echo $pass | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$d -o attn -- python $GRAFT_REPO_ROOT/tools/r3_probe.py pmcattn > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  tail -1 $d/run.log
  find $d -name "*kernel_trace.csv" -delete
done
python tools/pmc_by_kernel.py $O/r3_attn_pmc.json $O/pmc_SQ_WAVE_CYCLES $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES --match flash_attn 2>&1 | tail -120
# fp16 = the reference's own UNet dtype (ED:1012) and 8x less drift than bf16 (profiles/r3_precision.json): MIOpen find for the
# fp16 channels-last convolutions at the bench's batch sizes (20, 6), then the bench in both dtypes with the new kernels
( time ED_MIOPEN_FIND=1 timeout 700 python bench.py --dtype fp16 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_fp16_find.json 2> $O/bench_fp16_find.err; grep -v "amdgpu.ids" $O/bench_fp16_find.err | tail -3
tar czf $O/miopen_cache.tgz miopen_cache; ls -la miopen_cache | head
for dt in fp16 bf16; do
  ( time timeout 300 python bench.py --dtype $dt --steps 1 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_$dt.json 2> $O/bench_$dt.err; grep -v "amdgpu.ids" $O/bench_$dt.err | tail -2
  python - $dt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/s2/bench_{sys.argv[1]}.json') if l.startswith('{')][-1])
    print(sys.argv[1], {k: d.get(k) for k in ('value', 'ms_per_step', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e')})
    print(d['roofline'])
    print({k: (v['mean_us'], v['ms_per_image'], v['tflops'], v['gbs']) for k, v in d['unet_kernels'].items()})
except Exception as e:
    print('bench parse failed', sys.argv[1], e)
PY
done
( time timeout 400 python tools/r3_precision.py full ) > $O/precision_full.log 2>&1; grep "^{" $O/precision_full.log | cut -c1-1200; tail -2 $O/precision_full.log | grep -v "^{"
cp gpurun_out/r3_precision.json $O/ 2>/dev/null
( time timeout 420 python tools/vae_find.py enc_strips dec_full dec_tiles ) > $O/vae_find.log 2>&1; grep "^{" $O/vae_find.log; tail -3 $O/vae_find.log | grep -v "^{"
tar czf $O/miopen_cache.tgz miopen_cache; ls -la miopen_cache | head
du -sh $O
