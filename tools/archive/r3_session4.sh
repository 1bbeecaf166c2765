#!/bin/bash
# Round-3 GPU session 4: the N = 8 layout rehearsed cold with the full-size model (8 ranks share this box's one GPU over
# gloo), the other BASELINE.json workloads (cfg5 ControlNet, cfg2, cfg4 + rocprof of its tiled decode), and the rest of the
# -m gpu suite with the final code.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
( time timeout 300 python -m pytest -q -m gpu --timeout 250 -p no:cacheprovider tests/test_unet_kernels.py -k "groupnorm_f32 or fp32_vae or flash or softmax or vae_attention" ) > $O/pytest_kernels.log 2>&1
tail -3 $O/pytest_kernels.log; grep -E "^FAILED|^ERROR" $O/pytest_kernels.log | head
( time ED_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 2 --warmup 1 ) > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err
grep -v "amdgpu.ids\|Gloo\|socket" $O/bench_8rank_gloo.err | tail -6
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/s4/bench_8rank_gloo.json') if l.startswith('{')][-1])
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'n_gpus', 'dtype', 'latency_s_per_image', 'rccl', 'finite_output', 'graphs', 'layouts', 'extras', 'rows_computed_over_rows_total_rank0')})
    print(d['config']['parallelism'])
except Exception as e:
    print('8-rank parse failed', e)
PY
for wl in sdxl_1024x2048_controlnet sd15_512x1024 sdxl_2048x2048_tiled; do
  ( time timeout 400 python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_$wl.json 2> $O/bench_$wl.err; grep -v "amdgpu.ids" $O/bench_$wl.err | tail -2
  python - $wl <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/s4/bench_{sys.argv[1]}.json') if l.startswith('{')][-1])
    print(sys.argv[1], {k: d.get(k) for k in ('value', 'ms_per_step', 'dtype', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e')})
    print(d['roofline'])
except Exception as e:
    print('bench parse failed', sys.argv[1], e)
PY
done
mkdir -p $O/prof_tiled
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_tiled -o tiled -- python $GRAFT_REPO_ROOT/tools/tiled_decode_once.py > $GRAFT_REPO_ROOT/$O/prof_tiled/run.log 2>&1)
tail -2 $O/prof_tiled/run.log
find $O/prof_tiled -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tiled_decode_kernel_stats.csv
find $O/prof_tiled -name "*kernel_trace.csv" -delete
head -14 $O/tiled_decode_kernel_stats.csv | cut -c1-170
( time timeout 1100 python -m pytest -q -m gpu --timeout 900 -p no:cacheprovider --durations=12 tests/test_hip_parity.py tests/test_models_and_text.py tests/test_multiproc_gpu.py tests/test_real_arch_parity.py tests/test_cli_gpu.py ) > $O/pytest_rest.log 2>&1
tail -22 $O/pytest_rest.log; grep -E "^FAILED|^ERROR" $O/pytest_rest.log | head -20
tar czf $O/miopen_cache.tgz miopen_cache; ls -la miopen_cache | head -8
du -sh $O
