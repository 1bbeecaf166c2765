#!/bin/bash
# Round-3 GPU session 1: new tests (RCCL world-size-1 exchange, init_low, CLI, grid), precision evidence
# (tools/r3_precision.py), bf16 vs fp16 throughput with the derived fp16 conv records.  Results in gpurun_out/s1/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
( time timeout 600 python -m pytest -q -m gpu --timeout 500 -p no:cacheprovider -x tests/test_multiproc_gpu.py::test_rccl_world_size_one_exchange_path tests/test_hip_parity.py::test_verbose_init_low_matches_oracle tests/test_hip_parity.py::test_verbose_image_log tests/test_cli_gpu.py "tests/test_multiproc_gpu.py::test_sharded_ranks_reproduce_reference_latents" ) > $O/pytest_new.log 2>&1
tail -5 $O/pytest_new.log; grep -E "FAILED|Error" $O/pytest_new.log | head -20
( time timeout 900 python tools/r3_precision.py loop full long ) > $O/precision.log 2>&1
grep "^{" $O/precision.log | cut -c1-900; tail -3 $O/precision.log
cp gpurun_out/r3_precision.json $O/ 2>/dev/null
for dt in bf16 fp16; do
  ( time timeout 300 python bench.py --dtype $dt --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_$dt.json 2> $O/bench_$dt.err
  tail -2 $O/bench_$dt.err
  python - $dt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/s1/bench_{sys.argv[1]}.json') if l.startswith('{')][-1])
    print(sys.argv[1], {k: d.get(k) for k in ('value', 'ms_per_step', 'finite_output', 'graphs', 'phase_ms_last_image', 'roofline_e2e')})
except Exception as e:
    print('bench parse failed', sys.argv[1], e)
PY
done
( time timeout 200 python bench.py --force-exchange --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_force_exchange.json 2> $O/bench_force_exchange.err
tail -2 $O/bench_force_exchange.err; cut -c1-300 $O/bench_force_exchange.json | tail -1
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/s1/bench_force_exchange.json') if l.startswith('{')][-1])
    print('force-exchange', d['value'], d['rccl'], d['graphs'])
except Exception as e:
    print('parse failed', e)
PY
du -sh $O
