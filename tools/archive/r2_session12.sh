#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
( time timeout 150 python -m pytest -q -m gpu --timeout 120 -p no:cacheprovider tests/test_hip_parity.py -k "end_to_end_vs_oracle or generate_vs or verbose or interleaved" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log; grep -E "FAILED" $O/pytest.log | head -5
for a in 1 0; do
  ( ED_ASYNC_STRIPS=$a timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_async$a.json 2> $O/bench_async$a.err
  python - $a <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/s12/bench_async{sys.argv[1]}.json') if l.startswith('{')][-1])
    print('async', sys.argv[1], d['value'], d['ms_per_step'], d['phase_ms_last_image'], d['finite_output'])
except Exception as e:
    print('parse failed', sys.argv[1], e)
PY
done
