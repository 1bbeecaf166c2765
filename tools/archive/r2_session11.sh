#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
( time timeout 300 python bench.py --steps 1 --warmup 1 ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/s11/bench.json') if l.startswith('{')][-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'graphs', 'extras', 'roofline_e2e')})
print(d['roofline']['kernel'], d['roofline']['frac'], d.get('cpu_baseline', {}).get('value'), d.get('parity_bf16_rel_l2'))
PY
( time timeout 240 python -m pytest -q -m gpu --timeout 200 -p no:cacheprovider tests/test_multiproc_gpu.py -k rehearsal ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
