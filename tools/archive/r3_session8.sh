#!/bin/bash
# Round-3 GPU session 8: the exact multi-rank bench command with the FULL-SIZE fp16 model, 2 then 4 ranks sharing this box's
# one GPU over gloo (RCCL refuses two ranks per device): per-rank shapes 10 / 3, sharded pad-strip units, images in flight,
# rccl / latency fields.  Throughput numbers mean nothing here; what matters is that it runs, cold, and how long start-up takes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
for n in 2 4; do
  ( time ED_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout $((n*60+40)) python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29540+n)) bench.py --gpus $n --steps $((n/2)) --warmup 1 --no-kernel-timing ) > $O/bench_${n}rank_gloo.json 2> $O/bench_${n}rank_gloo.err
  grep -E "^real|Error|error" $O/bench_${n}rank_gloo.err | head -5
  python - $n <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/s8/bench_{sys.argv[1]}rank_gloo.json') if l.startswith('{')][-1])
    print(sys.argv[1], 'ranks:', {k: d.get(k) for k in ('value', 'ms_per_step', 'n_gpus', 'dtype', 'latency_s_per_image', 'rccl', 'finite_output', 'graphs', 'layouts', 'rows_computed_over_rows_total_rank0')})
    print(d['config']['parallelism'], d['extras'])
except Exception as e:
    print('parse failed', sys.argv[1], e)
PY
done
