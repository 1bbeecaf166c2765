#!/bin/bash
# Round-6 GPU session 7 (~10 GPU-minutes): (a) which half of the round-6 shape policy costs the batch-3 forward 4 % (projections or convolutions),
# (b) images in flight at N = 1: 2 / 3 / 4 (their pending model calls fused into 40+12, 60+18, 80+24-row forwards) against the one-image headline.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s7; mkdir -p $O
timeout 400 python tools/r6_policy_ab.py --batches 3,6,1 --arms r5,r6lin,r6conv,r6 > $O/policy_split.jsonl 2> $O/policy_split.err; cat $O/policy_split.jsonl; tail -2 $O/policy_split.err
for m in 2 3 4; do
  timeout 500 python bench.py --in-flight $m --steps $((2*m)) --warmup $m --no-extras --fp32-leg off --no-kernel-timing > $O/bench_inflight$m.json 2> $O/bench_inflight$m.err
  python - <<P
import json
try:
    d=json.loads(open("$O/bench_inflight$m.json").read().strip().splitlines()[-1])
    print("in_flight", $m, d["value"], d["ms_per_step"], d.get("latency_s_per_image"))
except Exception as e:
    print("in_flight", $m, "failed", e)
P
done
