#!/bin/bash
# Round-6 GPU session 8 (~8 GPU-minutes): in-situ search of the shape policy at the forwards the two-images-in-flight default runs (40 and 12 rows),
# and the forward time by batch (40 / 20 / 12 / 6).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s8; mkdir -p $O
timeout 700 python tools/r6_policy_search.py --batches 12,40 > $O/policy_search.jsonl 2> $O/policy_search.err; cat $O/policy_search.jsonl | cut -c1-260; tail -3 $O/policy_search.err
timeout 300 python tools/fwd_ab.py --libs product --batches 40,20,12,6 --modes fp16 > $O/fwd_by_batch.jsonl 2> $O/fwd.err; cat $O/fwd_by_batch.jsonl | cut -c1-200
