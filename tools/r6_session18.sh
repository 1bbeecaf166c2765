#!/bin/bash
# Round-6 GPU session 18 (~4 GPU-minutes): the in-situ policy search at the multi-GPU layout's per-rank forwards (3 rows, 1 row; 10 rows)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s18; mkdir -p $O
timeout 900 python tools/r6_policy_search.py --batches 3,1,10 > $O/policy_search_small.jsonl 2> $O/policy_search.err; cat $O/policy_search_small.jsonl | cut -c1-230; tail -3 $O/policy_search.err
