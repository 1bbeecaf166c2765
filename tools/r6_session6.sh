#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s6; mkdir -p $O
timeout 600 python tools/r6_policy_ab.py --batches 20,6,3,1 > $O/policy_ab.jsonl 2> $O/policy_ab.err; cat $O/policy_ab.jsonl; tail -2 $O/policy_ab.err
