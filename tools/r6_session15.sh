#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s15; mkdir -p $O
timeout 300 python tools/r6_ops_ab.py --prev tools/ab/libelastic_hip_r6_s14.so --rounds 7 --only gn > $O/ops_ab_gn_reverse.jsonl 2> $O/ops_ab.err; cat $O/ops_ab_gn_reverse.jsonl | cut -c1-260; tail -2 $O/ops_ab.err
