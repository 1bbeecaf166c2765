#!/bin/bash
# Round-4 GPU session 17 (<1 GPU-minute, measurement only): hipBLASLt vs this repo's GEMM variants on the projections the shape policy
# leaves with the library, every arm timed as a hipGraph of 10 launches.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s17; mkdir -p $O
( time timeout 200 python tools/gemm_sched/lib_vs_ours.py ) > $O/lib_vs_ours.jsonl 2> $O/lib_vs_ours.err
cat $O/lib_vs_ours.jsonl; tail -3 $O/lib_vs_ours.err
