#!/bin/bash
# Round-4 GPU session 15 (<1 GPU-minute, experiment only): where in a K tile the LDS-DMA pieces are issued, and a 4-interval loop
# (tools/gemm_sched) against the product kernel.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s15; mkdir -p $O
( time timeout 200 python tools/gemm_sched/run.py --rounds 5 ) > $O/gemm_sched_ab.jsonl 2> $O/gemm_sched_ab.err
cat $O/gemm_sched_ab.jsonl; tail -3 $O/gemm_sched_ab.err
