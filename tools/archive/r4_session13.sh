#!/bin/bash
# Round-4 GPU session 13 (~1.5 GPU-minutes, experiment only: nothing of the product changes): the persistent GEMM variant
# (tools/gemm_persist) against the product kernel.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s13; mkdir -p $O
( time timeout 200 python tools/gemm_persist/run.py ) > $O/gemm_persist_ab.jsonl 2> $O/gemm_persist_ab.err
cat $O/gemm_persist_ab.jsonl; tail -3 $O/gemm_persist_ab.err
