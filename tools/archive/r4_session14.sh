#!/bin/bash
# Round-4 GPU session 14 (~1.5 GPU-minutes, experiment only: nothing of the product changes): the persistent GEMM variants
# (tools/gemm_persist: v1 = next prologue after the last barrier pair, v2 = next tile's K tile 0 staged during the last K tile)
# against the product kernel.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s14; mkdir -p $O
( time timeout 240 python tools/gemm_persist/run.py --rounds 5 ) > $O/gemm_persist_ab.jsonl 2> $O/gemm_persist_ab.err
cat $O/gemm_persist_ab.jsonl; tail -3 $O/gemm_persist_ab.err
