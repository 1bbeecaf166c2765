#!/bin/bash
# Round-4 GPU session 19 (<1 GPU-minute, measurement only): fp16 vs bf16 operands through the same MFMA kernels (operand bit activity vs code path).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s19; mkdir -p $O
( time timeout 200 python tools/gemm_sched/fp16_vs_bf16_data.py ) > $O/fp16_vs_bf16_data.jsonl 2> $O/fp16_vs_bf16_data.err
cat $O/fp16_vs_bf16_data.jsonl; tail -3 $O/fp16_vs_bf16_data.err
