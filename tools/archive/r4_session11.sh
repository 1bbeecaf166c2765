#!/bin/bash
# Round-4 GPU session 11 (~3 GPU-minutes): "early start" of the GEMM main loop (first K tile after 4 of 14 prologue DMAs):
# parity tests, in-process A/B against the committed kernel, the UNet forward.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s11; mkdir -p $O
( time timeout 300 python -m pytest tests/test_unet_kernels.py -x -q -k "geglu_gemm or linear_hip or conv3x3 or gemm_wrappers" ) > $O/pytest_gemm.log 2>&1
tail -4 $O/pytest_gemm.log
( time timeout 200 python tools/gemm_ab/run.py ) > $O/gemm_early_start_ab.jsonl 2> $O/gemm_early_start_ab.err
cat $O/gemm_early_start_ab.jsonl; tail -2 $O/gemm_early_start_ab.err
for i in 1 2; do
  ED_DTYPE=fp16 timeout 150 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2
done > $O/unet_forward_early_start.txt
cat $O/unet_forward_early_start.txt
