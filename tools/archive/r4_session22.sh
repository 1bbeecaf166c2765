#!/bin/bash
# Round-4 GPU session 22 (<1 GPU-minute, measurement only): VALU issue rates (v_fma / v_exp / a softmax numerator, with and without MFMAs).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s22; mkdir -p $O
( time timeout 120 python tools/mfma_power/valu.py ) > $O/valu_rates.jsonl 2> $O/valu_rates.err
cat $O/valu_rates.jsonl; tail -3 $O/valu_rates.err
