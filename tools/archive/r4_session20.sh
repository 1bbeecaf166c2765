#!/bin/bash
# Round-4 GPU session 20 (<1 GPU-minute, measurement only): what the chip sustains on bare 16-bit MFMAs by shape / type / operand bits.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s20; mkdir -p $O
( time timeout 120 python tools/mfma_power/run.py ) > $O/mfma_power.jsonl 2> $O/mfma_power.err
cat $O/mfma_power.jsonl; tail -3 $O/mfma_power.err
