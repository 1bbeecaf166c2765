#!/bin/bash
# Round-4 GPU session 21 (<1 GPU-minute, experiment only): the 16x16x32 attention kernel (tools/attn16) against the product's v_path 5.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s21; mkdir -p $O
( time timeout 150 python tools/attn16/run.py --rounds 5 ) > $O/attn16.jsonl 2> $O/attn16.err
cat $O/attn16.jsonl; tail -5 $O/attn16.err
