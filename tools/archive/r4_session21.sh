#!/bin/bash
# Round-4 GPU session 21 (<1 GPU-minute per call, experiment only): the 16x16x32 attention kernel (tools/attn16) against the product's
# v_path 5: accuracy + timing, then the s_memtime build's per-segment cycles, then the no-check ablation.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s21; mkdir -p $O
( time timeout 100 python tools/attn16/run.py --rounds 5 ) > $O/attn16.jsonl 2> $O/attn16.err
( timeout 60 python tools/attn16/run.py --segments ) > $O/attn16_segments.jsonl 2>> $O/attn16.err
( timeout 60 python tools/attn16/run.py --nocheck ) > $O/attn16_nocheck.jsonl 2>> $O/attn16.err
cat $O/attn16.jsonl $O/attn16_segments.jsonl $O/attn16_nocheck.jsonl; tail -3 $O/attn16.err
