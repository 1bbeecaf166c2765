#!/bin/bash
# Round-6 GPU session 22 (~4 GPU-minutes): LayerNorm / add-LayerNorm walking their rows back to front, judged in the forward (in-situ A/B of two builds)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s22; mkdir -p $O
timeout 500 python tools/fwd_ab.py --libs tools/ab/libelastic_hip_r6_final2.so,product --batches 40,12,20,6 --modes fp16 > $O/fwd_ab_ln_reverse.jsonl 2> $O/fwd_ab.err; cat $O/fwd_ab_ln_reverse.jsonl | cut -c1-260; tail -2 $O/fwd_ab.err
