#!/bin/bash
# Round-4 GPU session 23 (<1 GPU-minute, experiment only): tools/attn16 -- accuracy (incl. the forced slow path) and timing of the plain
# variant, then the variant with the next tiles' global loads and LDS writes inside the MFMA region.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s23; mkdir -p $O
( time timeout 100 python tools/attn16/run.py --rounds 3 ) > $O/attn16.jsonl 2> $O/attn16.err
( timeout 100 python tools/attn16/run.py --inregion --rounds 5 ) > $O/attn16_inregion.jsonl 2>> $O/attn16.err
cat $O/attn16.jsonl $O/attn16_inregion.jsonl; tail -3 $O/attn16.err
