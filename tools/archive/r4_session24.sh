#!/bin/bash
# Round-4 GPU session 24 (<1 GPU-minute, experiment only): the PRODUCT attention kernel with in-region loads / LDS writes
# (tools/attn16/attn5_inregion.hip, generated from csrc/attention_kernels.hip) against ed_flash_attention(v_path = 5).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s24; mkdir -p $O
( time timeout 100 python tools/attn16/run.py --product-inregion --rounds 5 ) > $O/attn5_inregion.jsonl 2> $O/attn5_inregion.err
cat $O/attn5_inregion.jsonl; tail -3 $O/attn5_inregion.err
