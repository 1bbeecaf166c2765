#!/bin/bash
# Round-4 GPU session 1 (prepared at the end of round 3; ~12 GPU-minutes): the three things round 3 left unmeasured.
#   1. experiments/geglu_gemm: first run on hardware -- correctness, race screen, A/B against hipBLASLt (+ ed_geglu)
#   2. VAE layout A/B: models.VAE_NCHW_RESIDUAL off / on over the four VAE shapes, each with its own MIOpen find
#   3. attention: pipelined (v_path 4) against lazy-maximum (v_path 5) inside the UNet forward, same process
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s1; mkdir -p $O
( time timeout 300 python experiments/geglu_gemm/run_geglu_gemm.py --dtype fp16 --out $O/geglu_gemm_fp16.json ) > $O/geglu_gemm_fp16.log 2>&1
tail -25 $O/geglu_gemm_fp16.log
( time timeout 200 python experiments/geglu_gemm/run_geglu_gemm.py --dtype bf16 --rounds 3 --out $O/geglu_gemm_bf16.json ) > $O/geglu_gemm_bf16.log 2>&1
tail -3 $O/geglu_gemm_bf16.log
( time timeout 400 python tools/vae_find.py ) > $O/vae_find_default.jsonl 2> $O/vae_find_default.err
( time timeout 400 python tools/vae_find.py --nchw ) > $O/vae_find_nchw.jsonl 2> $O/vae_find_nchw.err
cat $O/vae_find_default.jsonl $O/vae_find_nchw.jsonl
tar czf $O/miopen_cache.tgz miopen_cache
for v in 4 5 4 5; do
  ED_CL=1 ED_DTYPE=fp16 ED_FLASH_VARIANT=$v timeout 200 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2 | sed "s/^/v_path $v: /"
done > $O/attention_variant_in_unet.txt
cat $O/attention_variant_in_unet.txt
