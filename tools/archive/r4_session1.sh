#!/bin/bash
# Round-4 GPU session 1 (~22 GPU-minutes): the full GPU suite first, then the three things round 3
# left unmeasured.
#   0. pytest -m gpu -x -q over the whole tree (the run the driver repeats at round end)
#   1. experiments/geglu_gemm: first run on hardware -- correctness, race screen, A/B against hipBLASLt (+ ed_geglu)
#   2. attention: pipelined (v_path 4) against lazy-maximum (v_path 5) inside the UNet forward, same process order 4 5 4 5
#   3. VAE layout A/B: models.VAE_NCHW_RESIDUAL off / on over the VAE shapes, each with its own MIOpen find
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s1; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -x -q ) > $O/pytest_full.log 2>&1
tail -6 $O/pytest_full.log
( time timeout 300 python experiments/geglu_gemm/run_geglu_gemm.py --dtype fp16 --out $O/geglu_gemm_fp16.json ) > $O/geglu_gemm_fp16.log 2>&1
tail -40 $O/geglu_gemm_fp16.log
for v in 4 5 4 5; do
  ED_CL=1 ED_DTYPE=fp16 ED_FLASH_VARIANT=$v timeout 150 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2 | sed "s/^/v_path $v: /"
done > $O/attention_variant_in_unet.txt
cat $O/attention_variant_in_unet.txt
( time timeout 150 python tools/vae_find.py dec_tiles enc_strips dec_full ) > $O/vae_find_default.jsonl 2> $O/vae_find_default.err
( time timeout 330 python tools/vae_find.py --nchw dec_tiles enc_strips dec_full ) > $O/vae_find_nchw.jsonl 2> $O/vae_find_nchw.err
cat $O/vae_find_default.jsonl $O/vae_find_nchw.jsonl
tail -3 $O/vae_find_nchw.err
tar czf $O/miopen_cache.tgz miopen_cache
