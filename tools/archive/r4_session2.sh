#!/bin/bash
# Round-4 GPU session 2 (~14 GPU-minutes): the GEMM / convolution kernels as product entry points, attention v_path 6.
#   1. their parity tests + every attention variant's tests + the real-architecture drift gates with the new switches on
#   2. tools/probe_gemm.py: every Linear / 3x3-convolution shape of the SDXL (batch 20, 6) and SD1.5 (batch 20) forwards,
#      library call vs this repo's kernel
#   3. UNet forward A/B per switch (same process order baseline / each / all)
#   4. a short bench run (2 images)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s2; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_kernels.py tests/test_real_arch_parity.py -x -q -k "flash or gemm or linear_hip or conv3x3 or real_arch or fused_kernels or full_width or drift" ) > $O/pytest_kernels.log 2>&1
tail -15 $O/pytest_kernels.log
( time timeout 240 python tools/probe_gemm.py sdxl 20,6 --out $O/probe_gemm_sdxl.jsonl ) > $O/probe_gemm_sdxl.log 2>&1
tail -4 $O/probe_gemm_sdxl.log
( time timeout 150 python tools/probe_gemm.py sd15 20 --out $O/probe_gemm_sd15.jsonl ) > $O/probe_gemm_sd15.log 2>&1
tail -2 $O/probe_gemm_sd15.log
ALL=HIP_GEGLU_GEMM,HIP_LINEAR,HIP_CONV3X3
for cfg in "base:$ALL:4" "geglu:HIP_LINEAR,HIP_CONV3X3:4" "linear:HIP_GEGLU_GEMM,HIP_CONV3X3:4" "conv:HIP_GEGLU_GEMM,HIP_LINEAR:4" "all_v4::4" "all_v5::5" "all_v6::6" "base:$ALL:4" "all_v6::6"; do
  IFS=: read name dis var <<< "$cfg"
  ED_DTYPE=fp16 ED_DISABLE=$dis ED_FLASH_VARIANT=$var timeout 150 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2 | sed "s/^/$name: /"
done > $O/unet_forward_ab.txt
cat $O/unet_forward_ab.txt
( time timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_2img.json 2> $O/bench_2img.err
tail -c 1500 $O/bench_2img.json; tail -5 $O/bench_2img.err
