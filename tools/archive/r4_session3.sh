#!/bin/bash
# Round-4 GPU session 3 (~14 GPU-minutes): the test files session 2's -x run did not reach, the multi-rank rehearsals, the fp32 line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s3; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_kernels.py tests/test_real_arch_parity.py tests/test_multiproc_gpu.py -x -q ) > $O/pytest_kernels_realarch_multiproc.log 2>&1
tail -8 $O/pytest_kernels_realarch_multiproc.log
# the driver's 8-GPU command with the full-size fp16 model, eight ranks sharing this one GPU over gloo, cold
( time ED_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29548 bench.py --gpus 8 --steps 4 --warmup 1 --no-kernel-timing --no-extras ) > $O/bench_8rank_gloo_fullsize.json 2> $O/bench_8rank_gloo_fullsize.err
tail -c 1200 $O/bench_8rank_gloo_fullsize.json; tail -4 $O/bench_8rank_gloo_fullsize.err
# what BASELINE.json's own tolerance costs: the same workload with the fp32 UNet (plain torch ops), one timed image
( time MIOPEN_FIND_MODE=FAST timeout 500 python bench.py --dtype fp32 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_fp32.json 2> $O/bench_fp32.err
tail -c 600 $O/bench_fp32.json; tail -3 $O/bench_fp32.err
