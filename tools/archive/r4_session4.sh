#!/bin/bash
# Round-4 GPU session 4 (~10 GPU-minutes): attention with the LDS operand reads three MFMAs ahead (v_path 7), the other
# workloads' lines with the round-4 kernels (cfg2 under rocprofv3 --kernel-trace --stats, cfg4, cfg5).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s4; mkdir -p $O
( time timeout 300 python -m pytest tests/test_unet_kernels.py -x -q -k "flash" ) > $O/pytest_flash.log 2>&1
tail -4 $O/pytest_flash.log
for v in 5 7 4 7 5; do
  ED_DTYPE=fp16 ED_FLASH_VARIANT=$v timeout 150 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2 | sed "s/^/v_path $v: /"
done > $O/attention_depth_in_unet.txt
cat $O/attention_depth_in_unet.txt
( time timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_cfg2 -o cfg2 -- python bench.py --workload sd15_512x1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
tail -c 400 $O/bench_cfg2.json; ls $O/prof_cfg2 | head
( time timeout 400 python bench.py --workload sdxl_2048x2048_tiled --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err
tail -c 300 $O/bench_cfg4.json; tail -2 $O/bench_cfg4.err
( time timeout 300 python bench.py --workload sdxl_1024x2048_controlnet --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -c 300 $O/bench_cfg5.json; tail -2 $O/bench_cfg5.err
find $O/prof_cfg2 -name "*kernel_stats.csv" | head -2
