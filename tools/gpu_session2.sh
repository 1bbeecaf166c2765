#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ls -la miopen_cache | head -5
python -m pytest tests -m gpu -q -x 2>&1 | tail -5
ED_FUSED=0 python tools/probe_unet.py sdxl 20,6 2>&1 | tail -2
ED_FUSED=1 python tools/probe_unet.py sdxl 20,6 2>&1 | tail -2
python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r1b.json 2> gpurun_out/bench_r1b.err; tail -c 2500 gpurun_out/bench_r1b.json; tail -2 gpurun_out/bench_r1b.err
mkdir -p gpurun_out/prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench4 -- python $GRAFT_REPO_ROOT/bench.py --timesteps 4 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench/run.log 2>&1)
ls -la gpurun_out/prof_bench; find gpurun_out/prof_bench -name "*kernel_trace*" -size +30M -delete
tar czf gpurun_out/miopen_cache.tgz miopen_cache; du -sh miopen_cache
