#!/bin/bash
# one GPU session: profile the UNet, run the headline bench, save the MIOpen cache + profiles under gpurun_out/
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_unet gpurun_out/prof_bench
python tools/probe_unet.py sdxl 20,6 > gpurun_out/probe_nofind.log 2>&1; tail -3 gpurun_out/probe_nofind.log
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_unet -o unet_b20 -- python $GRAFT_REPO_ROOT/tools/probe_unet.py sdxl 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_unet/run.log 2>&1)
ls gpurun_out/prof_unet | head
python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; tail -c 3000 gpurun_out/bench_r1.json; tail -3 gpurun_out/bench_r1.err
du -sh miopen_cache; tar czf gpurun_out/miopen_cache.tgz miopen_cache; ls -la gpurun_out/
find gpurun_out/prof_unet -name "*.csv" | head; find gpurun_out/prof_unet -name "*kernel_trace*" -size +20M -delete
