#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s21; mkdir -p $O
P=/tmp/prof_vae; mkdir -p $P
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o vae -- python $GRAFT_REPO_ROOT/tools/r5_vae_ab.py --cases encode --reps 10 --switch VAE_SPLIT_DOWNSAMPLE > $P/run.log 2>&1)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/vae_encode_kernel_stats.csv
head -30 $O/vae_encode_kernel_stats.csv | cut -c1-170
