#!/bin/bash
# Round-6 GPU session 5 (~8 GPU-minutes): where the batch-6 forward's time goes after the 128-row tiles and the new shape policy
# (rocprofv3 --kernel-trace --stats of eager forwards at batch 6 and 20), forward time at batch 20 / 6 / 3 / 1.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s5; mkdir -p $O
timeout 300 python tools/fwd_ab.py --libs product --batches 20,6,3,1 --modes fp16 > $O/fwd_policy.json 2> $O/fwd.err; cat $O/fwd_policy.json | cut -c1-200
for b in 6 20; do
  P=/tmp/prof_fwd$b; mkdir -p $P
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o fwd -- python $GRAFT_REPO_ROOT/tools/fwd_once.py $b sdxl > $P/run.log 2>&1)
  find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fwd_b${b}_kernel_stats.csv
  head -25 $O/fwd_b${b}_kernel_stats.csv | cut -c1-160
done
timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "wrappers or tile_height or channels_last" 2>&1 | tail -3
du -sh $O
