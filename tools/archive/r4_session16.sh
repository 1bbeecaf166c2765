#!/bin/bash
# Round-4 GPU session 16 (~1.5 GPU-minutes, measurement only): idle time between the kernels of a hipGraph-replayed UNet forward
# (batch 20 and 6), un-profiled wall first, then a kernel trace of the same command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s16; mkdir -p $O
for B in 20 6; do
  timeout 150 python tools/fwd_graph_gaps.py run $B > $O/unprofiled_b$B.json 2> $O/unprofiled_b$B.err
  cat $O/unprofiled_b$B.json
  P=/tmp/gaps_b$B; rm -rf $P
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $P -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/fwd_graph_gaps.py run $B ) > $O/traced_b$B.json 2> $O/traced_b$B.err
  tail -1 $O/traced_b$B.json
  python tools/fwd_graph_gaps.py analyse $(find $P -name "*kernel_trace.csv" | head -1) > $O/gaps_b$B.json 2>> $O/traced_b$B.err
  cat $O/gaps_b$B.json | cut -c1-3000
done
tail -3 $O/traced_b6.err
