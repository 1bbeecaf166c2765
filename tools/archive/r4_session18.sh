#!/bin/bash
# Round-4 GPU session 18 (~1.5 GPU-minutes, measurement only): the same replayed forward in fp16 and bf16, per-kernel totals of one
# replay -- which library kernels make fp16 4.5 % slower than bf16?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s18; mkdir -p $O
for B in 20 6; do
  for DT in fp16 bf16; do
    P=/tmp/fam_${DT}_b$B; rm -rf $P
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $P -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/fwd_graph_gaps.py run $B sdxl $DT ) > $O/traced_${DT}_b$B.json 2> $O/traced_${DT}_b$B.err
    tail -1 $O/traced_${DT}_b$B.json
    python tools/fwd_graph_gaps.py analyse $(find $P -name "*kernel_trace.csv" | head -1) > $O/kernels_${DT}_b$B.json 2>> $O/traced_${DT}_b$B.err
    rm -rf $P
  done
done
ls -la $O
