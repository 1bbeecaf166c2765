#!/bin/bash
# Round-6 GPU session 4 (~15 GPU-minutes): the 128-row tile mode (tile_phases_rows): tests in both tile heights, probe against 256-row tiles
# and the library at the under-filled shapes, forward A/B at batch 20 / 6 / 3 / 1.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s4; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "linear or conv3x3 or gemm or geglu" ) > $O/pytest_gemm.log 2>&1; tail -4 $O/pytest_gemm.log
timeout 400 python tools/r6_rows_probe.py --rounds 5 > $O/gemm_rows_mode.jsonl 2> $O/rows_probe.err; cat $O/gemm_rows_mode.jsonl | cut -c1-330; tail -2 $O/rows_probe.err
for b in "20,6" "3,1"; do
  ED_GEMM_ROWS=0 timeout 300 python tools/fwd_ab.py --libs product --batches $b --modes fp16 > $O/fwd_rows0_$b.json 2>> $O/fwd.err
  timeout 300 python tools/fwd_ab.py --libs product --batches $b --modes fp16 > $O/fwd_rowsauto_$b.json 2>> $O/fwd.err
  cat $O/fwd_rows0_$b.json $O/fwd_rowsauto_$b.json | cut -c1-200
done
du -sh $O
