#!/bin/bash
# Round-5 last GPU action (~8.5 GPU-minutes): the complete GPU suite and smoke() once more on the exact final tree (since the second final
# session only bench.py strings, CPU-side tests and documents changed) -- the round's remaining GPU minutes would otherwise be lost.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5final4; mkdir -p $O
( time timeout 620 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
tail -4 $O/pytest_gpu_full.log
( time timeout 60 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1
grep smoke $O/smoke.log | tail -2
