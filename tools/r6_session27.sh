#!/bin/bash
# Round-6 GPU session 27: the complete GPU suite + smoke on the last tree (after the convolution batch split)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s27; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1; tail -4 $O/pytest_gpu_full.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; grep smoke $O/smoke.log
