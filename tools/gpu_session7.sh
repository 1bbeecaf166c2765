#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
grep -n "Fatal\|Error\|error\|fault\|FAILED\|passed\|failed\|Memory access" gpurun_out/pytest_gpu.log | head -20
grep -n "File \"/root/repo\|File \"/tmp" gpurun_out/pytest_gpu.log | head -20
head -c 1500 gpurun_out/pytest_gpu.log
