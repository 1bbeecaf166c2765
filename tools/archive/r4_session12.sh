#!/bin/bash
# Round-4 GPU session 12 (~2 GPU-minutes): GEMM main loop with early start + bias loads no longer waited for before staging,
# in-process A/B against the kernel the final suite ran on; parity tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s12; mkdir -p $O
( time timeout 300 python -m pytest tests/test_unet_kernels.py -x -q -k "geglu_gemm or linear_hip or conv3x3 or gemm_wrappers" ) > $O/pytest_gemm.log 2>&1
tail -3 $O/pytest_gemm.log
( time timeout 200 python tools/gemm_ab/run.py ) > $O/gemm_ab.jsonl 2> $O/gemm_ab.err
cat $O/gemm_ab.jsonl; tail -2 $O/gemm_ab.err
ED_DTYPE=fp16 timeout 150 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2 > $O/unet_forward.txt; cat $O/unet_forward.txt
