#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s19; mkdir -p $O
timeout 300 python tools/r6_ops_ab.py --prev tools/ab/libelastic_hip_r6_s18.so --rounds 7 --only sd1,attn > $O/ops_ab_sd1_store.jsonl 2> $O/ops_ab.err; cat $O/ops_ab_sd1_store.jsonl | cut -c1-220; tail -2 $O/ops_ab.err
( time timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "attention or flash" ) > $O/pytest_attn.log 2>&1; tail -4 $O/pytest_attn.log
