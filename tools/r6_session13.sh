#!/bin/bash
# Round-6 GPU session 13 (~6 GPU-minutes): attention output rows stored as 4 dwordx4 per lane (two lanes of a row swap halves) instead of
# 8 dwordx2: ops-level A/B against the session-12 library (bit identity + time), the attention parity tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s13; mkdir -p $O
timeout 400 python tools/r6_ops_ab.py --prev tools/ab/libelastic_hip_r6_s12.so --rounds 7 --only attn,attn40,xattn > $O/ops_ab_store.jsonl 2> $O/ops_ab.err; cat $O/ops_ab_store.jsonl | cut -c1-220; tail -2 $O/ops_ab.err
( time timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "attention or flash" ) > $O/pytest_attn.log 2>&1; tail -4 $O/pytest_attn.log
