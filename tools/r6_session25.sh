#!/bin/bash
# Round-6 GPU session 25 (~5 GPU-minutes): convolution batch split (tail of a badly quantised grid as 128-row tiles): bit identity, op-level and in-forward A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s25; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "batch_split or conv3x3_nhwc or tile_height" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
cp elasticdiffusion_official_amd/libelastic_hip.so /tmp/same.so
timeout 300 python tools/r6_ops_ab.py --prev /tmp/same.so --rounds 7 --only convsplit > $O/ops_ab_convsplit.jsonl 2> $O/ops_ab.err; cat $O/ops_ab_convsplit.jsonl | cut -c1-220; tail -2 $O/ops_ab.err
timeout 400 python tools/r6_switch_ab.py --batches 40,12,6 --switches ops.CONV_BATCH_SPLIT > $O/switch_ab_convsplit.jsonl 2> $O/switch_ab.err; cat $O/switch_ab_convsplit.jsonl; tail -2 $O/switch_ab.err
