#!/bin/bash
# Round-6 GPU session 10 (~5 GPU-minutes): the stride-2 Downsample2D kernel (ed_conv3x3_nhwc_s2): parity, then A/B in the forward.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s10; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "downsampler or upsampl or wrappers" ) > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 400 python tools/r6_switch_ab.py --batches 40,12,20,6 --switches HIP_DOWNSAMPLE_CONV > $O/switch_ab_s2.jsonl 2> $O/switch_ab.err; cat $O/switch_ab_s2.jsonl; tail -3 $O/switch_ab.err
