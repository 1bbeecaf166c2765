#!/bin/bash
# Round-6 GPU session 9 (~12 GPU-minutes): the three round-6 fusions (skip concatenation never written, upsampler convolution gathering from
# the source, transformer closing add in the projection epilogue): parity tests, then A/B in the forward at 40 / 12 / 20 / 6 rows.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s9; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "concatenation or upsampl or wrappers or channels_last or conv3x3 or linear_hip" ) > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 500 python tools/r6_switch_ab.py --batches 40,12,20,6 > $O/switch_ab.jsonl 2> $O/switch_ab.err; cat $O/switch_ab.jsonl; tail -3 $O/switch_ab.err
( time timeout 900 python -m pytest tests/test_unet_kernels.py tests/test_models_and_text.py tests/test_real_arch_parity.py -m gpu -x -q ) > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
