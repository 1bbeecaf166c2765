#!/bin/bash
# Round-6 GPU session 11 (~6 GPU-minutes): residual adds inside the projections (FUSED_RESIDUAL_LINEAR): test, A/B in the forward; what the
# library charges for the residual as its C operand (the K >= 1280 projections).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s11; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_kernels.py -m gpu -x -q -k "residual_adds or channels_last_path or wrappers" ) > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 300 python tools/r6_addmm_probe.py > $O/addmm_probe.jsonl 2> $O/addmm.err; cat $O/addmm_probe.jsonl; tail -2 $O/addmm.err
timeout 400 python tools/r6_switch_ab.py --batches 40,12,20,6 --switches FUSED_RESIDUAL_LINEAR > $O/switch_ab_res.jsonl 2> $O/switch_ab.err; cat $O/switch_ab_res.jsonl; tail -3 $O/switch_ab.err
