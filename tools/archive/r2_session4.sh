#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
( time timeout 300 python tools/conv_immediate_diag.py ) > $O/conv_diag.log 2>&1; grep -E "^conv|Error" $O/conv_diag.log
( time timeout 300 python tools/r2_probe.py fused unet ) > $O/probe_nchw.log 2>&1; grep "^{" $O/probe_nchw.log
( time ED_CHANNELS_LAST=1 timeout 300 python tools/r2_probe.py unet table=20,6 ) > $O/probe_cl.log 2>&1; grep "^{" $O/probe_cl.log; tail -3 $O/probe_cl.log | cut -c1-300
( time timeout 600 python -m pytest -q -m gpu --timeout 600 -p no:cacheprovider -s tests/test_unet_kernels.py tests/test_real_arch_parity.py::test_fused_kernels_are_inside_the_bf16_loop ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "FAILED|Error|split vs single" $O/pytest.log | head -20
