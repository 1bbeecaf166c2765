#!/bin/bash
# Round-2 GPU session 3: tests changed since session 1 (fused glue, interleave, generate/verbose, bias folding, bench
# rehearsals), UNet A/B with the new switches, TunableOp experiment.  Results in gpurun_out/s3/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
( time timeout 900 python -m pytest -q -m gpu --timeout 600 -p no:cacheprovider -s tests/test_unet_kernels.py tests/test_hip_parity.py tests/test_real_arch_parity.py ) > $O/pytest_a.log 2>&1
tail -4 $O/pytest_a.log; grep -E "FAILED|Error" $O/pytest_a.log | head -20
( time timeout 900 python -m pytest -q -m gpu --timeout 600 -p no:cacheprovider -s tests/test_multiproc_gpu.py ) > $O/pytest_b.log 2>&1
tail -4 $O/pytest_b.log; grep -E "FAILED|Error" $O/pytest_b.log | head -20
( time timeout 300 python tools/r2_probe.py unet ) > $O/probe.log 2>&1
grep "^{" $O/probe.log
( time timeout 600 python tools/tune_gemms.py 20,6 8 4 ) > $O/tune.log 2>&1
grep -E "^\{|files|Error|error" $O/tune.log | head; tail -2 $O/tune.log
mkdir -p $O/tunableop_cache; cp tunableop_cache/* $O/tunableop_cache/ 2>/dev/null; ls -la $O/tunableop_cache
