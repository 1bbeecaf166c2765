#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_unet_kernels.py tests/test_multiproc_gpu.py -m gpu -q 2>&1 | tail -3
ED_SCGEMM=0 python tools/probe_unet.py sdxl 20,6 2>&1 | tail -2
ED_SCGEMM=1 python tools/probe_unet.py sdxl 20,6,10,3 2>&1 | tail -4
timeout 500 python tools/run_configs.py cfg2,cfg4,cfg5 3 2>&1 | grep -v amdgpu.ids | tail -8
tar czf gpurun_out/miopen_cache.tgz miopen_cache; du -sh miopen_cache
