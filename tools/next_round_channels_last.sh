#!/bin/bash
# Next-round experiment (needs ~15 GPU-minutes): populate MIOpen's find-db for the channels-last convolution shapes,
# then A/B the UNet forward with models.CHANNELS_LAST on/off.  If the batch-20 forward is faster, flip the default in
# elasticdiffusion_official_amd/models.py and re-run bench.py.
set -x
cd $GRAFT_REPO_ROOT
for b in 20 6 10 3; do
  ( time ED_CL=1 timeout 600 python tools/probe_unet.py sdxl $b find ) 2>&1 | grep -v amdgpu.ids | tail -3
  tar czf gpurun_out/miopen_cache.tgz miopen_cache   # keep what has been found so far
done
ED_CL=0 python tools/probe_unet.py sdxl 20,6 2>&1 | tail -2
ED_CL=1 python tools/probe_unet.py sdxl 20,6 2>&1 | tail -2
