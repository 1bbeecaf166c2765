"""One eager SDXL UNet forward of each phase batch (40 and 12 rows by default, see ROWS) at the headline workload's shapes, for
`rocprofv3 --pmc` passes over the hand-written kernels inside the UNet (flash attention, GEGLU, GroupNorm, LayerNorm,
fused adds): the launch mix is exactly the per-timestep mix of bench.py's workload."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import ElasticDiffusion

# rows of the phase-A / phase-B forwards: 40,12 = bench.py's default (two images in flight: 2 x 20, 2 x 6); `python tools/pmc_unet.py 20,6` = one image
ROWS = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40,12").split(","))
pipe = ElasticDiffusion("cuda:0", "XL1.0", view_batch_size=16, use_graphs=False)
cfg = pipe.unet.config
with torch.no_grad():
    for rep in range(2):  # first repetition warms (kernel load, fused weights); rocprofv3 sees both, the summary skips it
        for rows in ROWS:
            x = torch.randn(rows, 4, 128, 128, device="cuda", dtype=pipe.model_dtype)
            txt = torch.randn(rows, 77, cfg.cross_attention_dim, device="cuda", dtype=pipe.model_dtype)
            pl = torch.randn(rows, cfg.pooled_projection_dim, device="cuda", dtype=pipe.model_dtype)
            pipe._forward_rows(x, torch.tensor(500, device="cuda"), txt, pl, None)
        torch.cuda.synchronize()
print("done")
