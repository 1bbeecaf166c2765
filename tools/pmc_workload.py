"""Launch mix of one of bench.py's workloads for `rocprofv3 --pmc` passes: the workload's own pipeline (same classes, shapes, view batches,
ControlNet) runs TWO denoising timesteps eagerly (no hipGraph: counters are attributed per kernel launch); tools/pmc_summarise.py averages the
second half of the launches of every entry point, i.e. the second timestep.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o w -- python tools/pmc_workload.py sdxl_2048x2048_tiled"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench  # noqa: E402
import elasticdiffusion_official_amd  # noqa: F401,E402
from elasticdiffusion_official_amd import ElasticDiffusion  # noqa: E402

name = sys.argv[1]
wl = bench.WORKLOADS[name]
common = dict(view_batch_size=wl["vbs"], use_graphs=False)
if wl.get("controlnet") is not None:
    from elasticdiffusion_official_amd import ElasticDiffusionControlNet
    pipe = ElasticDiffusionControlNet("cuda:0", wl["sd"], "depth", **common)
else:
    pipe = ElasticDiffusion("cuda:0", wl["sd"], **common)
kw = dict(height=wl["H"], width=wl["W"], num_inference_steps=2, guidance_scale=wl["guidance"], resampling_steps=wl["R"], new_p=wl["new_p"],
          rrg_stop_t=wl["rrg_stop_t"], rrg_init_weight=wl["rrg_w"], cosine_scale=wl["cosine_scale"], repaint_sampling=True)
if wl.get("controlnet") is not None:
    hh, ww = pipe.get_downsample_size(wl["H"], wl["W"])
    yy = torch.linspace(0, 1, hh * 8).view(1, 1, -1, 1).expand(1, 1, hh * 8, ww * 8)
    xx = torch.linspace(0, 1, ww * 8).view(1, 1, 1, -1).expand(1, 1, hh * 8, ww * 8)
    kw.update(condition_image=torch.cat([yy, xx, 0.5 * (yy + xx)], dim=1).contiguous(), controlnet_conditioning_scale=wl["controlnet"])
pipe.seed_everything(0)
with torch.no_grad():
    pipe.generate_latents("An astronaut riding a corgi on the moon", "blurry", **kw)
torch.cuda.synchronize()
print("done")
