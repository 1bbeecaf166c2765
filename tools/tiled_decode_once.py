"""One cfg4 tiled decode (SDXL 2048x2048 latent 256x256 -> 64 tiles of 1024^2 px, fp32 VAE, tile batch 8) for
`rocprofv3 --kernel-trace --stats`: where the 8 s of the 30 s cfg4 image go."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import ElasticDiffusion, models as M

_, vae = M.build_models("XL1.0", device="cuda:0")[:2]


class _NoUNet(torch.nn.Module):  # the decode needs only the UNet's config
    def __init__(self):
        super().__init__()
        self.config = type("C", (), dict(sample_size=128, in_channels=4, cross_attention_dim=2048, pooled_projection_dim=1280))()
        self.p = torch.nn.Parameter(torch.zeros(1))


pipe = ElasticDiffusion("cuda:0", "XL1.0", unet=_NoUNet(), vae=vae)
z = torch.randn(1, 4, 256, 256, device="cuda:0")
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = pipe.tiled_decode(z)
    torch.cuda.synchronize()
    print(f"tiled decode {rep}: {time.perf_counter() - t0:.3f} s", tuple(img.shape), flush=True)
