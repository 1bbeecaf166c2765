"""Three eager SDXL UNet forwards at batch 20 (for `rocprofv3 --kernel-trace --stats`: where does one forward's time go?).
ED_CHANNELS_LAST=1 selects the channels-last path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.r2_probe import build_unet, inputs

from elasticdiffusion_official_amd import models as M

fam = sys.argv[2] if len(sys.argv) > 2 else "sdxl"
if fam == "sdxl":
    unet, cfg = build_unet()
    x, e, kw, t = inputs(cfg, int(sys.argv[1]) if len(sys.argv) > 1 else 20)
else:
    cfg = M.UNET_CONFIGS[fam]
    torch.manual_seed(0)
    unet = M.UNet2DConditionModel(**cfg).to("cuda", torch.bfloat16).eval().requires_grad_(False)
    if M.CHANNELS_LAST:
        unet = unet.to(memory_format=torch.channels_last)
    B, S = int(sys.argv[1]), cfg["sample_size"]
    x = torch.randn(B, 4, S, S, device="cuda", dtype=torch.bfloat16)
    e = torch.randn(B, 77, cfg["cross_attention_dim"], device="cuda", dtype=torch.bfloat16)
    kw, t = None, torch.tensor(500, device="cuda")
with torch.no_grad():
    for _ in range(3):
        unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)
torch.cuda.synchronize()
print("done")
