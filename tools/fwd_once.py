"""Three eager SDXL UNet forwards at batch 20 (for `rocprofv3 --kernel-trace --stats`: where does one forward's time go?).
ED_CHANNELS_LAST=1 selects the channels-last path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.r2_probe import build_unet, inputs

unet, cfg = build_unet()
x, e, kw, t = inputs(cfg, int(sys.argv[1]) if len(sys.argv) > 1 else 20)
with torch.no_grad():
    for _ in range(3):
        unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)
torch.cuda.synchronize()
print("done")
