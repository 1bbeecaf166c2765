"""Which solver does MIOpen pick in IMMEDIATE mode (no benchmark) with the in-tree db, NCHW vs channels-last?
Times a few SDXL convolutions at batch 20 both ways (bias-free, so only the convolution and its layout kernels run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401  (points MIOpen at miopen_cache/)
from tools.r2_probe import ev_time

for (cin, cout, hw, stride) in [(320, 320, 128, 1), (640, 640, 64, 1), (1280, 1280, 32, 1), (1920, 640, 64, 1),
                                (960, 320, 128, 1), (320, 320, 128, 2), (2560, 1280, 32, 1)]:
    w = torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.bfloat16) * 0.02
    x = torch.randn(20, cin, hw, hw, device="cuda", dtype=torch.bfloat16)
    t_nchw = ev_time(lambda: F.conv2d(x, w, None, stride=stride, padding=1), reps=10, warm=3)
    xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    t_cl = ev_time(lambda: F.conv2d(xc, wc, None, stride=stride, padding=1), reps=10, warm=3)
    flops = 2.0 * 9 * cin * cout * 20 * (hw // stride) ** 2
    print(f"conv {cin}->{cout} @{hw} s{stride}: NCHW {t_nchw:8.1f} us ({flops / t_nchw / 1e6:6.0f} TF)   "
          f"channels_last {t_cl:8.1f} us ({flops / t_cl / 1e6:6.0f} TF)", flush=True)
