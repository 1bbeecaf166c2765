"""GPU probe: ed_flash_attention variants at the SDXL self-attention shapes (q / k / v = column slices of a fused projection,
as inside the UNet), interleaved rounds, median.   python tools/probe_attention.py [--rounds 5] [--variants 4,5,9,10]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from elasticdiffusion_official_amd import ops


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="4,5,7,9,10,2")
a = ap.parse_args()
variants = [int(v) for v in a.variants.split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
for (B, H, N) in [(20, 10, 4096), (20, 20, 1024), (6, 10, 4096), (6, 20, 1024), (10, 10, 4096), (3, 20, 1024)]:
    qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(torch.float16)
    q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
    t = {vp: [] for vp in variants}
    for _ in range(a.rounds):
        for vp in variants:
            t[vp].append(timed(lambda: ops.flash_attention(q, k, v, H, v_path=vp)))
    flops = 4.0 * B * H * N * N * 64
    med = {vp: sorted(x)[len(x) // 2] for vp, x in t.items()}
    print(json.dumps({"B": B, "H": H, "N": N, "us": {vp: round(1e3 * m, 1) for vp, m in med.items()},
                      "tflops": {vp: round(flops / m / 1e9, 1) for vp, m in med.items()}}), flush=True)
