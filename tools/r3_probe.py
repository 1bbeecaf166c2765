"""GPU probe for round 3 (plumbing measurement, not a test).  One process, results as JSON lines on stdout:

  attn     ed_flash_attention variants (0 legacy, 2 legacy 64-row, 4 pipelined, 8 small-KV) vs SDPA at the SDXL shapes
  gn       ed_groupnorm_nhwc (2 launches) and ed_bias_residual_add (channels-last) at the UNet's shapes
  vae      ops.vae_attention (fp32 GEMM + ed_softmax_rows + GEMM) vs SDPA at the VAE's token counts
  pmcattn  launches each self-attention variant a few times (run under rocprofv3 --pmc ...)

usage: r3_probe.py [attn] [gn] [vae] [pmcattn]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401  (sets the MIOpen cache location)
from elasticdiffusion_official_amd import ops

DEV = "cuda:0"


def ev_time(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def emit(**kw):
    print(json.dumps(kw), flush=True)


def probe_attn():
    for (B, H, Nq, Nk) in [(20, 10, 4096, 4096), (20, 20, 1024, 1024), (6, 10, 4096, 4096), (6, 20, 1024, 1024),
                           (20, 10, 4096, 77), (20, 20, 1024, 77), (6, 20, 1024, 77), (10, 10, 4096, 4096), (3, 20, 1024, 1024)]:
        q, k, v = (torch.randn(B, n, H * 64, device=DEV).to(torch.bfloat16) for n in (Nq, Nk, Nk))
        q4, k4, v4 = (t.view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
        flops = 4.0 * B * H * Nq * Nk * 64
        nbytes = 2.0 * 2 * H * 64 * B * (Nq + Nk)
        t_sdpa = ev_time(lambda: F.scaled_dot_product_attention(q4, k4, v4))
        res = {"B": B, "H": H, "Nq": Nq, "Nk": Nk, "sdpa_us": round(t_sdpa, 1), "sdpa_tflops": round(flops / t_sdpa / 1e6, 1)}
        ref = ops.flash_attention(q, k, v, H, v_path=0).float()
        for path in ((0, 8) if Nk <= 96 else (0, 2, 4, 5)):
            t = ev_time(lambda: ops.flash_attention(q, k, v, H, v_path=path))
            res[f"v{path}_us"] = round(t, 1)
            res[f"v{path}_tflops"] = round(flops / t / 1e6, 1)
            res[f"v{path}_gbs"] = round(nbytes / t / 1e3, 1)
            res[f"v{path}_maxdiff_vs_v0"] = float((ops.flash_attention(q, k, v, H, v_path=path).float() - ref).abs().max())
        emit(probe="attn", **res)


def probe_gn():
    for shape in [(20, 320, 128, 128), (20, 960, 128, 128), (20, 640, 64, 64), (20, 1280, 32, 32), (20, 1920, 64, 64), (6, 320, 128, 128),
                  (6, 640, 64, 64)]:
        x = torch.randn(*shape, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        r = torch.randn(*shape, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        C = shape[1]
        w, bb = torch.ones(C, device=DEV, dtype=torch.bfloat16), torch.zeros(C, device=DEV, dtype=torch.bfloat16)
        cb = torch.randn(shape[0], C, device=DEV).to(torch.bfloat16)
        t = ev_time(lambda: ops.groupnorm_nhwc(x, w, bb, 32, 1e-5, silu=True, chan_bias=cb, conv_bias=bb))
        t2 = ev_time(lambda: ops.bias_residual_add(x, bb, r, None))
        emit(probe="gn_nhwc", shape=list(shape), gn_us=round(t, 1), gn_gbs=round(3 * x.numel() * 2 / t / 1e3, 1),
             bra_us=round(t2, 1), bra_gbs=round(3 * x.numel() * 2 / t2 / 1e3, 1))


def probe_vae():
    for (B, N) in [(1, 32768), (5, 4096), (8, 16384)]:
        q, k, v = (torch.randn(B, N, 512, device=DEV) for _ in range(3))
        t = ev_time(lambda: ops.vae_attention(q, k, v), reps=3, warm=1)
        ts = ev_time(lambda: F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)), reps=3, warm=1)
        flops = 4.0 * B * N * N * 512
        emit(probe="vae_attention", B=B, N=N, hip_us=round(t, 1), hip_tflops=round(flops / t / 1e6, 1), sdpa_us=round(ts, 1),
             sdpa_tflops=round(flops / ts / 1e6, 1))


def pmc_attn():
    B, H, N = 20, 10, 4096
    q, k, v = (torch.randn(B, N, H * 64, device=DEV).to(torch.bfloat16) for _ in range(3))
    for path in (0, 2, 4):
        for _ in range(3):
            ops.flash_attention(q, k, v, H, v_path=path)
    B, H, N = 20, 20, 1024
    q, k, v = (torch.randn(B, N, H * 64, device=DEV).to(torch.bfloat16) for _ in range(3))
    kc, vc = (torch.randn(B, 77, H * 64, device=DEV).to(torch.bfloat16) for _ in range(2))
    for path in (0, 4):
        for _ in range(3):
            ops.flash_attention(q, k, v, H, v_path=path)
    for path in (0, 8):
        for _ in range(3):
            ops.flash_attention(q, kc, vc, H, v_path=path)
    torch.cuda.synchronize()
    print("pmcattn done")


if __name__ == "__main__":
    what = sys.argv[1:] or ["attn", "gn", "vae"]
    with torch.no_grad():
        if "attn" in what:
            probe_attn()
        if "gn" in what:
            probe_gn()
        if "vae" in what:
            probe_vae()
        if "pmcattn" in what:
            pmc_attn()
