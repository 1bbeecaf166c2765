"""GPU probe for round 2 (plumbing measurement, not a test).  One process, results as JSON lines on stdout:

  attn     ed_flash_attention (both V paths) vs F.scaled_dot_product_attention at the SDXL shapes, HIP-event timed
  fused    ed_add_layernorm / ed_tokens_add_nchw vs the torch ops they replace
  unet     SDXL UNet forward (eager) at batch 20 / 6 with each round-2 switch toggled off in turn
  table    hipGraph-replayed forward time vs batch (the per-rank batches of 1/2/4/8-way row sharding)

usage: r2_probe.py [attn] [fused] [unet] [table=20,10,6,3]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401  (sets the MIOpen cache location)
from elasticdiffusion_official_amd import models as M, ops

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
                           (20, 10, 4096, 77), (20, 20, 1024, 77), (3, 20, 1024, 1024), (1, 20, 1024, 1024)]:
        q, k, v = (torch.randn(B, n, H * 64, device=DEV).to(torch.bfloat16) for n in (Nq, Nk, Nk))
        q4, k4, v4 = (t.view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
        flops = 4.0 * B * H * Nq * Nk * 64
        t_sdpa = ev_time(lambda: F.scaled_dot_product_attention(q4, k4, v4))
        res = {"B": B, "H": H, "Nq": Nq, "Nk": Nk, "sdpa_us": round(t_sdpa, 1), "sdpa_tflops": round(flops / t_sdpa / 1e6, 1)}
        for path in (0, 1, 2):
            t = ev_time(lambda: ops.flash_attention(q, k, v, H, v_path=path))
            res[f"flash{path}_us"] = round(t, 1)
            res[f"flash{path}_tflops"] = round(flops / t / 1e6, 1)
        emit(probe="attn", **res)


def probe_fused():
    for (M_, D) in [(20 * 1024, 1280), (20 * 4096, 640)]:
        a, b = (torch.randn(M_, D, device=DEV).to(torch.bfloat16) for _ in range(2))
        w, bb = torch.ones(D, device=DEV, dtype=torch.bfloat16), torch.zeros(D, device=DEV, dtype=torch.bfloat16)
        t_f = ev_time(lambda: ops.add_layernorm(a, b, w, bb, 1e-5))
        t_u = ev_time(lambda: ops.layernorm(a + b, w, bb, 1e-5))
        emit(probe="add_layernorm", M=M_, D=D, fused_us=round(t_f, 1), add_then_ln_us=round(t_u, 1),
             fused_gbs=round(4 * M_ * D * 2 / t_f / 1e3, 1))
    for (N, C, HW) in [(20, 1280, 1024), (20, 640, 4096)]:
        x = torch.randn(N, C, int(HW ** 0.5), int(HW ** 0.5), device=DEV).to(torch.bfloat16)
        tok = torch.randn(N, HW, C, device=DEV).to(torch.bfloat16)
        hh = int(HW ** 0.5)
        t_f = ev_time(lambda: ops.tokens_add_nchw(x, tok))
        t_u = ev_time(lambda: x + tok.view(N, hh, hh, C).permute(0, 3, 1, 2))
        emit(probe="tokens_add_nchw", N=N, C=C, HW=HW, fused_us=round(t_f, 1), torch_us=round(t_u, 1),
             fused_gbs=round(3 * N * C * HW * 2 / t_f / 1e3, 1))
    x = torch.randn(20, 320, 128, 128, device=DEV).to(torch.bfloat16)
    cb = torch.randn(20, 320, device=DEV).to(torch.bfloat16)
    w, bb = torch.ones(320, device=DEV, dtype=torch.bfloat16), torch.zeros(320, device=DEV, dtype=torch.bfloat16)
    t_f = ev_time(lambda: ops.groupnorm(x, w, bb, 32, 1e-5, silu=True, chan_bias=cb))
    t_u = ev_time(lambda: ops.groupnorm(x + cb[:, :, None, None], w, bb, 32, 1e-5, silu=True))
    emit(probe="groupnorm_chan_bias", shape=list(x.shape), fused_us=round(t_f, 1), add_then_gn_us=round(t_u, 1))
    for shape in [(20, 320, 128, 128), (20, 960, 128, 128), (20, 640, 64, 64), (20, 1280, 32, 32), (6, 320, 128, 128)]:
        x = torch.randn(*shape, device=DEV).to(torch.bfloat16)
        w, bb = torch.ones(shape[1], device=DEV, dtype=torch.bfloat16), torch.zeros(shape[1], device=DEV, dtype=torch.bfloat16)
        res = {}
        for split in (True, False):
            ops.GROUPNORM_SPLIT = split
            t = ev_time(lambda: ops.groupnorm(x, w, bb, 32, 1e-5, silu=True))
            res["split_us" if split else "single_us"] = round(t, 1)
            res["split_gbs" if split else "single_gbs"] = round(3 * x.numel() * 2 / t / 1e3, 1)
        ops.GROUPNORM_SPLIT = True
        emit(probe="groupnorm_split", shape=list(shape), **res)


def build_unet():
    torch.manual_seed(0)
    cfg = M.UNET_CONFIGS["sdxl"]
    unet = M.UNet2DConditionModel(**cfg).to(DEV, torch.bfloat16).eval().requires_grad_(False)
    if M.CHANNELS_LAST:
        unet = unet.to(memory_format=torch.channels_last)
    return unet, cfg


def inputs(cfg, B):
    S = cfg["sample_size"]
    x = torch.randn(B, 4, S, S, device=DEV, dtype=torch.bfloat16)
    e = torch.randn(B, 77, cfg["cross_attention_dim"], device=DEV, dtype=torch.bfloat16)
    kw = {"text_embeds": torch.randn(B, cfg["pooled_projection_dim"], device=DEV, dtype=torch.bfloat16),
          "time_ids": torch.zeros(B, 6, device=DEV)}
    return x, e, kw, torch.tensor(500, device=DEV)


def probe_unet(unet, cfg):
    names = ("FLASH_ATTENTION", "FUSED_QKV", "FUSED_ADD_LAYERNORM", "FUSED_TOKENS_ADD", "FUSED_TEMB_ADD", "FUSED_CONV_BIAS")
    for B in (20, 6):
        x, e, kw, t = inputs(cfg, B)

        def fwd():
            with torch.no_grad():
                return unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)

        res = {"B": B, "channels_last": M.CHANNELS_LAST}
        res["all_on_ms"] = round(ev_time(fwd, reps=3, warm=2) / 1e3, 2)
        for n in names:
            setattr(M, n, False)
            res[f"off_{n}_ms"] = round(ev_time(fwd, reps=3, warm=1) / 1e3, 2)
            setattr(M, n, True)
        for n in names:
            setattr(M, n, False)
        res["all_off_ms"] = round(ev_time(fwd, reps=3, warm=1) / 1e3, 2)
        for n in names:
            setattr(M, n, True)
        ops.FLASH_V_PATH = 0
        res["flash_variant0_ms"] = round(ev_time(fwd, reps=3, warm=1) / 1e3, 2)
        ops.FLASH_V_PATH = None
        emit(probe="unet", **res)


def probe_table(unet, cfg, batches):
    flops = 6.761e12
    for B in batches:
        x, e, kw, t = inputs(cfg, B)
        with torch.no_grad():
            t0 = time.perf_counter()
            for _ in range(2):
                unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)
            torch.cuda.synchronize()
            warm_s = time.perf_counter() - t0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)
        us = ev_time(g.replay, reps=5, warm=2)
        emit(probe="table", B=B, graph_ms=round(us / 1e3, 2), ms_per_sample=round(us / 1e3 / B, 2),
             tflops=round(B * flops / us / 1e6, 1), warmup_s=round(warm_s, 1))
        del g


if __name__ == "__main__":
    args = sys.argv[1:] or ["attn", "fused", "unet", "table=20,10,6,3"]
    if "attn" in args:
        probe_attn()
    if "fused" in args:
        probe_fused()
    unet = None
    if "unet" in args:
        unet, cfg = build_unet()
        probe_unet(unet, cfg)
    for a in args:
        if a.startswith("table="):
            if unet is None:
                unet, cfg = build_unet()
            probe_table(unet, cfg, [int(b) for b in a[6:].split(",")])
