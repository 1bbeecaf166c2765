"""Where does the 16-bit UNet's error against the fp32 reference path come from -- and which of it would an fp32 residual stream remove?

VERDICT r4 item 1: the fp16 loop drifts 4.7-5.4e-3 against the fp32 oracle; north_star names 1e-3.  This tool attributes the error to
the places where a 16-bit UNet rounds, by running the fp32 model (this repo's modules, plain torch ops) under a TorchFunctionMode that
injects exactly one class of roundings at a time (RNE to fp16 / bf16 through .to(dtype).float(); everything else stays fp32):

    w        weights of every Linear / Conv2d (what torch.autocast does to the reference's fp32 weights at every call, ED:1012)
    act      the activation operand of every Linear / Conv2d (the MFMA A operand; autocast does the same)
    attn     q, k, v and the un-normalised probabilities P of every attention (MFMA operands of the two attention contractions)
    out      the outputs of Linear / Conv2d / attention that do not go straight into another contraction (16-bit stores of branch results)
    stream   the result of every residual / broadcast add (the 16-bit residual stream: pure-fp16 modules AND autocast have it -- conv /
             linear outputs are fp16 under autocast, so fp16 + fp16 stays fp16)

and the combinations that correspond to real designs:

    fp16_model      = w + act + attn + out + stream     (today's product; the reference's GPU path differs only in rounding points
                                                          inside norms / GELU that the fused kernels do not have)
    mixed_out16     = w + act + attn + out              (fp32 residual stream, branch results stored in 16 bit, added in fp32)
    mixed_out32     = w + act + attn                    (fp32 residual stream, epilogues add the residual from the fp32 accumulator)

Two measurements, both on the CPU (no GPU minute needed): (1) ONE forward of the full-width SDXL architecture (2.567 B parameters,
seeded synthetic weights), rel-L2 of the output against the un-rounded fp32 forward; (2) the reduced-width architecture inside the
oracle's denoising loop (cfg3 geometry, guidance 10, RePaint + RRG), per-timestep rel-L2 of the latent against the fp32 loop.

    python tools/r5_precision_modes.py forward [--batch 1] [--dtype fp16] [--family sdxl] [--small]
    python tools/r5_precision_modes.py loop [--steps 3] [--dtype fp16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

from elasticdiffusion_official_amd import models as M

ADDS = {torch.add, torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.Tensor.__iadd__, torch.Tensor.add_}
MODES = {
    "w": {"w"}, "act": {"act"}, "attn": {"attn"}, "out": {"out"}, "stream": {"stream"},
    "fp16_model": {"w", "act", "attn", "out", "stream"},
    "mixed_out16": {"w", "act", "attn", "out"},
    "mixed_out32": {"w", "act", "attn"},
    "operands_only_no_w": {"act", "attn"},
    # candidates for a tolerance-meeting mode on the MFMA pipe: split (hi, lo) operands remove a rounding class entirely
    "attn_o": {"attn_o"},                              # only the 16-bit store of every attention's output
    "split_x3_fp16_attention": {"attn", "attn_o"},     # activations AND weights split (3 MFMA passes), fp32 stream; fp16 flash attention in / out
    "split_x2_fp16_attention": {"w", "attn", "attn_o"},   # activations split (2 passes), fp16 weights
    "split_x2_attention_f32out": {"w", "attn"},
}


class Rounding(TorchFunctionMode):
    """injects the roundings named in ``on`` into an fp32 forward"""

    def __init__(self, on, dtype=torch.float16):
        super().__init__()
        self.on, self.dt = set(on), dtype

    def q(self, t, what):
        if t is None or what not in self.on or not torch.is_tensor(t) or t.dtype != torch.float32:
            return t
        return t.to(self.dt).float()

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.linear or func is F.conv2d:
            x, w = args[0], args[1]
            b = args[2] if len(args) > 2 else kwargs.get("bias")
            rest = args[3:]
            kw = {k: v for k, v in kwargs.items() if k != "bias"}
            y = func(self.q(x, "act"), self.q(w, "w"), self.q(b, "w"), *rest, **kw)
            geglu = func is F.linear and w.shape[0] == 8 * w.shape[1]      # the fused GEGLU GEMM never stores its projection
            return y if geglu else self.q(y, "out")
        if func is F.scaled_dot_product_attention:
            q, k, v = (self.q(t, "attn") for t in args[:3])
            s = (q @ k.transpose(-1, -2)) * q.shape[-1] ** -0.5
            p = torch.exp(s - s.amax(-1, keepdim=True))
            o = (self.q(p, "attn") @ v) / p.sum(-1, keepdim=True)       # the kernels pack P to 16 bits for the second MFMA; l is an fp32 sum
            return self.q(self.q(o, "out"), "attn_o")
        y = func(*args, **kwargs)
        if func in ADDS and torch.is_tensor(y) and y.dtype == torch.float32 and y.dim() >= 2:
            return self.q(y, "stream")
        return y


class Rounded(nn.Module):
    """an fp32 module whose forward runs under a Rounding mode"""

    def __init__(self, mod, on, dtype):
        super().__init__()
        self.mod, self.on, self.dt = mod, on, dtype
        self.config = mod.config
        if hasattr(mod, "add_embedding"):
            self.add_embedding = mod.add_embedding

    def forward(self, *a, **kw):
        with Rounding(self.on, self.dt):
            return self.mod(*a, **kw)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@torch.no_grad()
def forward_report(family, small, batch, dtype, seed=0):
    cfg = (M.SMALL_UNET_CONFIGS if small else M.UNET_CONFIGS)[family]
    unet = M.UNet2DConditionModel(**cfg)
    M._seeded_init(unet, seed)
    unet = unet.eval().requires_grad_(False)
    S = cfg["sample_size"]
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 4, S, S, generator=g)
    txt = torch.randn(batch, 77, cfg["cross_attention_dim"], generator=g)
    kw = None
    if cfg["pooled_projection_dim"]:
        kw = {"text_embeds": torch.randn(batch, cfg["pooled_projection_dim"], generator=g),
              "time_ids": torch.tensor([[4096., 8192., 0., 0., 4096., 8192.]]).expand(batch, -1)}
    t = torch.tensor(500)
    t0 = time.time()
    ref = unet(x, t, encoder_hidden_states=txt, added_cond_kwargs=kw)["sample"]
    print(json.dumps({"what": "fp32 forward", "family": family, "small": small, "batch": batch, "seconds": round(time.time() - t0, 1),
                      "out_abs_mean": float(ref.abs().mean())}), flush=True)
    for name, on in MODES.items():
        y = Rounded(unet, on, dtype)(x, t, encoder_hidden_states=txt, added_cond_kwargs=kw)["sample"]
        print(json.dumps({"what": "forward", "mode": name, "roundings": sorted(on), "dtype": str(dtype)[6:], "rel_l2_vs_fp32": rel_l2(y, ref),
                          "finite": bool(torch.isfinite(y).all())}), flush=True)


def loop_report(steps, dtype, modes):
    from tests import realarch as R
    c = dict(R.REAL_CASES["cfg3_xl_1024x2048"], steps=steps)
    unet, vae, cn = R.build_small(c["sd"])
    t0 = time.time()
    want, _ = R.run_oracle(c, unet, vae, cn)
    print(json.dumps({"what": "fp32 oracle loop", "steps": steps, "seconds": round(time.time() - t0, 1)}), flush=True)
    for name in modes:
        got, _ = R.run_oracle(c, Rounded(unet, MODES[name], dtype), vae, cn)
        print(json.dumps({"what": "loop", "mode": name, "dtype": str(dtype)[6:], "rel_l2_per_step": [R.rel_l2(a, b) for a, b in zip(got, want)]}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=("forward", "loop"))
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--family", default="sdxl")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--modes", default="fp16_model,mixed_out16,mixed_out32,w,stream")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    if a.what == "forward":
        forward_report(a.family, a.small, a.batch, dt)
    else:
        loop_report(a.steps, dt, a.modes.split(","))
