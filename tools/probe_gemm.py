"""GPU probe: every distinct Linear / 3x3-convolution shape of a UNet forward, library call vs this repo's 8-phase kernel
(csrc/gemm_kernels.hip), interleaved rounds in one process (median).  The shapes are RECORDED from a real forward of the
architecture at the given batch sizes (library path, switches off), so the table is the workload's, with call counts.

    python tools/probe_gemm.py [sdxl|sd15] [batches, e.g. 20,6] [--dtype fp16|bf16] [--rounds 5] [--out file.jsonl]

Feeds ops.linear_wins / conv3x3_ok's thresholds (profiles/r4_*_probe_gemm_*.jsonl)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401  (MIOpen cache location)
from elasticdiffusion_official_amd import models as M, ops


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def med(v):
    return sorted(v)[len(v) // 2]


def record_shapes(fam, B, dt):
    """One library-path forward with F.linear / F.conv2d wrapped: {("linear", M, K, N): calls, ("conv3", B, H, W, Cin, N): calls,
    ("geglu", M, K, I): calls}."""
    cfg = M.UNET_CONFIGS[fam]
    torch.manual_seed(0)
    unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False).to(memory_format=torch.channels_last)
    S = cfg["sample_size"]
    x = torch.randn(B, 4, S, S, device="cuda", dtype=dt)
    e = torch.randn(B, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
    kw = None
    if cfg["pooled_projection_dim"]:
        kw = {"text_embeds": torch.randn(B, cfg["pooled_projection_dim"], device="cuda", dtype=dt), "time_ids": torch.zeros(B, 6, device="cuda")}
    seen = {}
    real_linear, real_conv = F.linear, F.conv2d
    in_geglu = [False]

    def lin(inp, w, b=None):
        key = ("geglu" if in_geglu[0] else "linear", inp.numel() // inp.shape[-1], inp.shape[-1], w.shape[0] // (2 if in_geglu[0] else 1))
        seen[key] = seen.get(key, 0) + 1
        return real_linear(inp, w, b)

    def conv(inp, w, b=None, stride=1, padding=0, *a, **k):
        st = stride if isinstance(stride, int) else stride[0]
        if w.shape[-1] == 3 and st == 1:
            key = ("conv3", inp.shape[0], inp.shape[2], inp.shape[3], inp.shape[1], w.shape[0])
            seen[key] = seen.get(key, 0) + 1
        return real_conv(inp, w, b, stride, padding, *a, **k)

    geglu_fwd = M.GEGLU.forward

    def geglu(self, t):
        in_geglu[0] = True
        try:
            return geglu_fwd(self, t)
        finally:
            in_geglu[0] = False

    saved = (M.HIP_GEGLU_GEMM, M.HIP_LINEAR, M.HIP_CONV3X3)
    M.HIP_GEGLU_GEMM = M.HIP_LINEAR = M.HIP_CONV3X3 = False
    F.linear, F.conv2d, M.GEGLU.forward = lin, conv, geglu
    try:
        with torch.no_grad():
            unet(x, torch.tensor(500, device="cuda"), encoder_hidden_states=e, added_cond_kwargs=kw)
    finally:
        F.linear, F.conv2d, M.GEGLU.forward = real_linear, real_conv, geglu_fwd
        M.HIP_GEGLU_GEMM, M.HIP_LINEAR, M.HIP_CONV3X3 = saved
    del unet
    torch.cuda.empty_cache()
    return seen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fam", nargs="?", default="sdxl")
    ap.add_argument("batches", nargs="?", default="20,6")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    ops.GEMM_MIN_BLOCKS = 1
    g = torch.Generator().manual_seed(0)
    out = open(a.out, "w") if a.out else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")

    for B in [int(b) for b in a.batches.split(",")]:
        shapes = record_shapes(a.fam, B, dt)
        saved_total = 0.0
        for key, calls in sorted(shapes.items(), key=lambda kv: kv[0]):
            kind = key[0]
            if kind == "linear":
                _, Mr, K, N = key
                if not (K % 64 == 0 and N % 8 == 0 and Mr >= 256):
                    continue
                x = ((torch.rand(Mr, K, generator=g) * 2 - 1)).to("cuda", dt)
                w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
                b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
                mine, lib = (lambda: ops.linear(x, w, b)), (lambda: F.linear(x, w, b))
                flops = 2.0 * Mr * K * N
                blocks = -(-Mr // 256) * -(-N // 256)
            elif kind == "geglu":
                _, Mr, K, I = key
                if not ops.geglu_gemm_ok(Mr, K, I):
                    continue
                x = ((torch.rand(Mr, K, generator=g) * 2 - 1)).to("cuda", dt)
                w = ((torch.rand(2 * I, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
                b = (torch.rand(2 * I, generator=g) * 2 - 1).to("cuda", dt)
                mine, lib = (lambda: ops.geglu_gemm(x, w, b)), (lambda: ops.geglu(F.linear(x, w, b), I))
                flops = 4.0 * Mr * K * I
                blocks = -(-Mr // 256) * (I // 128)
            else:
                _, Bc, H, W, Cin, N = key
                if not (Cin % 64 == 0 and N % 8 == 0):
                    continue
                cl = torch.channels_last
                x = ((torch.rand(Bc, Cin, H, W, generator=g) * 2 - 1)).to("cuda", dt).contiguous(memory_format=cl)
                w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to("cuda", dt).contiguous(memory_format=cl)
                b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
                mine, lib = (lambda: ops.conv3x3_nhwc(x, w, b)), (lambda: F.conv2d(x, w, None, padding=1))   # the product's convs run bias-free
                flops = 2.0 * Bc * H * W * 9 * Cin * N
                blocks = -(-(Bc * H * W) // 256) * -(-N // 256)
            tm, tl = [], []
            for _ in range(a.rounds):
                tm.append(timed(mine))
                tl.append(timed(lib))
            rec = {"batch": B, "kind": kind, "shape": list(key[1:]), "calls_per_forward": calls, "blocks": blocks,
                   "this_ms": round(med(tm), 4), "library_ms": round(med(tl), 4), "this_tflops": round(flops / med(tm) / 1e9, 1),
                   "library_tflops": round(flops / med(tl) / 1e9, 1), "speedup": round(med(tl) / med(tm), 3),
                   "saved_ms_per_forward": round(calls * (med(tl) - med(tm)), 3)}
            saved_total += max(0.0, calls * (med(tl) - med(tm)))
            emit(rec)
        emit({"batch": B, "family": a.fam, "dtype": a.dtype, "sum_of_positive_savings_ms_per_forward": round(saved_total, 2)})


if __name__ == "__main__":
    main()
