"""Why is every MFMA kernel ~5 % slower in fp16 than in bf16 (profiles/r4_s18_*)?  The product's GEGLU GEMM and flash attention on
  (a) fp16 operands, (b) bf16 operands of the same distribution, (c) fp16 operands whose mantissas were first rounded to bf16's 7 bits,
  (d) fp16 zeros -- same instruction stream in (a), (c), (d); if (c) runs like (b), the difference is the operands' bit activity (power ->
clock), not the fp16 code path.  Events around 10 (4) launches per arm, interleaved rounds, median."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

from elasticdiffusion_official_amd import ops


def graphed(fn, n=10):
    """(kernels of 0.45-1.1 ms: plain launches, events around n of them -- host time is hidden)"""
    fn()
    torch.cuda.synchronize()

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    return timed


g = torch.Generator().manual_seed(0)
M, K, I = 20480, 1280, 5120
x32 = torch.rand(M, K, generator=g) * 2 - 1
w32 = (torch.rand(2 * I, K, generator=g) * 2 - 1) / K ** 0.5
b32 = torch.rand(2 * I, generator=g) * 2 - 1
B, H, N, D = 20, 10, 4096, 64
q32, k32, v32 = (torch.randn(B, N, H * D, generator=g) for _ in range(3))


def arm(kind):
    if kind == "fp16":
        c = lambda t: t.to("cuda", torch.float16)   # noqa: E731
    elif kind == "bf16":
        c = lambda t: t.to("cuda", torch.bfloat16)   # noqa: E731
    elif kind == "fp16_with_bf16_mantissas":
        c = lambda t: t.to("cuda", torch.bfloat16).to(torch.float16)   # noqa: E731   (exact: bf16's 7 mantissa bits fit fp16's 10; range is small)
    else:
        c = lambda t: torch.zeros_like(t, device="cuda", dtype=torch.float16)   # noqa: E731
    x, w, b = c(x32), c(w32), c(b32)
    q, k, v = c(q32), c(k32), c(v32)
    return {"geglu_gemm": graphed(lambda: ops.geglu_gemm(x, w, b)), "flash_attention": graphed(lambda: ops.flash_attention(q, k, v, H), 4)}


kinds = ["fp16", "bf16", "fp16_with_bf16_mantissas", "fp16_zeros"]
arms = {kd: arm(kd) for kd in kinds}
ts = {(kd, op): [] for kd in kinds for op in ("geglu_gemm", "flash_attention")}
for _ in range(7):
    for kd in kinds:
        for op in ("geglu_gemm", "flash_attention"):
            ts[(kd, op)].append(arms[kd][op]())
flops = {"geglu_gemm": 4.0 * M * K * I, "flash_attention": 4.0 * B * H * N * N * D}
for op in ("geglu_gemm", "flash_attention"):
    rec = {"kernel": op}
    for kd in kinds:
        m = sorted(ts[(kd, op)])[3]
        rec[kd] = {"us": round(1e3 * m, 1), "tflops": round(flops[op] / m / 1e9, 1)}
    rec["fp16_over_bf16"] = round(rec["fp16"]["us"] / rec["bf16"]["us"], 4)
    rec["fp16_bf16_mantissas_over_bf16"] = round(rec["fp16_with_bf16_mantissas"]["us"] / rec["bf16"]["us"], 4)
    print(json.dumps(rec), flush=True)
