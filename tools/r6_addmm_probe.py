"""Round 6 probe: what the library charges for the residual as the GEMM's C operand (torch.addmm, beta = 1) against the plain projection, at the
K >= 1280 shapes hipBLASLt keeps -- the price of taking the residual add out of ed_add_layernorm for those projections.
python tools/r6_addmm_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator().manual_seed(0)
for (M, K, N) in [(20480, 1280, 1280), (40960, 1280, 1280), (12288, 1280, 1280), (6144, 1280, 1280), (20480, 5120, 1280), (40960, 5120, 1280),
                  (12288, 5120, 1280)]:
    x = torch.randn(M, K, generator=g).to("cuda", torch.float16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.float16)
    b = torch.randn(N, generator=g).to("cuda", torch.float16)
    r = torch.randn(M, N, generator=g).to("cuda", torch.float16)
    gam, bet = torch.ones(N, device="cuda", dtype=torch.float16), torch.zeros(N, device="cuda", dtype=torch.float16)
    rounds = {k: [] for k in ("linear", "addmm", "add_layernorm", "layernorm", "own_linear_residual")}
    y = F.linear(x, w, b)
    for _ in range(5):
        rounds["linear"].append(timed(lambda: F.linear(x, w, b)))
        rounds["addmm"].append(timed(lambda: torch.addmm(r, x, w.t())))
        rounds["add_layernorm"].append(timed(lambda: ops.add_layernorm(y, r, gam, bet, 1e-5)))
        rounds["layernorm"].append(timed(lambda: ops.layernorm(y, gam, bet, 1e-5)))
        rounds["own_linear_residual"].append(timed(lambda: ops.linear(x, w, b, r)))
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in rounds.items()}
    print(json.dumps({"M": M, "K": K, "N": N, **{k + "_us": v for k, v in med.items()},
                      "pair_now_us": round(med["linear"] + med["add_layernorm"], 1),
                      "pair_addmm_us": round(med["addmm"] + med["layernorm"], 1),
                      "pair_own_us": round(med["own_linear_residual"] + med["layernorm"], 1)}), flush=True)
