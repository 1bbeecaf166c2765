"""Round 6: in-process A/B of the shape POLICY (which contractions the model hands to this repo's kernels) and of the launcher's tile-height
choice, on the hipGraph-replayed SDXL forward: arm "r5" = round-5 policy, 256-row tiles only (ED_GEMM_ROWS=0 while capturing); "rows" = the
same policy with the launcher picking 128-row tiles; "r6" = the round-6 policy (under-filled grids as 128-row tiles, 288-tile convolutions).
Graphs of all arms are replayed in turn; median per arm.   python tools/r6_policy_ab.py [--batches 20,6,3,1]"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import models as M, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="20,6,3,1")
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--arms", default="r5,rows,r6")
a = ap.parse_args()
new_rows_mode, new_grid_ok = ops.gemm_rows_mode, ops._gemm_grid_ok
new_linear_wins, new_conv_wins = ops.linear_wins, ops.conv3x3_wins


def old_grid_ok(blocks, M_=None, ncb=None, min_fill=None):
    return new_grid_ok(blocks)          # round 5: blocks >= 96 and one round or rounds >= 70 % full


def _with_old(fn):
    """evaluate one of the round-6 shape predicates under the round-5 grid rules ("rows" arm)"""
    def g(*k):
        keep = ops.gemm_rows_mode, ops._gemm_grid_ok
        ops.gemm_rows_mode, ops._gemm_grid_ok = (lambda *kk: False), old_grid_ok
        try:
            return fn(*k)
        finally:
            ops.gemm_rows_mode, ops._gemm_grid_ok = keep
    return g


def set_policy(name):
    ops.linear_wins, ops.conv3x3_wins = new_linear_wins, new_conv_wins
    if name in ("r6", "r6lin", "r6conv"):
        ops.gemm_rows_mode, ops._gemm_grid_ok = new_rows_mode, new_grid_ok
        os.environ.pop("ED_GEMM_ROWS", None)
        if name == "r6lin":        # round-6 policy for the projections only
            ops.conv3x3_wins = _with_old(new_conv_wins)
        if name == "r6conv":       # ... for the convolutions only
            ops.linear_wins = _with_old(new_linear_wins)
    else:
        ops.gemm_rows_mode, ops._gemm_grid_ok = (lambda *k: False), old_grid_ok
        if name == "r5":
            os.environ["ED_GEMM_ROWS"] = "0"
        else:
            os.environ.pop("ED_GEMM_ROWS", None)


cfg = M.UNET_CONFIGS["sdxl"]
dt = torch.float16
torch.manual_seed(0)
unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False).to(memory_format=torch.channels_last)
S = cfg["sample_size"]
for batch in [int(v) for v in a.batches.split(",")]:
    x = torch.randn(batch, 4, S, S, device="cuda", dtype=dt)
    e = torch.randn(batch, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
    kw = {"text_embeds": torch.randn(batch, cfg["pooled_projection_dim"], device="cuda", dtype=dt), "time_ids": torch.zeros(batch, 6, device="cuda")}
    t = torch.tensor(500, device="cuda")
    arms = []
    for name in a.arms.split(","):
        set_policy(name)
        with torch.no_grad():
            kv = unet.cross_attention_kv(e, None)
            fwd = lambda: unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw, cross_kv=kv).sample   # noqa: E731
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    fwd()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fwd()
        torch.cuda.synchronize()
        arms.append({"policy": name, "graph": g, "out": out, "kv": kv, "ms": []})
    set_policy("r6")
    for arm in arms:
        arm["graph"].replay()
    torch.cuda.synchronize()
    for _ in range(a.reps):
        for arm in arms:
            t0 = time.perf_counter()
            arm["graph"].replay()
            torch.cuda.synchronize()
            arm["ms"].append(1e3 * (time.perf_counter() - t0))
    base = statistics.median(arms[0]["ms"])
    ref = arms[0]["out"].float()
    for arm in arms:
        med = statistics.median(arm["ms"])
        print(json.dumps({"batch": batch, "policy": arm["policy"], "median_ms": round(med, 3), "min_ms": round(min(arm["ms"]), 3),
                          "speedup_vs_r5": round(base / med, 4),
                          "rel_l2_vs_r5": float(f"{float((arm['out'].float() - ref).norm() / ref.norm()):.3e}")}), flush=True)
    del arms
