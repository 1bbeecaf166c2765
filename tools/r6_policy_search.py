"""Round 6: in-situ search of the shape policy.  One hipGraph of the SDXL forward per arm, replayed in turn: the baseline (the product's
policy) and, for every (kind, shape) decision taken inside that forward whose kernel CAN run the shape (ops.*_ok), the forward with just
that one decision flipped.  Isolated-loop wins of short kernels did not survive in the forward (profiles/r6_s7_policy_split.jsonl), so the
policy is judged here, where it runs.   python tools/r6_policy_search.py [--batches 40,12] [--reps 10]"""
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
ap.add_argument("--batches", default="40,12")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--family", default="sdxl")
a = ap.parse_args()

BASE = {"linear": ops.linear_wins, "conv": ops.conv3x3_wins, "geglu": ops.geglu_gemm_wins}
OK = {"linear": ops.linear_ok, "conv": ops.conv3x3_ok, "geglu": ops.geglu_gemm_ok}
seen, flip = {}, {}


def hook(kind):
    def f(*k):
        d = BASE[kind](*k)
        seen.setdefault((kind,) + tuple(int(v) for v in k), [d, 0])[1] += 1
        key = (kind,) + tuple(int(v) for v in k)
        return (not d) if flip.get(key) else d
    return f


ops.linear_wins, ops.conv3x3_wins, ops.geglu_gemm_wins = hook("linear"), hook("conv"), hook("geglu")
cfg = M.UNET_CONFIGS[a.family]
dt = torch.float16
torch.manual_seed(0)
unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False).to(memory_format=torch.channels_last)
S = cfg["sample_size"]
for batch in [int(v) for v in a.batches.split(",")]:
    x = torch.randn(batch, 4, S, S, device="cuda", dtype=dt)
    e = torch.randn(batch, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
    kw = {}
    if cfg.get("pooled_projection_dim"):
        kw = {"added_cond_kwargs": {"text_embeds": torch.randn(batch, cfg["pooled_projection_dim"], device="cuda", dtype=dt),
                                     "time_ids": torch.zeros(batch, 6, device="cuda")}}
    t = torch.tensor(500, device="cuda")

    def capture():
        with torch.no_grad():
            kv = unet.cross_attention_kv(e, None)
            fwd = lambda: unet(x, t, encoder_hidden_states=e, cross_kv=kv, **kw).sample   # noqa: E731
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
        return {"graph": g, "out": out, "kv": kv, "ms": []}

    seen.clear(), flip.clear()
    arms = [dict(capture(), name="baseline")]
    seen_base = {k: list(v) for k, v in seen.items()}
    cands = [k for k, (d, n) in sorted(seen_base.items()) if OK[k[0]](*k[1:])]
    for key in cands:
        flip.clear()
        flip[key] = True
        try:
            arms.append(dict(capture(), name=key))
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"batch": batch, "flip": list(key), "error": str(ex)[:200]}), flush=True)
    flip.clear()
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
    print(json.dumps({"batch": batch, "arm": "baseline", "median_ms": round(base, 3), "decisions": len(seen), "candidates": len(cands)}), flush=True)
    for arm in arms[1:]:
        med = statistics.median(arm["ms"])
        d, n = seen_base[arm["name"]]
        print(json.dumps({"batch": batch, "flip": list(arm["name"]), "product_uses_own_kernel": bool(d), "calls_per_forward": n // 3,
                          "median_ms": round(med, 3), "gain_ms_if_flipped": round(base - med, 3),
                          "rel_l2_vs_baseline": float(f"{float((arm['out'].float() - ref).norm() / ref.norm()):.3e}")}), flush=True)
    del arms
    torch.cuda.empty_cache()
