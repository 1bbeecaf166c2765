"""In-situ A/B of builds of libelastic_hip.so: the hipGraph-replayed SDXL UNet forward (product switches) at the batch sizes of the
headline workload.  ONE process, one set of weights: ops.py looks the entry points up in `_hip.lib()` at every call, so a forward
captured while `_hip._LIB` is library A is a graph of A's kernels; the graphs of all arms are then replayed in turn (A B A B ...), the
median per arm is reported, and the outputs are compared bit for bit.

    python tools/fwd_ab.py --libs tools/r5_patches/build/libelastic_hip_r4_product.so,product [--batches 20,6] [--modes fp16]

`--modes` are UNet precision modes: fp16 | bf16 | mixed (fp16 weights / MFMA operands with an fp32 residual stream,
`UNet2DConditionModel.set_residual_dtype`, when the library build has it)."""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import elasticdiffusion_official_amd  # noqa: F401,E402
from elasticdiffusion_official_amd import _hip, models as M  # noqa: E402


def load(path):
    if path == "product":
        return _hip.lib()
    L = ctypes.CDLL(os.path.abspath(path))
    for name, argtypes in _hip.SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is None:
            continue
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == "ed_error_string" else ctypes.c_int64 if name.endswith("_workspace") else ctypes.c_int)
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="tools/r5_patches/build/libelastic_hip_r4_product.so,product")
    ap.add_argument("--batches", default="20,6")
    ap.add_argument("--modes", default="fp16")
    ap.add_argument("--family", default="sdxl")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    product = _hip.lib()
    libs = [(p, load(p)) for p in a.libs.split(",")]
    cfg = M.UNET_CONFIGS[a.family]
    for mode in a.modes.split(","):
        dt = torch.bfloat16 if mode == "bf16" else torch.float16
        torch.manual_seed(0)
        unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False)
        if M.CHANNELS_LAST:
            unet = unet.to(memory_format=torch.channels_last)
        if mode == "mixed":
            unet.set_residual_dtype(torch.float32)
        S = cfg["sample_size"]
        for batch in [int(v) for v in a.batches.split(",")]:
            x = torch.randn(batch, 4, S, S, device="cuda", dtype=dt)
            e = torch.randn(batch, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
            kw = None
            if cfg["pooled_projection_dim"]:
                kw = {"text_embeds": torch.randn(batch, cfg["pooled_projection_dim"], device="cuda", dtype=dt),
                      "time_ids": torch.zeros(batch, 6, device="cuda")}
            t = torch.tensor(500, device="cuda")
            arms = []
            for path, L in libs:
                _hip._LIB = L
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
                arms.append({"lib": path, "graph": g, "out": out, "kv": kv, "ms": []})
            _hip._LIB = product
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
            for arm in arms:
                med = statistics.median(arm["ms"])
                print(json.dumps({"family": a.family, "mode": mode, "batch": batch, "lib": arm["lib"], "median_ms": round(med, 3),
                                  "min_ms": round(min(arm["ms"]), 3), "speedup_vs_first": round(base / med, 4),
                                  "bit_identical_to_first": bool(torch.equal(arm["out"], arms[0]["out"])),
                                  "finite": bool(torch.isfinite(arm["out"].float()).all())}), flush=True)
            del arms
        del unet
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
