"""The per-rank forward table behind DESIGN 7's multi-GPU model, re-measured with the current library (the table in
DESIGN was from round 2 and predates every MFMA kernel of rounds 3-5).

For each batch size (the per-rank row counts of 1 / 2 / 4 / 8-way row sharding of the headline workload's 20- and 6-row forwards, with
1 .. 4 images in flight): the hipGraph-replayed SDXL UNet forward time, and from ONE eager forward under a TorchFunctionMode + ops.TIMER,
where the dense-contraction FLOPs went -- this repo's kernels (ed_geglu_gemm / ed_linear / ed_conv3x3_nhwc / ed_flash_attention) or the
library calls the shape policy left them with (F.linear -> hipBLASLt, F.conv2d -> MIOpen), e.g. because a grid is under GEMM_MIN_BLOCKS.

    python tools/r5_batch_table.py [--batches 20,12,10,8,6,5,4,3,2,1] [--family sdxl] [--dtype fp16]
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

import elasticdiffusion_official_amd  # noqa: F401,E402
from elasticdiffusion_official_amd import models as M, ops  # noqa: E402


class LibraryCalls(TorchFunctionMode):
    """FLOPs of the dense contractions that go to the libraries in one forward"""

    def __init__(self):
        super().__init__()
        self.flops = {"F.linear": 0.0, "F.conv2d": 0.0, "sdpa": 0.0, "matmul": 0.0}
        self.calls = {k: 0 for k in self.flops}

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.linear:
            x, w = args[0], args[1]
            self.flops["F.linear"] += 2.0 * x.numel() * w.shape[0]
            self.calls["F.linear"] += 1
        elif func is F.conv2d:
            x, w = args[0], args[1]
            stride = args[3] if len(args) > 3 else kwargs.get("stride", 1)
            s = stride if isinstance(stride, int) else stride[0]
            self.flops["F.conv2d"] += 2.0 * x.shape[0] * (x.shape[2] // s) * (x.shape[3] // s) * w.numel()
            self.calls["F.conv2d"] += 1
        elif func is F.scaled_dot_product_attention:
            q, k = args[0], args[1]
            self.flops["sdpa"] += 4.0 * q.numel() * k.shape[-2]
            self.calls["sdpa"] += 1
        elif func in (torch.matmul, torch.Tensor.matmul, torch.bmm, torch.baddbmm):
            self.calls["matmul"] += 1
        return func(*args, **kwargs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="20,12,10,8,6,5,4,3,2,1")
    ap.add_argument("--family", default="sdxl")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    cfg = M.UNET_CONFIGS[a.family]
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    torch.manual_seed(0)
    unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False)
    if M.CHANNELS_LAST:
        unet = unet.to(memory_format=torch.channels_last)
    S = cfg["sample_size"]
    for batch in [int(v) for v in a.batches.split(",")]:
        x = torch.randn(batch, 4, S, S, device="cuda", dtype=dt)
        e = torch.randn(batch, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
        kw = None
        if cfg["pooled_projection_dim"]:
            kw = {"text_embeds": torch.randn(batch, cfg["pooled_projection_dim"], device="cuda", dtype=dt), "time_ids": torch.zeros(batch, 6, device="cuda")}
        t = torch.tensor(500, device="cuda")
        t_first = time.perf_counter()
        with torch.no_grad():
            kv = unet.cross_attention_kv(e, None)
            fwd = lambda: unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw, cross_kv=kv).sample   # noqa: E731
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd()
                torch.cuda.synchronize()
                first_s = time.perf_counter() - t_first          # includes any first-use library search for this batch size
                fwd()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fwd()
            torch.cuda.synchronize()
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize()
            ms = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                g.replay()
                torch.cuda.synchronize()
                ms.append(1e3 * (time.perf_counter() - t0))
            ops.TIMER.start()
            with LibraryCalls() as lc:
                fwd()
            kt = ops.TIMER.stop()
            work = {k: v[0] for k, v in ops.TIMER.work.items() if v[0]}
        ours = sum(work.values())
        lib = lc.flops["F.linear"] + lc.flops["F.conv2d"] + lc.flops["sdpa"]
        med = statistics.median(ms)
        print(json.dumps({"family": a.family, "dtype": a.dtype, "rows": batch, "replay_ms": round(med, 3), "ms_per_row": round(med / batch, 3),
                          "first_eager_forward_s": round(first_s, 2), "finite": bool(torch.isfinite(out.float()).all()),
                          "contraction_tflop_ours": {k: round(v / 1e12, 4) for k, v in work.items()},
                          "contraction_tflop_library": {k: round(v / 1e12, 4) for k, v in lc.flops.items() if v},
                          "library_calls": {k: v for k, v in lc.calls.items() if v},
                          "library_share_of_contraction_flops": round(lib / max(ours + lib, 1.0), 4),
                          "our_launches": {k: v[0] for k, v in kt.items()}}), flush=True)
        del g, out, kv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
