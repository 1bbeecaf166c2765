"""Round 6: the 128-row tile mode of ed_linear / ed_conv3x3_nhwc (tile_phases_rows) against 256-row tiles (ED_GEMM_ROWS=1 / 0, read at every
launch) and against the library call, at the under-filled shapes of the batch-6 forward and of the 1- / 3-row per-rank forwards of the multi-GPU
layout.  Interleaved rounds, median.   python tools/r6_rows_probe.py [--rounds 7]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import ops

ops.GEMM_MIN_BLOCKS = 1


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def arms(name, flops, fns, rounds, blocks):
    t = {k: [] for k in fns}
    outs = {}
    for k, fn in fns.items():
        outs[k] = fn().clone()
    for _ in range(rounds):
        for k, fn in fns.items():
            t[k].append(timed(fn))
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    rec = {"case": name, "tiles256": blocks, **{k + "_us": round(v, 1) for k, v in med.items()},
           "rows128_vs_rows256": round(med["tiles256"] / med["tiles128"], 4), "rows128_vs_library": round(med["library"] / med["tiles128"], 4),
           "rows256_vs_library": round(med["library"] / med["tiles256"], 4), "tiles128_tflops": round(flops / med["tiles128"] / 1e6, 1),
           "bit_identical_128_vs_256": bool(torch.equal(outs["tiles128"], outs["tiles256"]))}
    print(json.dumps(rec), flush=True)


def with_rows(mode, fn):
    def run():
        os.environ["ED_GEMM_ROWS"] = mode
        try:
            return fn()
        finally:
            os.environ.pop("ED_GEMM_ROWS", None)
    return run


ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
a = ap.parse_args()
dt, cl, dev = torch.float16, torch.channels_last, "cuda"
g = torch.Generator().manual_seed(0)
for (B, H, W, Cin, N) in [(6, 32, 32, 1280, 1280), (6, 32, 32, 2560, 1280), (6, 64, 64, 640, 640), (6, 64, 64, 1280, 640), (3, 32, 32, 1280, 1280),
                          (3, 64, 64, 640, 640), (3, 128, 128, 320, 320), (1, 64, 64, 640, 640), (1, 128, 128, 320, 320), (20, 32, 32, 1280, 1280)]:
    x = (torch.rand(B, Cin, H, W, generator=g) * 2 - 1).to(dev, dt).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev, dt).contiguous(memory_format=cl)
    b = (torch.rand(N, generator=g) * 2 - 1).to(dev, dt)
    blocks = -(-(B * H * W) // 256) * -(-N // 256)
    arms(f"conv {B}x{H}x{W} {Cin}->{N}", 2.0 * B * H * W * 9 * Cin * N,
         {"tiles256": with_rows("0", lambda: ops.conv3x3_nhwc(x, w, b)), "tiles128": with_rows("1", lambda: ops.conv3x3_nhwc(x, w, b)),
          "library": lambda: F.conv2d(x, w, b, padding=1)}, a.rounds, blocks)
for (M, K, N) in [(6144, 1280, 1280), (6144, 5120, 1280), (6144, 1280, 3840), (24576, 640, 640), (24576, 640, 1920), (24576, 2560, 640),
                  (3072, 1280, 1280), (3072, 5120, 1280), (12288, 640, 640), (12288, 2560, 640), (1024, 1280, 1280), (4096, 640, 640)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev, dt)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dev, dt)
    b = (torch.rand(N, generator=g) * 2 - 1).to(dev, dt)
    blocks = -(-M // 256) * -(-N // 256)
    arms(f"linear {M}x{K}->{N}", 2.0 * M * K * N,
         {"tiles256": with_rows("0", lambda: ops.linear(x, w, b)), "tiles128": with_rows("1", lambda: ops.linear(x, w, b)),
          "library": lambda: F.linear(x, w, b)}, a.rounds, blocks)
