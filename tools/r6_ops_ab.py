"""Round 6: ops-level A/B of two builds of libelastic_hip.so in ONE process (ops.py looks the entry point up in `_hip.lib()` at every
call, so swapping `_hip._LIB` swaps the kernels): flash attention at the UNet's self-attention shapes (the one-vote lazy loop), the
channels-last GroupNorm at the UNet's shapes (block size per column count, >= 8 blocks per CU, separate finalize launch) and, with
--gelu, the GEGLU GEMM.  Interleaved rounds, median; outputs compared bit for bit (attention, GEGLU) / against fp32 torch (GroupNorm:
the chunking changed, so the fp32 partial sums differ in the last bit).

    python tools/r6_ops_ab.py --prev tools/ab/libelastic_hip_r5.so [--new product] [--rounds 7] [--only attn,gn]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import _hip, ops
from tools.fwd_ab import load


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


def ab(name, libs, fn, rounds, extra=None, check=None):
    outs, t = {}, {k: [] for k in libs}
    for k, L in libs.items():
        _hip._LIB = L
        outs[k] = fn().clone()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, L in libs.items():
            _hip._LIB = L
            t[k].append(timed(fn))
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    rec = {"case": name, "prev_us": round(med["prev"], 1), "new_us": round(med["new"], 1), "speedup": round(med["prev"] / med["new"], 4),
           "bit_identical": bool(torch.equal(outs["prev"], outs["new"]))}
    if extra:
        rec.update({k: round(v / med["new"], 1) for k, v in extra.items()})
    if check is not None:
        rec.update(check(outs))
    print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prev", default="tools/ab/libelastic_hip_r5.so")
    ap.add_argument("--new", default="product")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--only", default="attn,gn")
    a = ap.parse_args()
    product = _hip.lib()
    libs = {"prev": load(a.prev), "new": load(a.new)}
    dev, dt = "cuda", torch.float16
    g = torch.Generator().manual_seed(0)
    only = set(a.only.split(","))
    if "attn" in only:
        for (B, H, N) in [(20, 10, 4096), (20, 20, 1024), (6, 10, 4096), (6, 20, 1024)]:
            q, k, v = (torch.randn(B, N, H * 64, generator=g).to(dev, dt) for _ in range(3))
            flops = 4.0 * B * H * N * N * 64
            ab(f"flash_attention self B{B} H{H} N{N}", libs, lambda: ops.flash_attention(q, k, v, H), a.rounds,
               extra={"new_tflops": flops / 1e6})
        # a row whose maximum jumps late: the slow path of the one-vote loop (redo + rescale in one branch)
        B, H, N = 2, 10, 1024
        q, k, v = (torch.randn(B, N, H * 64, generator=g).to(dev, dt) for _ in range(3))
        k[:, 700:703] *= 12.0
        ref = F.scaled_dot_product_attention(*(t.view(B, N, H, 64).transpose(1, 2).float() for t in (q, k, v))).transpose(1, 2).reshape(B, N, H * 64)
        ab("flash_attention late-maximum rows", libs, lambda: ops.flash_attention(q, k, v, H), 3,
           check=lambda o: {"new_max_abs_err_vs_fp32": float((o["new"].float() - ref).abs().max())})
    if "xattn" in only:    # the 77-key cross attention (k_flash_attn_smallkv): a streaming kernel -- q read, o written
        for (B, H, N) in [(40, 10, 4096), (40, 20, 1024), (20, 10, 4096), (20, 20, 1024), (12, 20, 1024), (6, 10, 4096)]:
            q = torch.randn(B, N, H * 64, generator=g).to(dev, dt)
            k, v = (torch.randn(B, 77, H * 64, generator=g).to(dev, dt) for _ in range(2))
            ab(f"flash_attention cross B{B} H{H} N{N} Nk77", libs, lambda: ops.flash_attention(q, k, v, H), a.rounds,
               extra={"new_gbs": 2.0 * q.numel() * 2 / 1e3})
    if "attn40" in only:
        for (B, H, N) in [(40, 10, 4096), (40, 20, 1024), (12, 10, 4096), (12, 20, 1024)]:
            q, k, v = (torch.randn(B, N, H * 64, generator=g).to(dev, dt) for _ in range(3))
            flops = 4.0 * B * H * N * N * 64
            ab(f"flash_attention self B{B} H{H} N{N}", libs, lambda: ops.flash_attention(q, k, v, H), a.rounds,
               extra={"new_tflops": flops / 1e6})
    if "sd1" in only:     # SD 1.x head dims (k_flash_attn_gen): self attention at the 64 x 64 / 32 x 32 / 16 x 16 levels of the 512 x 1024 workload
        for (B, H, N, hd) in [(20, 8, 4096, 40), (20, 8, 1024, 80), (20, 8, 256, 160), (6, 8, 4096, 40)]:
            q, k, v = (torch.randn(B, N, H * hd, generator=g).to(dev, dt) for _ in range(3))
            flops = 4.0 * B * H * N * N * hd
            ab(f"flash_attention self B{B} H{H} N{N} hd{hd}", libs, lambda: ops.flash_attention(q, k, v, H), a.rounds,
               extra={"new_tflops": flops / 1e6})
    if "convsplit" in only:      # ops.CONV_BATCH_SPLIT off / on in the SAME library: the batch tail as 128-row tiles
        cl = torch.channels_last
        for (B, Hh, W, Cin, N) in [(40, 32, 32, 1280, 1280), (40, 32, 32, 2560, 1280), (12, 64, 64, 640, 640), (6, 64, 64, 640, 640)]:
            x = (torch.rand(B, Cin, Hh, W, generator=g) * 2 - 1).to(dev, dt).contiguous(memory_format=cl)
            w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev, dt).contiguous(memory_format=cl)
            b = (torch.rand(N, generator=g) * 2 - 1).to(dev, dt)

            def run():
                ops.CONV_BATCH_SPLIT = _hip._LIB is libs["new"]
                try:
                    return ops.conv3x3_nhwc(x, w, b)
                finally:
                    ops.CONV_BATCH_SPLIT = True
            ab(f"conv3x3 batch split {B}x{Hh}x{W} {Cin}->{N}", libs, run, a.rounds, extra={"new_tflops": 2.0 * B * Hh * W * 9 * Cin * N / 1e6})
    if "gn" in only:
        cl = torch.channels_last
        for shape in [(40, 320, 128, 128), (40, 640, 64, 64), (40, 1280, 32, 32), (12, 320, 128, 128), (12, 1280, 32, 32), (20, 320, 128, 128), (20, 960, 128, 128), (20, 640, 64, 64), (20, 1920, 64, 64), (20, 1280, 32, 32), (20, 2560, 32, 32),
                      (6, 320, 128, 128), (6, 640, 64, 64), (6, 1280, 32, 32)]:
            x = torch.randn(*shape, generator=g).to(dev, dt).contiguous(memory_format=cl)
            C = shape[1]
            w, b = (torch.randn(C, generator=g) * 0.2 + 1).to(dev, dt), (torch.randn(C, generator=g) * 0.1).to(dev, dt)
            want = F.silu(F.group_norm(x.float(), 32, w.float(), b.float(), 1e-5))
            ab(f"groupnorm_nhwc {list(shape)}", libs, lambda: ops.groupnorm_nhwc(x, w, b, 32, 1e-5, silu=True), a.rounds,
               extra={"new_gbs": 3.0 * x.numel() * 2 / 1e3},
               check=lambda o: {k_ + "_max_err_vs_fp32": float((o[k_].float() - want).abs().max()) for k_ in ("prev", "new")})
    _hip._LIB = product


if __name__ == "__main__":
    main()
