"""GPU probe of the 16x16x32 attention experiment (tools/attn16/libattn16.so) against the product kernel (ed_flash_attention, v_path 5):
accuracy of both against an fp32 reference (incl. an outlier case that forces the lazy loop's exact slow path), then interleaved timing at
the SDXL self-attention shapes (q / k / v = column slices of a fused projection, as inside the UNet).
    python tools/attn16/run.py [--rounds 5]"""
import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

from elasticdiffusion_official_amd import ops

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
X = ctypes.CDLL(os.path.join(HERE, "libattn16.so"))
X.ed_x_flash_attention16.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_i64] * 8 + [_f, _vp, _vp]


def attn16(q, k, v, H, timing=None):
    B, Nq, HD = q.shape
    out = torch.empty(B, Nq, HD, dtype=q.dtype, device=q.device)
    tptr = None if timing is None else (1 if timing == "nocheck" else 2 if timing == "inregion" else timing.data_ptr())
    rc = X.ed_x_flash_attention16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1 if q.dtype == torch.float16 else 2, B, H, Nq, k.shape[1],
                                  q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                  64 ** -0.5, tptr, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


X5 = None


def attn5_inregion(q, k, v, H):
    global X5
    if X5 is None:
        X5 = ctypes.CDLL(os.path.join(HERE, "libattn5_inregion.so"))
        X5.ed_x_flash_attention5_inregion.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_i64] * 8 + [_f, _vp]
    B, Nq, HD = q.shape
    out = torch.empty(B, Nq, HD, dtype=q.dtype, device=q.device)
    rc = X5.ed_x_flash_attention5_inregion(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1 if q.dtype == torch.float16 else 2, B, H, Nq,
                                           k.shape[1], q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0),
                                           out.stride(1), 64 ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


def ref(q, k, v, H):
    B, N, HD = q.shape
    f = lambda t: t.float().reshape(B, t.shape[1], H, 64).transpose(1, 2)   # noqa: E731
    s = f(q) @ f(k).transpose(-1, -2) * 64 ** -0.5
    return (torch.softmax(s, -1) @ f(v)).transpose(1, 2).reshape(B, N, HD)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--segments", action="store_true", help="only the s_memtime build: where a wave's cycles go")
ap.add_argument("--inregion", action="store_true", help="only the variant with the global loads / LDS writes inside the MFMA region")
ap.add_argument("--product-inregion", action="store_true", help="only the PRODUCT kernel's in-region variant (attn5_inregion.hip) vs v_path 5")
ap.add_argument("--nocheck", action="store_true", help="only the ablation without the lazy loop's per-tile check")
a = ap.parse_args()
assert (ops.ED_F16, ops.ED_BF16) == (1, 2)
g = torch.Generator(device="cuda").manual_seed(0)
SEG = ["global loads issued", "S(t+1) = K Q^T half (16 MFMAs + 16 softmax slices)", "O += V^T P^T half (16 MFMAs + 16 slices)",
       "lazy check (+ slow path, rescale)", "wait for the global loads + LDS writes", "barrier", "drain + epilogue", "before the iteration"]
if a.segments:
    for (B, H, N) in [(20, 10, 4096), (20, 20, 1024)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(torch.float16)
        q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
        tm = torch.zeros(64 * 4 * 10, device="cuda", dtype=torch.int32)
        plain = timed(lambda: attn16(q, k, v, H))
        stamped = timed(lambda: attn16(q, k, v, H, tm))
        want = attn16(q, k, v, H)
        same = bool(torch.equal(want, attn16(q, k, v, H, tm)))
        t = tm.cpu().reshape(64 * 4, 10).double()
        tiles = float(t[0, 9])
        rec = {"check": "segments", "B": B, "H": H, "N": N, "plain_us": round(1e3 * plain, 1), "stamped_us": round(1e3 * stamped, 1),
               "same_result": same, "tiles": tiles, "wave_total_cycles_mean": round(float(t[:, 8].mean()), 0),
               "cycles_per_tile": {SEG[i]: round(float(t[:, i].mean()) / tiles, 1) for i in range(6)},
               "cycles_once": {SEG[6]: round(float(t[:, 6].mean()), 0), SEG[7]: round(float(t[:, 7].mean()), 0)},
               "share_of_wave_total": {SEG[i]: round(float(t[:, i].mean() / t[:, 8].mean()), 3) for i in range(8)},
               "spread_over_waves_cycles_per_tile_min_max": {SEG[i]: [round(float(t[:, i].min()) / tiles, 1), round(float(t[:, i].max()) / tiles, 1)] for i in (1, 2, 4, 5)}}
        print(json.dumps(rec), flush=True)
    sys.exit(0)
if a.product_inregion:
    for dt in (torch.float16, torch.bfloat16):
        for (B, H, Nq, Nk, outlier) in [(2, 3, 256, 256, False), (1, 2, 200, 200, False), (2, 2, 384, 1024, False), (1, 2, 256, 512, True), (1, 3, 130, 333, False)]:
            qkv = torch.randn(B, max(Nq, Nk), 3 * H * 64, device="cuda", generator=g).to(dt)
            q, k, v = qkv[:, :Nq, :H * 64], qkv[:, :Nk, H * 64:2 * H * 64], qkv[:, :Nk, 2 * H * 64:]
            if outlier:
                k = k.clone()
                k[:, 300:303] = (q[:, 5:8] * 6).to(dt)
                k[:, 450] = (q[:, 100] * 12).to(dt)
            same = all(bool(torch.equal(ops.flash_attention(q, k, v, H, v_path=5), attn5_inregion(q, k, v, H))) for _ in range(4))
            print(json.dumps({"check": "product kernel, in-region variant == v_path 5, bit for bit (4 runs)", "dtype": str(dt)[6:], "B": B, "H": H, "Nq": Nq,
                              "Nk": Nk, "outlier_keys": outlier, "ok": same}), flush=True)
    for dt in (torch.float16, torch.bfloat16):
        for (B, H, N) in [(20, 10, 4096), (20, 20, 1024), (6, 10, 4096), (6, 20, 1024)]:
            qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(dt)
            q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
            t5, tn = [], []
            for _ in range(a.rounds):
                t5.append(timed(lambda: ops.flash_attention(q, k, v, H, v_path=5)))
                tn.append(timed(lambda: attn5_inregion(q, k, v, H)))
            med = lambda x: sorted(x)[len(x) // 2]   # noqa: E731
            flops = 4.0 * B * H * N * N * 64
            print(json.dumps({"check": "timing: product kernel with loads / LDS writes inside the region", "dtype": str(dt)[6:], "B": B, "H": H, "N": N,
                              "product_v5_tflops": round(flops / med(t5) / 1e9, 1), "inregion_tflops": round(flops / med(tn) / 1e9, 1),
                              "speedup": round(med(t5) / med(tn), 4)}), flush=True)
    sys.exit(0)
if a.inregion:
    for (B, H, Nq, Nk) in [(2, 3, 256, 256), (1, 2, 200, 128), (2, 2, 384, 1024)]:
        qkv = torch.randn(B, max(Nq, Nk), 3 * H * 64, device="cuda", generator=g).to(torch.float16)
        q, k, v = qkv[:, :Nq, :H * 64], qkv[:, :Nk, H * 64:2 * H * 64], qkv[:, :Nk, 2 * H * 64:]
        same = all(bool(torch.equal(attn16(q, k, v, H), attn16(q, k, v, H, "inregion"))) for _ in range(5))
        print(json.dumps({"check": "in-region variant == plain variant, bit for bit (5 runs)", "B": B, "H": H, "Nq": Nq, "Nk": Nk, "ok": same}), flush=True)
    for (B, H, N) in [(20, 10, 4096), (20, 20, 1024), (6, 10, 4096), (6, 20, 1024)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(torch.float16)
        q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
        t5, t16, tn = [], [], []
        for _ in range(a.rounds):
            t5.append(timed(lambda: ops.flash_attention(q, k, v, H, v_path=5)))
            t16.append(timed(lambda: attn16(q, k, v, H)))
            tn.append(timed(lambda: attn16(q, k, v, H, "inregion")))
        med = lambda x: sorted(x)[len(x) // 2]   # noqa: E731
        same = bool(torch.equal(attn16(q, k, v, H), attn16(q, k, v, H, "inregion")))
        flops = 4.0 * B * H * N * N * 64
        print(json.dumps({"check": "timing: loads / LDS writes inside the region", "B": B, "H": H, "N": N, "product_v5_tflops": round(flops / med(t5) / 1e9, 1),
                          "x16_tflops": round(flops / med(t16) / 1e9, 1), "x16_inregion_tflops": round(flops / med(tn) / 1e9, 1),
                          "inregion_over_x16": round(med(t16) / med(tn), 4), "inregion_over_product": round(med(t5) / med(tn), 4), "same_result": same}), flush=True)
    sys.exit(0)
if a.nocheck:
    for (B, H, N) in [(20, 10, 4096), (20, 20, 1024)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(torch.float16)
        q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
        t5, t16, tn = [], [], []
        for _ in range(a.rounds):
            t5.append(timed(lambda: ops.flash_attention(q, k, v, H, v_path=5)))
            t16.append(timed(lambda: attn16(q, k, v, H)))
            tn.append(timed(lambda: attn16(q, k, v, H, "nocheck")))
        med = lambda x: sorted(x)[len(x) // 2]   # noqa: E731
        same = bool(torch.equal(attn16(q, k, v, H), attn16(q, k, v, H, "nocheck")))
        flops = 4.0 * B * H * N * N * 64
        print(json.dumps({"check": "ablation: no per-tile check", "B": B, "H": H, "N": N, "product_v5_tflops": round(flops / med(t5) / 1e9, 1),
                          "x16_tflops": round(flops / med(t16) / 1e9, 1), "x16_nocheck_tflops": round(flops / med(tn) / 1e9, 1),
                          "nocheck_over_x16": round(med(t16) / med(tn), 4), "same_result_on_this_data": same}), flush=True)
    sys.exit(0)
# ---- accuracy -------------------------------------------------------------------------------------------------------------------
for dt in (torch.float16, torch.bfloat16):
    for (B, H, Nq, Nk, outlier) in [(2, 3, 256, 256, False), (1, 2, 200, 128, False), (2, 2, 384, 1024, False), (1, 2, 256, 512, True)]:
        qkv = torch.randn(B, max(Nq, Nk), 3 * H * 64, device="cuda", generator=g).to(dt)
        q, k, v = qkv[:, :Nq, :H * 64], qkv[:, :Nk, H * 64:2 * H * 64], qkv[:, :Nk, 2 * H * 64:]
        if outlier:   # late keys that beat the reference of their rows by far more than 2^6: the exact slow path must run
            k = k.clone()
            k[:, 300:303] = (q[:, 5:8] * 6).to(dt)
            k[:, 450] = (q[:, 100] * 12).to(dt)
        want = ref(q, k, v, H)
        e16 = float((attn16(q, k, v, H).float() - want).abs().max())
        e5 = float((ops.flash_attention(q, k, v, H, v_path=5).float() - want).abs().max())
        print(json.dumps({"check": "accuracy", "dtype": str(dt)[6:], "B": B, "H": H, "Nq": Nq, "Nk": Nk, "outlier_keys": outlier,
                          "max_abs_err_16x16x32": round(e16, 5), "max_abs_err_product_v5": round(e5, 5),
                          "ok": bool(e16 <= 1.5 * e5 + 2e-3)}), flush=True)
# ---- timing ---------------------------------------------------------------------------------------------------------------------
for dt in (torch.float16, torch.bfloat16):
    for (B, H, N) in [(20, 10, 4096), (20, 20, 1024), (6, 10, 4096), (6, 20, 1024)]:
        qkv = torch.randn(B, N, 3 * H * 64, device="cuda", generator=g).to(dt)
        q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
        t5, t16 = [], []
        for _ in range(a.rounds):
            t5.append(timed(lambda: ops.flash_attention(q, k, v, H, v_path=5)))
            t16.append(timed(lambda: attn16(q, k, v, H)))
        m5, m16 = sorted(t5)[len(t5) // 2], sorted(t16)[len(t16) // 2]
        flops = 4.0 * B * H * N * N * 64
        print(json.dumps({"check": "timing", "dtype": str(dt)[6:], "B": B, "H": H, "N": N, "product_v5_us": round(1e3 * m5, 1),
                          "x16_us": round(1e3 * m16, 1), "product_v5_tflops": round(flops / m5 / 1e9, 1),
                          "x16_tflops": round(flops / m16 / 1e9, 1), "speedup": round(m5 / m16, 4)}), flush=True)
