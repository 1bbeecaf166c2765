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
X.ed_x_flash_attention16.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_i64] * 8 + [_f, _vp]


def attn16(q, k, v, H):
    B, Nq, HD = q.shape
    out = torch.empty(B, Nq, HD, dtype=q.dtype, device=q.device)
    rc = X.ed_x_flash_attention16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1 if q.dtype == torch.float16 else 2, B, H, Nq, k.shape[1],
                                  q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                  64 ** -0.5, torch.cuda.current_stream().cuda_stream)
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
a = ap.parse_args()
assert (ops.ED_F16, ops.ED_BF16) == (1, 2)
g = torch.Generator(device="cuda").manual_seed(0)
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
