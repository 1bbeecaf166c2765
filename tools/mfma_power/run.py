"""Register-only MFMA loop on the MI355X (tools/mfma_power/mfma_power.hip): sustained TFLOP/s by MFMA shape, type and operand bit activity.
    python tools/mfma_power/run.py"""
import ctypes
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libmfma_power.so"))
_vp, _i = ctypes.c_void_p, ctypes.c_int
L.ed_mfma_loop.argtypes = [_i, _i, _i, _vp, _vp, _i, _i, _vp]
g = torch.Generator().manual_seed(0)
ITERS = 20000


def operands(kind, blocks):
    n = blocks * 512 * 8 * 8        # 8 x 16 bytes per thread = 64 16-bit values
    r = torch.rand(n, generator=g) * 2 - 1
    if kind == "f16_random":
        return r.to("cuda", torch.float16)
    if kind == "bf16_random":
        return r.to("cuda", torch.bfloat16)
    if kind == "f16_bf16_mantissas":
        return r.to("cuda", torch.bfloat16).to(torch.float16)
    return torch.zeros(n, device="cuda", dtype=torch.float16)


def timed(shape, bf, buf, blocks, nacc):
    out = torch.empty(blocks * 512, device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    assert L.ed_mfma_loop(shape, bf, nacc, buf.data_ptr(), out.data_ptr(), blocks, 200, st) == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.ed_mfma_loop(shape, bf, nacc, buf.data_ptr(), out.data_ptr(), blocks, ITERS, st) == 0
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    flops = 2.0 * blocks * 8 * ITERS * 131072
    return ms, flops / ms / 1e9, bool(torch.isfinite(out).all())


for blocks, label in ((256, "2 waves / SIMD"), (512, "4 waves / SIMD")):
    for kind in ("f16_random", "f16_bf16_mantissas", "bf16_random", "zeros"):
        buf = operands(kind, blocks)
        bf = 1 if kind == "bf16_random" else 0
        rec = {"occupancy": label, "operands": kind}
        for shape, naccs in ((16, (2, 4, 8, 16)), (32, (1, 2, 4, 8))):
            for nacc in (naccs if kind in ("f16_random", "zeros") else naccs[-2:]):
                ms, tf, fin = timed(shape, bf, buf, blocks, nacc)
                rec[("16x16x32" if shape == 16 else "32x32x16") + f" acc {nacc}"] = round(tf, 1)
                assert fin
        print(json.dumps(rec), flush=True)
