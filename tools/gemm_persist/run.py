"""GPU probe of the persistent GEMM experiment (tools/gemm_persist/libgemm_persist.so) against the product kernel
(libelastic_hip.so): results must be bit-identical (same arithmetic, same order); interleaved rounds, median.
    python tools/gemm_persist/run.py [--rounds 7] [--grid 256]"""
import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

from elasticdiffusion_official_amd import _hip

_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64


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
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--grid", type=int, default=256)
a = ap.parse_args()
prod = _hip.lib()
P = ctypes.CDLL(os.path.join(HERE, "libgemm_persist.so"))
P.ed_p_geglu_gemm.argtypes = [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]
P.ed_p_linear.argtypes = [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]
P.ed_p2_geglu_gemm.argtypes = P.ed_p_geglu_gemm.argtypes
P.ed_p2_linear.argtypes = P.ed_p_linear.argtypes
P.ed_p_conv3x3_nhwc.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]
st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
g = torch.Generator().manual_seed(0)
dt, cl, G = torch.float16, torch.channels_last, a.grid
cases = []
for (M, K, I) in [(20480, 1280, 5120), (81920, 640, 2560), (6144, 1280, 5120), (24576, 640, 2560), (300, 192, 256)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(2 * I, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(2 * I, generator=g) * 2 - 1).to("cuda", dt)
    o = [torch.empty(M, I, device="cuda", dtype=dt) for _ in range(2)]
    cases.append((f"geglu {M}x{K}->{I}", 4.0 * M * K * I, o,
                  lambda o, x=x, w=w, b=b, M=M, K=K, I=I: prod.ed_geglu_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, st()),
                  lambda o, x=x, w=w, b=b, M=M, K=K, I=I: P.ed_p_geglu_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, G, st()),
                  lambda o, x=x, w=w, b=b, M=M, K=K, I=I: P.ed_p2_geglu_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, G, st())))
for (M, K, N, res) in [(81920, 640, 640, 0), (81920, 640, 640, 1), (81920, 640, 1920, 0), (81920, 2560, 640, 1), (20480, 1280, 1280, 0),
                       (20480, 1280, 3840, 0), (20480, 5120, 1280, 1), (1000, 320, 200, 0), (1000, 384, 200, 1), (2000, 448, 520, 0)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    r = (torch.rand(M, N, generator=g) * 2 - 1).to("cuda", dt) if res else None
    rp = r.data_ptr() if res else None
    o = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(2)]
    cases.append((f"linear {M}x{K}->{N}{' + residual' if res else ''}", 2.0 * M * K * N, o,
                  lambda o, x=x, w=w, b=b, M=M, K=K, N=N, rp=rp, r=r: prod.ed_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), rp, o.data_ptr(), 1, M, K, N, st()),
                  lambda o, x=x, w=w, b=b, M=M, K=K, N=N, rp=rp, r=r: P.ed_p_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), rp, o.data_ptr(), 1, M, K, N, G, st()),
                  lambda o, x=x, w=w, b=b, M=M, K=K, N=N, rp=rp, r=r: P.ed_p2_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), rp, o.data_ptr(), 1, M, K, N, G, st())))
for (B, H, W, Cin, N) in [(20, 32, 32, 1280, 1280), (20, 128, 128, 320, 320), (2, 12, 20, 64, 200)]:
    x = (torch.rand(B, Cin, H, W, generator=g) * 2 - 1).to("cuda", dt).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to("cuda", dt).contiguous(memory_format=cl)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    o = [torch.empty(B, N, H, W, device="cuda", dtype=dt).contiguous(memory_format=cl) for _ in range(2)]
    cases.append((f"conv {B}x{H}x{W} {Cin}->{N} (bias only)", 2.0 * B * H * W * 9 * Cin * N, o,
                  lambda o, x=x, w=w, b=b, B=B, H=H, W=W, Cin=Cin, N=N: prod.ed_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, o.data_ptr(), 1, B, H, W, Cin, N, st()),
                  lambda o, x=x, w=w, b=b, B=B, H=H, W=W, Cin=Cin, N=N: P.ed_p_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, o.data_ptr(), 1, B, H, W, Cin, N, G, st()),
                  None))
for name, flops, o, call_prod, call_p, call_p2 in cases:
    o[0].zero_(), o[1].zero_()
    assert call_prod(o[0]) == 0 and call_p(o[1]) == 0
    torch.cuda.synchronize()
    same = bool(torch.equal(o[0], o[1]))
    again = all(call_p(o[1]) == 0 and bool(torch.equal(o[0], o[1])) for _ in range(5))
    same2 = None
    if call_p2 is not None:      # v2: the next tile's K tile 0 staged during the last K tile (plain GEMM, even K tile count >= 6)
        o[1].zero_()
        same2 = all(call_p2(o[1]) == 0 and bool(torch.equal(o[0], o[1])) for _ in range(6))
    tp, tq, t2 = [], [], []
    for _ in range(a.rounds):
        tp.append(timed(lambda: call_prod(o[0])))
        tq.append(timed(lambda: call_p(o[1])))
        if call_p2 is not None:
            t2.append(timed(lambda: call_p2(o[1])))
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    mp, mq = med(tp), med(tq)
    rec = {"case": name, "bit_identical": same and again, "product_us": round(1e3 * mp, 1), "persistent_us": round(1e3 * mq, 1),
           "product_tflops": round(flops / mp / 1e9, 1), "persistent_tflops": round(flops / mq / 1e9, 1), "speedup": round(mp / mq, 4)}
    if t2:
        rec.update({"v2_bit_identical": same2, "v2_us": round(1e3 * med(t2), 1), "v2_tflops": round(flops / med(t2) / 1e9, 1),
                    "v2_speedup": round(mp / med(t2), 4)})
    print(json.dumps(rec), flush=True)
