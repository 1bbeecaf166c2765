"""GPU probe of the GEMM schedule experiment (tools/gemm_sched/libgemm_sched.so) against the product kernel (libelastic_hip.so):
every schedule must be bit-identical to the product (same arithmetic, same order); interleaved rounds, median.
    python tools/gemm_sched/run.py [--rounds 5] [--scheds 0,1,2]"""
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
NAMES = ["product_order", "late", "mid", "spread", "mfma_all", "r4", "two_read", "two_mfma", "half_tiles"]


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
ap.add_argument("--scheds", default=None)
a = ap.parse_args()
prod = _hip.lib()
S = ctypes.CDLL(os.path.join(HERE, "libgemm_sched.so"))
S.ed_s_geglu_gemm.argtypes = [_i, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]
S.ed_s_linear.argtypes = [_i, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]
S.ed_s_conv3x3_nhwc.argtypes = [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
scheds = [int(v) for v in a.scheds.split(",")] if a.scheds else list(range(S.ed_s_count()))
st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
g = torch.Generator().manual_seed(0)
dt = torch.float16
cases = []
for (M, K, I) in [(20480, 1280, 5120), (81920, 640, 2560), (6144, 1280, 5120), (300, 192, 256)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(2 * I, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(2 * I, generator=g) * 2 - 1).to("cuda", dt)
    o = [torch.empty(M, I, device="cuda", dtype=dt) for _ in range(2)]
    cases.append((f"geglu {M}x{K}->{I}", 4.0 * M * K * I, o,
                  lambda o, x=x, w=w, b=b, M=M, K=K, I=I: prod.ed_geglu_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, st()),
                  lambda s, o, x=x, w=w, b=b, M=M, K=K, I=I: S.ed_s_geglu_gemm(s, x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, st())))
for (M, K, N) in [(81920, 640, 640), (81920, 640, 1920), (20480, 1280, 1280), (20480, 1280, 3840), (20480, 5120, 1280), (8192, 8192, 8192),
                  (81920, 2560, 640), (327680, 640, 320), (327680, 960, 320), (1000, 320, 200), (2000, 448, 520), (500, 64, 256), (700, 128, 304), (300, 256, 104)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    o = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(2)]
    cases.append((f"linear {M}x{K}->{N}", 2.0 * M * K * N, o,
                  lambda o, x=x, w=w, b=b, M=M, K=K, N=N: prod.ed_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), 1, M, K, N, st()),
                  lambda s, o, x=x, w=w, b=b, M=M, K=K, N=N: S.ed_s_linear(s, x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, N, st())))
conv_cases = []
for (B, H, W, Cin, N) in [(20, 32, 32, 1280, 1280), (20, 64, 64, 640, 640), (20, 128, 128, 320, 320), (6, 32, 32, 1280, 1280), (2, 12, 20, 64, 200)]:
    cl = torch.channels_last
    x = (torch.rand(B, Cin, H, W, generator=g) * 2 - 1).to("cuda", dt).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to("cuda", dt).contiguous(memory_format=cl)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    o = [torch.empty(B, N, H, W, device="cuda", dtype=dt).contiguous(memory_format=cl) for _ in range(2)]
    conv_cases.append((f"conv {B}x{H}x{W} {Cin}->{N} (bias only)", 2.0 * B * H * W * 9 * Cin * N, o,
                       lambda o, x=x, w=w, b=b, B=B, H=H, W=W, Cin=Cin, N=N: prod.ed_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, o.data_ptr(), 1, B, H, W, Cin, N, st()),
                       lambda s, o, x=x, w=w, b=b, B=B, H=H, W=W, Cin=Cin, N=N: S.ed_s_conv3x3_nhwc(s, x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, B, H, W, Cin, N, st())))
all_scheds = scheds
for name, flops, o, call_prod, call_s in cases + conv_cases:
    scheds = [s_ for s_ in all_scheds if s_ in (0, 6)] if name.startswith("conv") else all_scheds     # the convolution has two schedules
    if not name.startswith("geglu") and not a.scheds:      # + the half-tile mode (value half only where the gate half is beyond N)
        scheds = scheds + [8]
    o[0].zero_()
    assert call_prod(o[0]) == 0
    torch.cuda.synchronize()
    same = {}
    for s in scheds:
        ok = True
        for _ in range(3):
            o[1].zero_()
            assert call_s(s, o[1]) == 0
            ok = ok and bool(torch.equal(o[0], o[1]))
        same[s] = ok
    n = 10 if flops < 5e11 else 3
    tp, ts = [], {s: [] for s in scheds}
    for _ in range(a.rounds):
        tp.append(timed(lambda: call_prod(o[0]), n))
        for s in scheds:
            ts[s].append(timed(lambda: call_s(s, o[1]), n))
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    mp = med(tp)
    rec = {"case": name, "product_us": round(1e3 * mp, 1), "product_tflops": round(flops / mp / 1e9, 1)}
    for s in scheds:
        rec[NAMES[s]] = {"bit_identical": same[s], "us": round(1e3 * med(ts[s]), 1), "tflops": round(flops / med(ts[s]) / 1e9, 1),
                         "speedup": round(mp / med(ts[s]), 4)}
    print(json.dumps(rec), flush=True)
