"""In-process A/B of two builds of csrc/gemm_kernels.hip (tools/gemm_ab/libgemm_prev.so = the committed kernel, libgemm_new.so = the
working tree's) on the UNet's largest shapes, interleaved rounds, median; results of the two builds must be bit-identical.
    python tools/gemm_ab/run.py [--rounds 7]"""
import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch

from elasticdiffusion_official_amd import _hip


def load(name):
    L = ctypes.CDLL(name if os.path.isabs(name) or os.sep in name else os.path.join(HERE, name))
    for fn in ("ed_geglu_gemm", "ed_linear", "ed_conv3x3_nhwc"):
        getattr(L, fn).argtypes = _hip.SIGNATURES[fn]
        getattr(L, fn).restype = ctypes.c_int
    return L


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
ap.add_argument("--prev", default="libgemm_prev.so", help="a file in tools/gemm_ab/, or a path (round 6: tools/ab/libelastic_hip_r5.so = the round-5 library)")
ap.add_argument("--new", default="libgemm_new.so", help="... or a path (elasticdiffusion_official_amd/libelastic_hip.so = the product)")
a = ap.parse_args()
libs = {"prev": load(a.prev), "new": load(a.new)}
st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
g = torch.Generator().manual_seed(0)
dt, cl = torch.float16, torch.channels_last
cases = []
for (M, K, I) in [(20480, 1280, 5120), (81920, 640, 2560), (6144, 1280, 5120), (24576, 640, 2560)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(2 * I, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(2 * I, generator=g) * 2 - 1).to("cuda", dt)
    out = {n: torch.empty(M, I, device="cuda", dtype=dt) for n in libs}
    cases.append((f"geglu {M}x{K}->{I}", 4.0 * M * K * I, out,
                  lambda L, o, x=x, w=w, b=b, M=M, K=K, I=I: L.ed_geglu_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), 1, M, K, I, st())))
for (M, K, N) in [(81920, 640, 640), (81920, 640, 1920), (81920, 2560, 640), (20480, 1280, 1280)]:
    x = (torch.rand(M, K, generator=g) * 2 - 1).to("cuda", dt)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to("cuda", dt)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    out = {n: torch.empty(M, N, device="cuda", dtype=dt) for n in libs}
    cases.append((f"linear {M}x{K}->{N}", 2.0 * M * K * N, out,
                  lambda L, o, x=x, w=w, b=b, M=M, K=K, N=N: L.ed_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), 1, M, K, N, st())))
for (B, H, W, Cin, N) in [(20, 32, 32, 1280, 1280), (20, 64, 64, 640, 640), (20, 128, 128, 320, 320), (6, 32, 32, 1280, 1280)]:
    x = (torch.rand(B, Cin, H, W, generator=g) * 2 - 1).to("cuda", dt).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to("cuda", dt).contiguous(memory_format=cl)
    b = (torch.rand(N, generator=g) * 2 - 1).to("cuda", dt)
    sb = (torch.rand(B, N, generator=g) * 2 - 1).to("cuda", dt)
    out = {n: torch.empty(B, N, H, W, device="cuda", dtype=dt).contiguous(memory_format=cl) for n in libs}
    cases.append((f"conv {B}x{H}x{W} {Cin}->{N}", 2.0 * B * H * W * 9 * Cin * N, out,
                  lambda L, o, x=x, w=w, b=b, sb=sb, B=B, H=H, W=W, Cin=Cin, N=N: L.ed_conv3x3_nhwc(
                      x.data_ptr(), w.data_ptr(), b.data_ptr(), sb.data_ptr(), None, o.data_ptr(), 1, B, H, W, Cin, N, st())))
for name, flops, out, call in cases:
    for n, L in libs.items():
        assert call(L, out[n]) == 0
    torch.cuda.synchronize()
    same = bool(torch.equal(out["prev"], out["new"]))
    t = {n: [] for n in libs}
    for _ in range(a.rounds):
        for n, L in libs.items():
            t[n].append(timed(lambda: call(L, out[n])))
    med = {n: sorted(v)[len(v) // 2] for n, v in t.items()}
    print(json.dumps({"case": name, "bit_identical": same, "prev_us": round(1e3 * med["prev"], 1), "new_us": round(1e3 * med["new"], 1),
                      "prev_tflops": round(flops / med["prev"] / 1e9, 1), "new_tflops": round(flops / med["new"] / 1e9, 1),
                      "speedup": round(med["prev"] / med["new"], 4)}), flush=True)
